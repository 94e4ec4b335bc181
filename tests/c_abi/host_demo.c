/* A plain-C host of libaero_stark.so: what a compiled (C / C++ / Rust-over-FFI) caller of the drop-in boundary does, without any
 * Python in the process. Counterpart of miden-proof-generator/src/main.rs:23-51 (options, prove, self-verify, ProofData container)
 * for the built-in AIR:
 *      host_demo <log_n> <width> <out.bin>
 * builds the synthetic trace in (pinned) host memory, proves it through aero_prove_fib_host, verifies the proof with the library's
 * host-side verifier under a caller-side policy, re-encodes it as protobuf, writes the container and prints one line.
 * Exit codes: 0 ok, 2 no GPU (AERO_E_HIP from aero_ctx_create: there is no CPU fallback), 1 anything else.
 * Built by tests/test_c_abi_host.py with: gcc -std=c11 -Wall -Wextra -Werror -I include host_demo.c -L aero_amd -laero_stark */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aero_stark.h"

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: host_demo log_n width out.bin\n"); return 1; }
    const uint32_t log_n = (uint32_t)atoi(argv[1]), width = (uint32_t)atoi(argv[2]);
    aero_ctx* ctx = NULL;
    int32_t rc = aero_ctx_create(0, &ctx);
    if (rc == AERO_E_HIP) { printf("no device: %s\n", aero_last_error(NULL)); return 2; }
    if (rc != AERO_OK) { fprintf(stderr, "ctx_create: %d %s\n", rc, aero_last_error(NULL)); return 1; }

    const size_t cells = (size_t)width << log_n;
    uint64_t* trace = (uint64_t*)malloc(cells * 8);
    if (!trace || aero_fib_trace(width, log_n, trace) != AERO_OK) { fprintf(stderr, "trace\n"); return 1; }
    if (aero_host_register(trace, cells * 8) != AERO_OK) { fprintf(stderr, "pin: %s\n", aero_last_error(NULL)); return 1; }

    const aero_proof_options opt = {27, 8, 16, 4, 1, 8, 8};       /* ProofOptions::with_96_bit_security() */
    uint8_t* proof = NULL;
    size_t proof_len = 0;
    uint64_t* pub = (uint64_t*)calloc(width / 2, 8);
    rc = aero_prove_fib_host(ctx, trace, width, log_n, &opt, &proof, &proof_len, pub);
    if (rc != AERO_OK) { fprintf(stderr, "prove: %d %s\n", rc, aero_last_error(ctx)); return 1; }

    /* verification is the caller's business: state what is acceptable (security floor, trace length, exact options) */
    aero_fib_air air = {0, 0, 2};
    aero_verify_policy pol;
    memset(&pol, 0, sizeof pol);
    pol.min_query_security_bits = 96; pol.expected_log_n = log_n; pol.require_options = 1; pol.options = opt;
    char err[256];
    rc = aero_verify_fib(proof, proof_len, pub, width / 2, &air, &pol, err, sizeof err);
    if (rc != AERO_OK) { fprintf(stderr, "verify: %d %s\n", rc, err); return 1; }
    /* a wrong statement must be rejected */
    pub[0] ^= 1;
    if (aero_verify_fib(proof, proof_len, pub, width / 2, &air, &pol, err, sizeof err) != AERO_E_VERIFY) { fprintf(stderr, "accepted a false statement\n"); return 1; }
    pub[0] ^= 1;

    uint8_t *pb = NULL, *container = NULL;
    size_t pb_len = 0, container_len = 0;
    if (aero_proof_to_protobuf(proof, proof_len, &pb, &pb_len, err, sizeof err) != AERO_OK) { fprintf(stderr, "protobuf: %s\n", err); return 1; }
    if (aero_proof_container((const uint8_t*)pub, (size_t)(width / 2) * 8, proof, proof_len, &container, &container_len) != AERO_OK) return 1;
    FILE* f = fopen(argv[3], "wb");
    if (!f || fwrite(container, 1, container_len, f) != container_len) { fprintf(stderr, "write\n"); return 1; }
    fclose(f);

    uint32_t qbits = 0, fbits = 0;
    aero_proof_security_bits(proof, proof_len, &qbits, &fbits);
    printf("proved %u x 2^%u: %zu proof bytes, %zu protobuf bytes, query security %u bits, container %zu bytes, result[0] = %llu\n", width, log_n,
           proof_len, pb_len, qbits, container_len, (unsigned long long)pub[0]);

    aero_free(pb); aero_free(container); aero_free(proof);
    aero_host_unregister(trace);
    free(trace); free(pub);
    aero_ctx_destroy(ctx);
    return 0;
}
