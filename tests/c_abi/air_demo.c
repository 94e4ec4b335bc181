/* A foreign AIR through the C ABI from a plain C host (include/aero_air.h): the constraint system arrives as an AEROAIR program, the
 * trace in host memory; prove, verify with the out-of-domain check over the program, reject a wrong statement. The program here is
 * the library's VM-shaped synthetic one (a Miden build would export its own `Air` once); what the reference does in
 * miden-proof-generator/src/main.rs:23-51 with the AIR compiled in.
 *     air_demo <log_n> <pairs> <aux> <out.proof>       exit 0 = proven + verified, 2 = no GPU (the library has no CPU fallback) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aero_air.h"

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int32_t rc_ = (call);                                                             \
        if (rc_ != AERO_OK) {                                                             \
            printf("%s failed: %d (%s)\n", #call, rc_, aero_last_error(ctx));             \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 5) { printf("usage: air_demo log_n pairs aux out.proof\n"); return 64; }
    const uint32_t log_n = (uint32_t)atoi(argv[1]), pairs = (uint32_t)atoi(argv[2]), aux = (uint32_t)atoi(argv[3]), rands = 4;
    aero_ctx* ctx = NULL;
    if (aero_device_count() < 1 || aero_ctx_create(0, &ctx) != AERO_OK) { printf("no device: %s\n", aero_last_error(NULL)); return 2; }

    uint8_t* program = NULL; size_t program_len = 0;
    CHECK(aero_air_synth_vm_program(log_n, pairs, aux, rands, &program, &program_len));
    aero_air* air = NULL;
    char err[256] = {0};
    if (aero_air_load(program, program_len, &air, err, sizeof err) != AERO_OK) { printf("aero_air_load: %s\n", err); return 1; }
    uint32_t info[16];
    CHECK(aero_air_info(air, info));
    const uint32_t width = info[0], n_pub = info[3];
    printf("program: %zu bytes, %u + %u columns, %u transition constraints, %u assertions, %u composition columns\n", program_len, info[0], info[1],
           info[5] + info[6], info[7] + info[8], info[9]);

    uint64_t* trace = malloc(((size_t)width << log_n) * sizeof(uint64_t));
    uint64_t* pub = malloc((n_pub ? n_pub : 1) * sizeof(uint64_t));
    if (!trace || !pub) return 1;
    CHECK(aero_air_synth_vm_trace(log_n, pairs, trace, pub));

    const aero_proof_options opt = {27, 8, 16, 4, 1, 4, 8};
    CHECK(aero_air_jit_compile(air, log_n, 1, 1));                 /* optional: the kernel is otherwise built inside the first proof */
    uint8_t* proof = NULL; size_t proof_len = 0;
    CHECK(aero_prove_air_host(ctx, air, trace, log_n, pub, n_pub, &opt, &proof, &proof_len));

    aero_verify_policy policy;
    memset(&policy, 0, sizeof policy);
    policy.min_query_security_bits = 96;
    policy.expected_log_n = log_n;
    if (aero_verify_air(proof, proof_len, pub, n_pub, air, &policy, err, sizeof err) != AERO_OK) { printf("verification failed: %s\n", err); return 1; }
    pub[0] ^= 1;                                                   /* another statement */
    if (aero_verify_air(proof, proof_len, pub, n_pub, air, &policy, err, sizeof err) == AERO_OK) { printf("a wrong public input was accepted\n"); return 1; }

    FILE* f = fopen(argv[4], "wb");
    if (!f || fwrite(proof, 1, proof_len, f) != proof_len) return 1;
    fclose(f);
    printf("proved and verified 2^%u x %u: %zu proof bytes\n", log_n, width, proof_len);
    aero_free(proof); aero_free(program); free(trace); free(pub);
    aero_air_free(air);
    aero_ctx_destroy(ctx);
    return 0;
}
