/* One proof sharded over `world` ranks of ONE process (a context + a host thread per rank on GPU 0, exchanges through the
 * library's in-process communicator: stream-ordered device copies, no RCCL, no Python), the trace handed over in host memory:
 *      sharded_local <log_n> <width> <world> [aux_width aux_rands aux_degree]
 * Every rank's proof must equal the single-GPU proof of the same trace. Prints one JSON line. Exit codes as host_demo.c.
 * Reference: the gather steps these exchanges replace are the worker-pool fan-ins of proving_worker.rs:302-310,428-437. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aero_stark.h"

int main(int argc, char** argv) {
    if (argc != 4 && argc != 7) { fprintf(stderr, "usage: sharded_local log_n width world [aux_width aux_rands aux_degree]\n"); return 1; }
    const uint32_t log_n = (uint32_t)atoi(argv[1]), width = (uint32_t)atoi(argv[2]), world = (uint32_t)atoi(argv[3]);
    aero_fib_air air = {0, 0, 2};
    if (argc == 7) { air.aux_width = (uint32_t)atoi(argv[4]); air.aux_rands = (uint32_t)atoi(argv[5]); air.aux_degree = (uint32_t)atoi(argv[6]); }
    aero_ctx* ctx = NULL;
    int32_t rc = aero_ctx_create(0, &ctx);
    if (rc == AERO_E_HIP) { printf("no device: %s\n", aero_last_error(NULL)); return 2; }
    if (rc != AERO_OK) return 1;
    const size_t cells = (size_t)width << log_n;
    uint64_t* trace = (uint64_t*)malloc(cells * 8);
    if (!trace || aero_fib_trace(width, log_n, trace) != AERO_OK) return 1;
    if (aero_host_register(trace, cells * 8) != AERO_OK) return 1;
    const aero_proof_options opt = {27, 8, 16, 4, 1, 8, 8};
    uint8_t* single = NULL;
    size_t single_len = 0;
    uint64_t* pub = (uint64_t*)calloc(width / 2 + 1, 8);
    rc = aero_prove_fib_air_host(ctx, trace, width, log_n, &air, &opt, &single, &single_len, pub);
    if (rc != AERO_OK) { fprintf(stderr, "single: %d %s\n", rc, aero_last_error(ctx)); return 1; }

    int32_t* devices = (int32_t*)calloc(world, sizeof(int32_t));          /* all ranks share GPU 0 */
    uint8_t** proofs = (uint8_t**)calloc(world, sizeof(uint8_t*));
    size_t* lens = (size_t*)calloc(world, sizeof(size_t));
    double* ms = (double*)calloc(world, sizeof(double));
    uint64_t* sent = (uint64_t*)calloc(world, sizeof(uint64_t));
    uint64_t* pub2 = (uint64_t*)calloc(width / 2 + 1, 8);
    char err[512] = {0};
    rc = aero_prove_fib_sharded_local(devices, world, trace, width, log_n, &air, &opt, 0, proofs, lens, pub2, ms, sent, err, sizeof err);
    if (rc != AERO_OK) { fprintf(stderr, "sharded: %d %s\n", rc, err); return 1; }
    int same = memcmp(pub, pub2, (width / 2) * 8) == 0;
    for (uint32_t r = 0; r < world; r++) same = same && lens[r] == single_len && memcmp(proofs[r], single, single_len) == 0;
    printf("{\"world\": %u, \"identical\": %s, \"proof_bytes\": %zu, \"rank0_ms\": %.3f, \"bytes_sent_rank0\": %llu}\n", world, same ? "true" : "false",
           single_len, ms[0], (unsigned long long)sent[0]);
    for (uint32_t r = 0; r < world; r++) aero_free(proofs[r]);
    aero_free(single);
    aero_host_unregister(trace);
    aero_ctx_destroy(ctx);
    return same ? 0 : 1;
}
