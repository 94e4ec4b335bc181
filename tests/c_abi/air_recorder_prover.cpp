// An AIR written in C++ (include/aero_air_builder.hpp), handed to the library as bytes, proven on the GPU and verified - no Python, no
// torch in this process. The AIR is version 2: a Fibonacci pair, a counter with a SEQUENCE assertion, an auxiliary running SUM (affine
// builder) and a general (squaring) auxiliary column. What a Rust `impl Prover` does after recording its `Air` once
// (constraints_worker.rs:32-43 -> aero_air_load; proving_worker.rs:465-467 -> aero_prove_air_host).
//   air_recorder_prover <log_n> <out.proof>     prints one JSON line; exit 2 = no device (the library has no CPU fallback)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/aero_air.h"
#include "../../include/aero_air_builder.hpp"

using aero_air_builder::Builder;
using aero_air_builder::Expr;
static const uint64_t P = aero_air_builder::P;
static uint64_t addp(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % P); }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: air_recorder_prover <log_n> <out.proof>\n"); return 1; }
    const int log_n = atoi(argv[1]);
    const size_t n = (size_t)1 << log_n;
    // ---- trace: columns 0,1 Fibonacci pair, 2 counter 5 + 3 i
    std::vector<uint64_t> trace(3 * n);
    uint64_t x = 1, y = 2;
    for (size_t i = 0; i < n; i++) {
        trace[i] = x; trace[n + i] = y; trace[2 * n + i] = (5 + 3 * (uint64_t)i) % P;
        const uint64_t nx = addp(x, y), ny = addp(y, nx);
        x = nx; y = ny;
    }
    const uint64_t pub[1] = {trace[n + n - 1]};
    // ---- the AIR
    Builder b(3, 2, 2, 1);
    Expr a = b.main(0), bb = b.main(1), na = b.main_next(0), nb = b.main_next(1);
    b.transition(na - (a + bb), 1);
    b.transition(nb - (bb + na), 1);
    b.transition(b.main_next(2) - b.main(2) - 3, 1);
    Expr term_den = b.rand(0) + b.main(2);
    b.aux_transition((b.aux_next(0) - b.aux(0)) * term_den - b.main(0), 2);                 // running sum of m0 / (r0 + m2)
    Expr sq = b.aux(1) * b.aux(1);
    b.aux_transition(b.aux_next(1) - (sq + b.rand(1) * b.aux(0) + b.main(1)), 2);           // squares its own previous value
    b.assert_single(0, 0, (uint64_t)1);
    b.assert_single(1, 0, (uint64_t)2);
    b.assert_single(1, -1, b.pub(0));
    std::vector<uint64_t> seq;
    for (size_t i = 0; i < n / 8; i++) seq.push_back(trace[2 * n + 3 + 8 * i]);
    b.assert_sequence(2, 3, 8, seq);                                                          // the counter at steps 3, 11, 19, ...
    b.aux_assert_single(0, 0, (uint64_t)0);
    b.aux_assert_single(1, 0, (uint64_t)5);
    b.aux_builder(0, b.constant(0), b.constant(1), Expr(), b.main(0), term_den);
    b.aux_builder_general(1, b.constant(5), sq + b.rand(1) * b.aux(0) + b.main(1));
    const std::vector<uint8_t> program = b.to_bytes();

    char err[512] = {0};
    aero_air* air = nullptr;
    if (aero_air_load(program.data(), program.size(), &air, err, sizeof err) != AERO_OK) { fprintf(stderr, "aero_air_load: %s\n", err); return 1; }
    aero_ctx* ctx = nullptr;
    if (aero_ctx_create(0, &ctx) != AERO_OK) { printf("no device: %s\n", aero_last_error(nullptr)); aero_air_free(air); return 2; }
    const aero_proof_options opt = {27, 8, 8, 4, 1, 4, 6};
    (void)aero_air_prepare(air, (uint32_t)log_n, &opt, 1);
    uint8_t* proof = nullptr;
    size_t len = 0;
    int32_t rc = aero_prove_air_host(ctx, air, trace.data(), (uint32_t)log_n, pub, 1, &opt, &proof, &len);
    if (rc != AERO_OK) { fprintf(stderr, "aero_prove_air_host: %s\n", aero_last_error(ctx)); return 1; }
    aero_verify_policy policy{};
    policy.expected_log_n = (uint32_t)log_n;
    rc = aero_verify_air(proof, len, pub, 1, air, &policy, err, sizeof err);
    const uint64_t wrong[1] = {pub[0] ^ 1};
    const int32_t rc_wrong = aero_verify_air(proof, len, wrong, 1, air, &policy, err, sizeof err);
    if (FILE* f = fopen(argv[2], "wb")) { fwrite(proof, 1, len, f); fclose(f); }
    printf("{\"program_bytes\": %zu, \"version\": %d, \"proof_bytes\": %zu, \"verified\": %s, \"wrong_statement_rejected\": %s}\n", program.size(), (int)program[7], len,
           rc == AERO_OK ? "true" : "false", rc_wrong != AERO_OK ? "true" : "false");
    aero_free(proof);
    aero_ctx_destroy(ctx);
    aero_air_free(air);
    return rc == AERO_OK && rc_wrong != AERO_OK ? 0 : 1;
}
