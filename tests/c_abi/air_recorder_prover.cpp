// An AIR written in C++ (include/aero_air_builder.hpp), handed to the library as bytes, proven on the GPU and verified - no Python, no
// torch in this process. The AIR is version 2: a Fibonacci pair, a counter with a SEQUENCE assertion, an auxiliary running SUM (affine
// builder) and a general (squaring) auxiliary column. What a Rust `impl Prover` does after recording its `Air` once
// (constraints_worker.rs:32-43 -> Air / aero_air_load; proving_worker.rs:465-467 -> AirProver::prove / aero_prove_air_host), written
// against the C++ host surface of include/aero_prover.hpp (Winterfell's names).
//   air_recorder_prover <log_n> <out.proof>     prints one JSON line; exit 2 = no device (the library has no CPU fallback)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/aero_air_builder.hpp"
#include "../../include/aero_prover.hpp"

using aero_air_builder::Builder;
using aero_air_builder::Expr;
using namespace aero_host;
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

    try {
        Air air(program);                                                   // aero_air_load: `ProcessorAir::new`
        Context ctx(0);
        const ProofOptions options(27, 8, 8, HashFunction::Blake2s_256, FieldExtension::None, 4, 64);
        air.prepare((uint32_t)log_n, options);
        TraceTable table(3, n);
        for (size_t c = 0; c < 3; c++) for (size_t i = 0; i < n; i++) table.set(c, i, trace[c * n + i]);
        AirProver prover(ctx, air, options, {pub[0]});
        const StarkProof proof = prover.prove(table);                       // `Prover::prove(trace)`
        AcceptableOptions acceptable;
        acceptable.min_query_security = 80;
        acceptable.expected_log_trace_length = (uint32_t)log_n;
        verify(proof, prover.get_pub_inputs(table), air, acceptable);       // `winter_verifier::verify::<AIR>`
        bool rejected = false;
        try { verify(proof, std::vector<uint64_t>{pub[0] ^ 1}, air, acceptable); } catch (const VerifierError&) { rejected = true; }
        const std::vector<uint8_t>& bytes = proof.to_bytes();
        if (FILE* f = fopen(argv[2], "wb")) { fwrite(bytes.data(), 1, bytes.size(), f); fclose(f); }
        printf("{\"program_bytes\": %zu, \"version\": %d, \"proof_bytes\": %zu, \"verified\": true, \"wrong_statement_rejected\": %s}\n", program.size(), (int)program[7],
               bytes.size(), rejected ? "true" : "false");
        return rejected ? 0 : 1;
    } catch (const ProverError& e) {
        if (e.status == AERO_E_HIP) { printf("no device: %s\n", e.what()); return 2; }
        fprintf(stderr, "ProverError(%d): %s\n", e.status, e.what());
        return 1;
    } catch (const VerifierError& e) {
        fprintf(stderr, "VerifierError(%d): %s\n", e.status, e.what());
        return 1;
    }
}
