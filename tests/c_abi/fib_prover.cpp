// The reference's `miden-proof-generator` flow (main.rs:20-51) against the C++ host surface of include/aero_prover.hpp:
// build the trace, Prover::prove, verify, write the bincode ProofData container - plus the error behaviour of the trait surface.
// usage: fib_prover <width> <log_n> <container_out>   -> one JSON line; exit 3 when no GPU context can be created
#include <cstdio>
#include <fstream>

#include "aero_prover.hpp"

using namespace aero_host;

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: fib_prover <width> <log_n> <container_out>\n"); return 2; }
    const size_t width = (size_t)atoi(argv[1]);
    const int log_n = atoi(argv[2]);
    try {
        Context ctx(0);
        const ProofOptions proof_security = ProofOptions::with_96_bit_security();
        FibProver prover(ctx, proof_security);
        const TraceTable trace = FibProver::build_trace(width, (size_t)1 << log_n);
        const FibPublicInputs pub_inputs = prover.get_pub_inputs(trace);

        // execute-and-prove, then what main.rs:46 does with miden_verifier::verify(..).unwrap()
        const StarkProof proof = prover.prove(trace);
        verify(proof, pub_inputs, prover.air());
        // the same with the verification INSIDE prove (proving_worker.rs:196-203): identical bytes, a rejection would be a ProverError
        ctx.set_self_verify(Context::SelfVerify::On);
        const StarkProof checked = prover.prove(trace);
        ctx.set_self_verify(Context::SelfVerify::Auto);
        if (checked.to_bytes() != proof.to_bytes()) { fprintf(stderr, "proof bytes change with the self check on\n"); return 1; }
        const uint32_t security = proof.security_level();

        // a wrong statement and an unacceptable parameter set are Err(VerifierError), not a panic
        int rejected = 0;
        try { FibPublicInputs wrong = pub_inputs; wrong.results[0] ^= 1; verify(proof, wrong, prover.air()); } catch (const VerifierError& e) { rejected += e.status == AERO_E_VERIFY; }
        try { AcceptableOptions strict; strict.min_conjectured_security = 128; verify(proof, pub_inputs, prover.air(), strict); } catch (const VerifierError& e) { rejected += e.status == AERO_E_VERIFY; }
        try { AcceptableOptions other; other.expected_log_trace_length = (uint32_t)log_n + 1; verify(proof, pub_inputs, prover.air(), other); } catch (const VerifierError& e) { rejected += e.status == AERO_E_VERIFY; }
        // a trace element outside the field is Err(ProverError) (BaseElement::new would have reduced it; this boundary takes raw u64)
        int prover_errors = 0;
        try { TraceTable bad = trace; bad.set(0, 3, MODULUS); prover.prove(bad); } catch (const ProverError& e) { prover_errors += e.status == AERO_E_BAD_ARG; }
        try { ProofOptions(27, 8, 16, HashFunction::Blake2s_256, FieldExtension::None, 8, 100); } catch (const ProverError& e) { prover_errors += e.status == AERO_E_BAD_ARG; }

        // ProofData { input_bytes, proof_bytes } -> bincode -> disk -> back (main.rs:36-51); input bytes here: the results, 8 bytes each
        ProofData data;
        for (uint64_t r : pub_inputs.results) for (int i = 0; i < 8; i++) data.input_bytes.push_back((uint8_t)(r >> (8 * i)));
        data.proof_bytes = proof.to_bytes();
        const std::vector<uint8_t> blob = data.serialize();
        { std::ofstream f(argv[3], std::ios::binary); f.write((const char*)blob.data(), (std::streamsize)blob.size()); }
        const ProofData back = ProofData::deserialize(blob);
        verify(StarkProof::from_bytes(back.proof_bytes), pub_inputs, prover.air());
        const bool round_trip = back.input_bytes == data.input_bytes && back.proof_bytes == data.proof_bytes;

        // the same statement proven twice gives the same bytes (the transcript is deterministic)
        const bool deterministic = prover.prove(trace).to_bytes() == proof.to_bytes();
        printf("{\"ok\": %s, \"proof_bytes\": %zu, \"security_level\": %u, \"rejected\": %d, \"prover_errors\": %d, \"results\": %zu}\n",
               (round_trip && deterministic && rejected == 3 && prover_errors == 2) ? "true" : "false", proof.to_bytes().size(), security, rejected, prover_errors,
               pub_inputs.results.size());
        return (round_trip && deterministic && rejected == 3 && prover_errors == 2) ? 0 : 1;
    } catch (const ProverError& e) {
        fprintf(stderr, "ProverError(%d): %s\n", e.status, e.what());
        return 3;
    } catch (const VerifierError& e) {
        fprintf(stderr, "VerifierError(%d): %s\n", e.status, e.what());
        return 4;
    }
}
