// Host side of the boundary in C++: the surface the reference programs against, over the C ABI of aero_stark.h.
//
// The reference's host code is Rust against Winterfell 0.4 (miden-proof-generator/src/main.rs:20-51; aero-sdk/miden-wasm/src/
// proving_worker.rs:14,69,165-169,239-268,465-467): a `Prover` trait (`get_pub_inputs(&trace)`, `options()`, `prove(trace) ->
// Result<StarkProof, ProverError>`), `ProofOptions::new / with_96_bit_security`, a `TraceTable` filled row by row,
// `StarkProof::to_bytes / from_bytes`, `verify(..) -> Result<(), VerifierError>` and the bincode `ProofData` container. No Rust
// toolchain exists here, so this header carries the same names, argument meaning and error behaviour in C++ (header only, C++17;
// link with -laero_stark). A Rust `impl winter_prover::Prover` would forward exactly like `GpuProver::prove` below
// (INTEGRATION.md section 3). Nothing in here computes: every call ends in the C ABI, and without a GPU `Context` construction
// throws (there is no CPU fallback).
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "aero_air.h"
#include "aero_stark.h"

namespace aero_host {

// ---- errors: Result<_, ProverError> / Result<_, VerifierError> become exceptions carrying the status of the C ABI ----------------
struct ProverError : std::runtime_error {
    int32_t status;
    ProverError(int32_t st, const std::string& what) : std::runtime_error(what), status(st) {}
};
struct VerifierError : std::runtime_error {
    int32_t status;
    VerifierError(int32_t st, const std::string& what) : std::runtime_error(what), status(st) {}
};

constexpr uint64_t MODULUS = 0xFFFFFFFF00000001ull;      // Goldilocks, BaseElement::MODULUS

enum class HashFunction : uint8_t { Blake2s_256 = 4 };   // ids of the proof header
enum class FieldExtension : uint8_t { None = 1, Quadratic = 2 };

// winter_air::ProofOptions (convert_inputs.rs:54-66)
class ProofOptions {
public:
    ProofOptions(uint8_t num_queries, uint8_t blowup_factor, uint8_t grinding_factor, HashFunction hash_fn, FieldExtension field_extension,
                 uint8_t fri_folding_factor, uint32_t fri_max_remainder_size) {
        uint8_t log = 0;
        while ((1u << log) < fri_max_remainder_size) log++;
        if ((1u << log) != fri_max_remainder_size) throw ProverError(AERO_E_BAD_ARG, "ProofOptions: fri_max_remainder_size must be a power of two");
        o_ = aero_proof_options{num_queries, blowup_factor, grinding_factor, (uint8_t)hash_fn, (uint8_t)field_extension, fri_folding_factor, log};
    }
    static ProofOptions with_96_bit_security() { return ProofOptions(27, 8, 16, HashFunction::Blake2s_256, FieldExtension::None, 8, 256); }
    uint8_t num_queries() const { return o_.num_queries; }
    uint8_t blowup_factor() const { return o_.blowup_factor; }
    uint8_t grinding_factor() const { return o_.grinding_factor; }
    FieldExtension field_extension() const { return (FieldExtension)o_.field_extension; }
    uint8_t fri_folding_factor() const { return o_.fri_folding_factor; }
    uint32_t fri_max_remainder_size() const { return 1u << o_.fri_log_max_remainder; }
    const aero_proof_options& raw() const { return o_; }

private:
    aero_proof_options o_;
};

// winter_prover::TraceTable: `width` columns of `length` base-field elements, column-major in host memory
class TraceTable {
public:
    TraceTable(size_t width, size_t length) : width_(width), length_(length), data_(width * length, 0) {
        if (width == 0 || length < 8 || (length & (length - 1))) throw ProverError(AERO_E_BAD_ARG, "TraceTable: length must be a power of two >= 8");
    }
    size_t width() const { return width_; }
    size_t length() const { return length_; }
    uint32_t log_length() const { uint32_t l = 0; while (((size_t)1 << l) < length_) l++; return l; }
    uint64_t get(size_t column, size_t step) const { return data_[column * length_ + step]; }
    void set(size_t column, size_t step, uint64_t value) { data_[column * length_ + step] = value; }
    const uint64_t* get_column(size_t column) const { return data_.data() + column * length_; }
    const uint64_t* data() const { return data_.data(); }
    // TraceTable::fill(init, update): `init` writes row 0, `update(step, state)` turns row `step` into row `step + 1`
    void fill(const std::function<void(std::vector<uint64_t>&)>& init, const std::function<void(size_t, std::vector<uint64_t>&)>& update) {
        std::vector<uint64_t> state(width_, 0);
        init(state);
        for (size_t c = 0; c < width_; c++) set(c, 0, state[c]);
        for (size_t step = 0; step + 1 < length_; step++) {
            update(step, state);
            for (size_t c = 0; c < width_; c++) set(c, step + 1, state[c]);
        }
    }

private:
    size_t width_, length_;
    std::vector<uint64_t> data_;
};

// winter_air::proof::StarkProof as its byte form (the layout IS the interface: src/stark_verifier reads these bytes)
class StarkProof {
public:
    StarkProof() = default;
    static StarkProof from_bytes(std::vector<uint8_t> bytes) { StarkProof p; p.bytes_ = std::move(bytes); return p; }
    const std::vector<uint8_t>& to_bytes() const { return bytes_; }
    // Winterfell's conjectured security estimate from the proof's own parameters: min(query term, field term)
    uint32_t security_level() const {
        uint32_t q = 0, f = 0;
        if (aero_proof_security_bits(bytes_.data(), bytes_.size(), &q, &f) != AERO_OK) throw VerifierError(AERO_E_VERIFY, "StarkProof: malformed proof bytes");
        return q < f ? q : f;
    }

private:
    std::vector<uint8_t> bytes_;
};

// miden_proof_generator::ProofData (lib.rs:1-6): bincode of the two byte vectors
struct ProofData {
    std::vector<uint8_t> input_bytes, proof_bytes;
    std::vector<uint8_t> serialize() const {
        uint8_t* out = nullptr;
        size_t n = 0;
        if (aero_proof_container(input_bytes.data(), input_bytes.size(), proof_bytes.data(), proof_bytes.size(), &out, &n) != AERO_OK)
            throw ProverError(AERO_E_OOM, "ProofData: could not serialise");
        std::vector<uint8_t> v(out, out + n);
        aero_free(out);
        return v;
    }
    static ProofData deserialize(const std::vector<uint8_t>& b) {
        auto u64 = [&](size_t o) { if (b.size() < o + 8) throw VerifierError(AERO_E_VERIFY, "ProofData: truncated"); uint64_t v; memcpy(&v, b.data() + o, 8); return v; };
        ProofData d;
        const uint64_t n = u64(0);
        if (n > b.size() - 8) throw VerifierError(AERO_E_VERIFY, "ProofData: truncated");
        d.input_bytes.assign(b.begin() + 8, b.begin() + 8 + n);
        const uint64_t m = u64(8 + n);
        if (m != b.size() - 16 - n) throw VerifierError(AERO_E_VERIFY, "ProofData: length mismatch");
        d.proof_bytes.assign(b.begin() + 16 + n, b.end());
        return d;
    }
};

// one GPU + one stream + a memory pool (aero_ctx); not thread-safe, movable
class Context {
public:
    explicit Context(int32_t device_id = 0) {
        const int32_t rc = aero_ctx_create(device_id, &h_);
        if (rc != AERO_OK) throw ProverError(rc, std::string("Context: ") + aero_last_error(nullptr));
    }
    ~Context() { if (h_) aero_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    Context(Context&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    aero_ctx* raw() const { return h_; }
    // prove-then-verify inside `prove`, as the reference's worker does (proving_worker.rs:196-203; the CLI verifies right behind `prove`,
    // main.rs:47): with the check on, a proof the library's own verifier rejects surfaces as ProverError{AERO_E_SELF_VERIFY} instead of
    // a StarkProof. Default: on for proofs made by more than one rank, off on one GPU (aero_ctx_set_self_verify).
    enum class SelfVerify : int32_t { Auto = AERO_SELF_VERIFY_AUTO, Off = AERO_SELF_VERIFY_OFF, On = AERO_SELF_VERIFY_ON };
    void set_self_verify(SelfVerify mode) {
        const int32_t rc = aero_ctx_set_self_verify(h_, (int32_t)mode);
        if (rc != AERO_OK) throw ProverError(rc, "Context: bad self-verify mode");
    }

private:
    aero_ctx* h_ = nullptr;
};

// The Prover trait (proving_worker.rs:14,69,165-169): an implementation names its public inputs and options; `prove` is the
// provided method and runs on the GPU.
template <class PublicInputs> class Prover {
public:
    virtual ~Prover() = default;
    virtual PublicInputs get_pub_inputs(const TraceTable& trace) const = 0;
    virtual const ProofOptions& options() const = 0;
    virtual StarkProof prove(const TraceTable& trace) const = 0;
};

// The built-in AIR behind this boundary: FibAir(width) - column pair k = (a, b), a' = a + b, b' = b + a', assertions a(0), b(0),
// b(n - 1); public inputs = the width / 2 final results. `air` adds the stand-in auxiliary segment.
struct FibPublicInputs {
    std::vector<uint64_t> results;
};
class FibProver : public Prover<FibPublicInputs> {
public:
    FibProver(Context& ctx, ProofOptions options, aero_fib_air air = aero_fib_air{0, 0, 0}) : ctx_(ctx), options_(options), air_(air) {}
    // winterfell's fib2 example trace, generalised to `width / 2` independent pairs with seeds (1 + 2k, 2 + 2k)
    static TraceTable build_trace(size_t width, size_t length) {
        if (width < 2 || (width & 1)) throw ProverError(AERO_E_BAD_ARG, "FibProver: the trace width must be even");
        TraceTable t(width, length);
        auto addm = [](uint64_t a, uint64_t b) { const unsigned __int128 s = (unsigned __int128)a + b; return (uint64_t)(s >= MODULUS ? s - MODULUS : s); };
        t.fill([&](std::vector<uint64_t>& s) { for (size_t k = 0; k < width / 2; k++) { s[2 * k] = 1 + 2 * k; s[2 * k + 1] = 2 + 2 * k; } },
               [&](size_t, std::vector<uint64_t>& s) { for (size_t k = 0; k < width / 2; k++) { s[2 * k] = addm(s[2 * k], s[2 * k + 1]); s[2 * k + 1] = addm(s[2 * k + 1], s[2 * k]); } });
        return t;
    }
    FibPublicInputs get_pub_inputs(const TraceTable& trace) const override {
        FibPublicInputs p;
        for (size_t k = 0; k < trace.width() / 2; k++) p.results.push_back(trace.get(2 * k + 1, trace.length() - 1));
        return p;
    }
    const ProofOptions& options() const override { return options_; }
    // Prover::prove(trace): the trace is handed over in host memory, the proof bytes come back in host memory
    StarkProof prove(const TraceTable& trace) const override {
        uint8_t* out = nullptr;
        size_t n = 0;
        std::vector<uint64_t> pub(trace.width() / 2 + 1);
        const int32_t rc = aero_prove_fib_air_host(ctx_.raw(), trace.data(), (uint32_t)trace.width(), trace.log_length(), air_.aux_width ? &air_ : nullptr,
                                                   &options_.raw(), &out, &n, pub.data());
        if (rc != AERO_OK) throw ProverError(rc, aero_last_error(ctx_.raw()));
        std::vector<uint8_t> bytes(out, out + n);
        aero_free(out);
        return StarkProof::from_bytes(std::move(bytes));
    }
    const aero_fib_air& air() const { return air_; }

private:
    Context& ctx_;
    ProofOptions options_;
    aero_fib_air air_;
};

// What the verifier demands of a proof's self-declared parameters (upstream later added winter_verifier::AcceptableOptions):
//   min_query_security ......... floor on num_queries * log2(blowup) + grinding (the soundness of the query phase)
//   min_conjectured_security ... floor on min(query term, 64 * extension degree - log2(LDE domain)) - the quantity
//                                StarkProof::security_level() reports and AcceptableOptions::MinConjecturedSecurity means;
//                                0 = not required. A base-field proof of a 2^20-row trace has 41 bits of it.
struct AcceptableOptions {
    uint32_t min_query_security = 96;
    uint32_t min_conjectured_security = 0;
    uint32_t expected_log_trace_length = 0;      // 0 = any
};
// winter_verifier::verify::<FibAir>(proof, pub_inputs, &acceptable_options): host code, throws VerifierError when rejected
inline void verify(const StarkProof& proof, const FibPublicInputs& pub_inputs, const aero_fib_air& air, const AcceptableOptions& acceptable = AcceptableOptions()) {
    aero_verify_policy policy{};
    policy.min_query_security_bits = acceptable.min_query_security;
    policy.min_conjectured_security_bits = acceptable.min_conjectured_security;
    policy.expected_log_n = acceptable.expected_log_trace_length;
    char err[512] = {0};
    const std::vector<uint8_t>& b = proof.to_bytes();
    const int32_t rc = aero_verify_fib(b.data(), b.size(), pub_inputs.results.data(), (uint32_t)pub_inputs.results.size(), &air, &policy, err, sizeof err);
    if (rc != AERO_OK) throw VerifierError(rc, err);
}

// ---- any AIR, handed over as a constraint program (include/aero_air.h; recorded e.g. with include/aero_air_builder.hpp) ------------
// `Air` = what `ProcessorAir::new(trace_info, pub_inputs, options)` gives the reference (constraints_worker.rs:32-36): loaded once,
// immutable, shareable between contexts and threads.
class Air {
public:
    explicit Air(const std::vector<uint8_t>& program) {
        char err[512] = {0};
        const int32_t rc = aero_air_load(program.data(), program.size(), &h_, err, sizeof err);
        if (rc != AERO_OK) throw ProverError(rc, std::string("Air: ") + err);
    }
    ~Air() { if (h_) aero_air_free(h_); }
    Air(const Air&) = delete;
    Air& operator=(const Air&) = delete;
    Air(Air&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    const aero_air* raw() const { return h_; }
    uint32_t main_width() const { uint32_t v[16]; aero_air_info(h_, v); return v[0]; }
    uint32_t num_pub_inputs() const { uint32_t v[16]; aero_air_info(h_, v); return v[3]; }
    // build the evaluation kernel a proof of 2^log_length rows will use, ahead of the first proof (aero_air_prepare)
    void prepare(uint32_t log_length, const ProofOptions& options) const { (void)aero_air_prepare(h_, log_length, &options.raw(), 1); }

private:
    aero_air* h_ = nullptr;
};
// The Prover trait for a program AIR: the public inputs are the statement's elements in the order the program's PUB operands number
// them; `prove` is `Prover::prove(trace)` + `to_bytes()` on the GPU (proving_worker.rs:465-467 with the generic `Air`).
class AirProver : public Prover<std::vector<uint64_t>> {
public:
    AirProver(Context& ctx, const Air& air, ProofOptions options, std::vector<uint64_t> pub_inputs)
        : ctx_(ctx), air_(air), options_(options), pub_(std::move(pub_inputs)) {}
    std::vector<uint64_t> get_pub_inputs(const TraceTable&) const override { return pub_; }
    const ProofOptions& options() const override { return options_; }
    StarkProof prove(const TraceTable& trace) const override {
        if (trace.width() != air_.main_width()) throw ProverError(AERO_E_BAD_ARG, "AirProver: the trace width is not the program's main width");
        uint8_t* out = nullptr;
        size_t n = 0;
        const int32_t rc = aero_prove_air_host(ctx_.raw(), air_.raw(), trace.data(), trace.log_length(), pub_.data(), (uint32_t)pub_.size(), &options_.raw(), &out, &n);
        if (rc != AERO_OK) throw ProverError(rc, aero_last_error(ctx_.raw()));
        std::vector<uint8_t> bytes(out, out + n);
        aero_free(out);
        return StarkProof::from_bytes(std::move(bytes));
    }

private:
    Context& ctx_;
    const Air& air_;
    ProofOptions options_;
    std::vector<uint64_t> pub_;
};
// winter_verifier::verify::<AIR>(proof, pub_inputs, &acceptable_options) for a program AIR (with the out-of-domain constraint check)
inline void verify(const StarkProof& proof, const std::vector<uint64_t>& pub_inputs, const Air& air, const AcceptableOptions& acceptable = AcceptableOptions()) {
    aero_verify_policy policy{};
    policy.min_query_security_bits = acceptable.min_query_security;
    policy.min_conjectured_security_bits = acceptable.min_conjectured_security;
    policy.expected_log_n = acceptable.expected_log_trace_length;
    char err[512] = {0};
    const std::vector<uint8_t>& b = proof.to_bytes();
    const int32_t rc = aero_verify_air(b.data(), b.size(), pub_inputs.data(), (uint32_t)pub_inputs.size(), air.raw(), &policy, err, sizeof err);
    if (rc != AERO_OK) throw VerifierError(rc, err);
}

}  // namespace aero_host
