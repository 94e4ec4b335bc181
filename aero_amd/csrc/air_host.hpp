// Host side of a program AIR inside a proof: builds the per-proof device tables of a compiled AEROAIR program (scalars, periodic
// tables on the evaluation coset, degree-adjustment exponents, boundary divisors) and launches the interpreter kernels.
// Reference seam: `ConstraintEvaluator::new(&air, aux_rand_elements, &constraint_coeffs)` + `evaluate_fragment`
// (aero-sdk/miden-wasm/src/constraints_worker.rs:38-59), `build_aux_segment` (proving_worker.rs:323-332).
#pragma once
#include "air_kernels.hpp"

namespace aero {

// Several small host arrays -> ONE pinned staging block -> ONE async H2D copy; pointers are handed out afterwards.
struct ParamPack {
    Context* ctx;
    struct Item { const void* src; size_t bytes, off; };
    std::vector<Item> items;
    size_t total = 0;
    uint8_t* dev = nullptr;
    explicit ParamPack(Context* c) : ctx(c) {}
    template <class T> size_t add(const std::vector<T>& v) {
        Item it{v.data(), v.size() * sizeof(T), total};
        items.push_back(it);
        total += (it.bytes + 15) & ~(size_t)15;
        return items.size() - 1;
    }
    void commit() {
        dev = (uint8_t*)ctx->scratch_alloc(total + 16);
        constexpr size_t PIECE = (size_t)2 << 20;      // the context's pinned staging block is 8 MiB and is shared with every other small copy
        if (total + 16 <= PIECE) {
            uint8_t* host = (uint8_t*)ctx->stage_alloc(total + 16);
            for (auto& it : items) if (it.bytes) memcpy(host + it.off, it.src, it.bytes);
            AERO_HIP(hipMemcpyAsync(dev, host, total + 16, hipMemcpyHostToDevice, ctx->stream));
            return;
        }
        // a large pack (the raw values of long sequence assertions: 2^20 and more entries) travels piece by piece through the staging block
        // (stage_alloc waits for the earlier copies when the block wraps) instead of failing with "staging request too large"
        for (auto& it : items)
            for (size_t off = 0; off < it.bytes; off += PIECE) {
                const size_t len = it.bytes - off < PIECE ? it.bytes - off : PIECE;
                uint8_t* host = (uint8_t*)ctx->stage_alloc(len);
                memcpy(host, static_cast<const uint8_t*>(it.src) + off, len);
                AERO_HIP(hipMemcpyAsync(dev + it.off + off, host, len, hipMemcpyHostToDevice, ctx->stream));
            }
    }
    template <class T> const T* ptr(size_t id) const { return reinterpret_cast<const T*>(dev + items[id].off); }
};

// Where the frames of the evaluation domain live: `rows` points x_s = offset * w_rows^s; row s of that domain is matrix row
// s * ce_step of `lde` / `aux` (frame_rows rows each, successor = + frame_rows / n), de-interleaved by split_log.
struct AirGeometry {
    const uint64_t* lde = nullptr;
    const uint64_t* aux = nullptr;
    size_t frame_rows = 0;
    uint32_t split_log = 0;
    size_t rows = 0;           // evaluation domain size (a multiple of the trace length)
    size_t first = 0, count = 0;
    uint64_t offset = gl::GEN;
};
template <class F> struct AirCoeffs {     // composition coefficients in draw order
    std::vector<typename F::T> ta, tb, ba, bb;
};
// mode 0: numerator columns -> out_cols ((1 + groups) * DEG x count); mode 1: H -> out_h[d] (indexed by evaluation row)
template <class F>
void air_eval_constraints(Context* ctx, const air::Program& p, const air::Instance& in, const AirGeometry& g, const AirCoeffs<F>& cc,
                          const uint64_t* pub, const typename F::T* rands, int mode, uint64_t* out_cols, uint64_t* const out_h[2]);
// H = sum_j column_j / divisor_j for host-supplied numerator columns already on the device ((1 + groups) * DEG x rows)
template <class F>
void air_divide_columns(Context* ctx, const air::Program& p, const air::Instance& in, const uint64_t* cols_dev, size_t rows, uint64_t offset,
                        uint64_t* const out_h[2]);
// (A * DEG) x n component columns of the auxiliary segment
template <class F>
void air_build_aux(Context* ctx, const air::Program& p, const uint64_t* trace_dev, int log_n, const uint64_t* pub, const typename F::T* rands, uint64_t* out);

// `Trace::validate(&air)` on the device (run-time compiled kernel, air_jit.hip mode 2): trace = W x n main segment, aux = (A * DEG) x n
// component columns or null (then only the main constraints and assertions are checked). Returns ~0 when every constraint holds, else
// row << 24 | id of the first failure (id = transition index in the program's order, or 0x800000 | assertion index, main first).
template <class F>
uint64_t air_validate_trace(Context* ctx, const air::Program& p, const air::Instance& in, const uint64_t* trace_dev, const uint64_t* aux_dev,
                            const uint64_t* pub, const typename F::T* rands);

}  // namespace aero
