// Argument blocks and launchers of the AIR-program kernels (air_kernels.hip): the device interpreter of a compiled AEROAIR
// program (air_program.hpp) over the constraint-evaluation domain, the auxiliary-segment builder and the generic division by the
// constraint divisors.
#pragma once
#include "aero_internal.hpp"
#include "air_program.hpp"

namespace aero {

// boundary divisor group as the kernels see it: divisor x^a - b; numerator multiplied by (alpha + beta x^adj)
struct AirBGroupDev {
    uint64_t a_exp;      // a mod rows: x^a = ha * w_rows^(s * a_exp mod rows)
    uint64_t ha;         // offset^a
    uint64_t b;
    uint64_t adj_exp;    // adj mod rows (offset^adj is folded into the beta coefficients)
    uint64_t seq;        // 0, or 1 + index of the group's sequence-assertion table in AirConsArgs::seq_tab (DEG columns of `rows` entries each)
};

// LOAD descriptor on the device (one u64): bits 0-31 column index (main / aux) or offset into the periodic tables, bits 32-43
// register slot, bits 44-49 log2 of the periodic table's length, bit 62 periodic, bit 63 next row.
inline uint64_t air_dev_desc(bool next, bool periodic, uint32_t slot, uint32_t col_or_off, uint32_t log_period) {
    return (uint64_t)col_or_off | ((uint64_t)slot << 32) | ((uint64_t)log_period << 44) | ((uint64_t)(periodic ? 1 : 0) << 62) | ((uint64_t)(next ? 1 : 0) << 63);
}

template <class F> struct AirConsArgs {
    typedef typename F::T T;
    // frame source: trace LDE (W columns) and auxiliary LDE (A * DEG component columns) with `N` rows each; evaluation row s sits at
    // matrix row s * ce_step, its successor `blowup` rows further (mod N); split_log as in FibConsArgs (de-interleaved compact copy)
    const uint64_t* lde;
    const uint64_t* aux;
    size_t N;
    uint32_t W, A, blowup, ce_step, split_log;
    size_t rows;                 // size of the evaluation domain (the points offset * w_rows^s)
    size_t first, count;         // rows [first, first + count) are evaluated by this launch
    // program
    const air::Insn* code;       // patched for this proof: poff = position of the instruction's scalars in `pool`; BOUND: dst = group
    const uint64_t* pool;        // scalar operands | coefficient pairs (alpha, beta') per constraint / assertion | LOAD descriptors
    uint32_t slotsB, slotsE;
    const uint64_t* ptab;        // periodic tables, concatenated
    const uint64_t* dg_exp;      // per degree group: adj mod rows (offset^adj is folded into beta')
    const AirBGroupDev* bgroups;
    uint32_t n_bgroups;
    const T *gA, *gB;            // per group: sum alpha_m value_m, sum beta'_m value_m (the assertions' constant parts)
    const uint64_t* seq_tab;     // row-dependent parts (sequence assertions): table t, component d at seq_tab[((t * DEG) + d) * rows + s]
    // domain
    const uint64_t *tw_lo, *tw_hi;   // two-level table of w_rows
    int tw_h;
    uint64_t offset;
    // fused division (MODE 1)
    const uint64_t* zn_inv;      // (xmask + 1) entries: (x^n - 1)^-1 for the distinct values of x^n, indexed by s & xmask
    uint32_t xmask;
    const uint64_t* exempt;      // n_exempt points w_n^(n-i): the transition divisor is (x^n - 1) / prod (x - exempt[i])
    uint32_t n_exempt;
    uint64_t* out_cols;          // MODE 0: ((1 + n_bgroups) * DEG) x count, column-major
    uint64_t* out_h[2];          // MODE 1: DEG component arrays indexed by s
};
// mode 0: numerator columns; mode 1: divided and summed (H). Returns false when mode 1 cannot run fused (register file too large
// for the LDS): the caller then evaluates mode 0 and divides with launch_air_divide.
template <class F> bool launch_air_constraints(Context* ctx, const AirConsArgs<F>& a, int mode);

// The same evaluation by a kernel generated from the program and compiled at run time (air_jit.hip). Returns false when the context
// runs with the interpreter (AERO_AIR_JIT=0) or the kernel could not be built (air_jit_last_error() says why): the caller then
// interprets. pdesc: per periodic column offset | mask << 32 into ptab; oSE / oT / oB: layout of `pool` (air_host.hip: build_pool).
template <class F> bool launch_air_jit(Context* ctx, const air::Program& p, const air::Instance& in, const AirConsArgs<F>& a, const uint64_t* pdesc, uint64_t oSE,
                                       uint64_t oT, uint64_t oB, int mode);
std::string air_jit_last_error();
std::string air_jit_source(const air::Program& p, const air::Instance& in, int deg, int mode);     // the generated HIP source (diagnosis, tests)
bool air_jit_prepare(const air::Program& p, int log_n, int deg, size_t rows_per_launch, std::string* err);   // the kernel a proof will use, ahead of it
bool air_jit_compile_only(const air::Program& p, const air::Instance& in, int deg, int mode, std::string* err);   // no device needed

// H = sum_j column_j / divisor_j over the evaluation domain (the unfused `ConstraintEvaluationTable::into_poly` division)
template <class F> struct AirDivideArgs {
    const uint64_t* cols;        // ((1 + n_bgroups) * DEG) x rows numerators, column-major
    size_t rows;
    const AirBGroupDev* bgroups;
    uint32_t n_bgroups;
    const uint64_t *tw_lo, *tw_hi;
    int tw_h;
    uint64_t offset;
    const uint64_t* zn_inv;
    uint32_t xmask;
    const uint64_t* exempt;
    uint32_t n_exempt;
    uint64_t* out_h[2];
};
template <class F> void launch_air_divide(Context* ctx, const AirDivideArgs<F>& a);
// dst[pos[i]] = val[i]
void launch_air_scatter(Context* ctx, uint64_t* dst, const uint64_t* pos, const uint64_t* val, size_t count);

// Auxiliary segment from the program's builders: column c(0) = init_c, c(i+1) = c(i) * num_c(i) / den_c(i) [+ add_c(i) / add_den_c(i)].
template <class F> struct AirAuxArgs {
    typedef typename F::T T;
    const uint64_t* trace;       // W x n main segment
    size_t n;
    uint32_t W, A;
    const air::Insn* code;       // patched: poff
    const uint64_t* pool;
    uint32_t slotsB, slotsE;
    const uint64_t* ptab;
    const uint8_t* has_den;      // per aux column (device)
    const uint8_t* has_add;      // per aux column (device): 0 product only, 1 additive term, 3 additive term with denominator (affine builders)
    const T* init;               // per aux column (device)
    uint64_t* out;               // (A * DEG) x n component columns
};
template <class F> void launch_air_aux(Context* ctx, const AirAuxArgs<F>& a, const std::vector<uint8_t>& has_den_host, const std::vector<uint8_t>& has_add_host);

// General auxiliary recurrence of ONE column on the device (round 5; the reference builds its auxiliary columns inside
// commit_to_trace_and_validate, aero-sdk/miden-wasm/src/proving_worker.rs:323-332): column(i + 1) = expr(main row i, main row i + 1,
// auxiliary row i of the columns up to its own, periodic values, scalars). Nothing about it can be scanned; what CAN run in parallel is
// every sub-expression that does not read the column itself. One wavefront walks the column in blocks of 64 rows:
//   loads     lane l fetches the operands of row base + l (coalesced) into LDS slots,
//   parallel  lane l evaluates the nodes that do not depend on the column, for its row, into LDS slots,
//   serial    the wave as a whole (uniform control flow) walks the 64 rows: the dependent nodes read the row's slots as LDS broadcasts, the
//             column's current value from a register and each other's results through lane registers (value of serial node s lives in lane s:
//             v_readlane / a select, no LDS round trip on the chain); lane r keeps the value of row base + r + 1 and the block is stored coalesced.
// Operand kinds of an instruction: slot (LDS, per row), serial (lane register), x (the column's current value), constant (pool).
enum : uint32_t { GOP_SLOT = 0, GOP_SER = 1, GOP_X = 2, GOP_CONST = 3 };
enum : uint32_t { GLD_MAIN_CUR = 0, GLD_MAIN_NXT = 1, GLD_AUX_CUR = 2, GLD_PERIODIC = 3 };
struct GenInsn { uint32_t op, dst, ka, ia, kb, ib; };       // op: 1 add, 2 sub, 3 mul; dst: slot (parallel section) or serial lane (serial section)
struct GenLoad { uint32_t kind, col, slot, mask; };          // periodic: col = offset into ptab, mask = cycle length - 1
constexpr uint32_t GEN_MAX_SLOTS = 40, GEN_MAX_SERIAL = 64;
template <class F> struct AirGeneralArgs {
    typedef typename F::T T;
    const uint64_t* trace;       // W x n main segment
    uint64_t* aux;               // (A * DEG) x n component columns: earlier columns are read, column `col` is written
    size_t n;
    uint32_t col;
    const GenLoad* loads; uint32_t n_loads;
    const GenInsn* par; uint32_t n_par;
    const GenInsn* ser; uint32_t n_ser;
    uint32_t res_kind, res_idx;  // operand that holds the next value once the serial section has run
    const T* consts;
    const uint64_t* ptab;
    T init;
};
template <class F> void launch_air_general_column(Context* ctx, const AirGeneralArgs<F>& a);

}  // namespace aero
