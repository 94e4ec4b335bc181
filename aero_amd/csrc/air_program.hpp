// AIR-as-data, host side: parse + validate an AEROAIR program (include/aero_air.h), compile it for the device interpreter
// (air_kernels.hip), derive everything Winterfell's `Air` / `ConstraintEvaluator` derive from a trace length, and run the compiled
// program on the host over E for the verifier's out-of-domain check.
//
// Reference seam: the `Air` object behind `ProcessorAir::new(trace_info, public_inputs, proof_options)` ->
// `ConstraintEvaluator::new(&air, aux_rand_elements, &constraint_coeffs)` -> `evaluate_fragment`
// (/root/reference/aero-sdk/miden-wasm/src/constraints_worker.rs:32-59, proving_worker.rs:255-259,374-395); the grouping /
// degree-adjustment rules of winter-air 0.4 are restated from their verifier-side mirror in
// src/stark_verifier/air/transitions/evaluator.cairo:79-86,131-150,216-218.
//
// Compilation (at load time, independent of the trace length):
//   1. type inference: a node is E-valued iff it depends on the auxiliary frame or a random element; everything else stays in the
//      base field (with the quadratic extension that is a third of the multiplications);
//   2. constant folding: nodes that depend on no frame value and no periodic column (constants, public inputs, random elements)
//      are evaluated once per proof on the host and become scalar operands;
//   3. transition constraints are grouped by their declared degree (the key of Winterfell's degree groups); code is emitted group
//      by group, each constraint's expression tree in post-order with common sub-expressions computed once, followed by an EMIT
//      that folds the value into the group's two accumulators (sum alpha_k t_k and sum beta_k t_k); GROUP_END multiplies the
//      second by x^adj - ONE power per group and row, obtained as a twiddle-table lookup, not an exponentiation;
//   4. lowering: every frame access becomes explicit - LOAD instructions fetch 8 frame values (4 of the auxiliary segment) at a time,
//      looking ahead in the stream; the assertions on a column are bound right behind the load of its current-row value; the last one
//      or two nodes of a constraint are folded into its EMIT (EMIT_ADD/SUB/MUL_B, EMIT3_B); a value whose next use is far is re-loaded;
//   5. register allocation: a linear scan over the lowered stream assigns every live value a slot of the per-lane register file
//      (LDS on the device), reusing slots after the last use.
// The SAME parsed program (nodes, types, folded scalars, degree and boundary groups) also feeds the run-time code generator of
// air_jit.hip, which is the default evaluator on the device; the instruction stream above is what the interpreter (air_kernels.hip)
// and the host evaluation of the verifier (`host_evaluate`) run.
#pragma once
#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <tuple>
#include <queue>
#include <string>
#include <vector>

#include "aero_internal.hpp"

#include <memory>

namespace aero {
namespace air {

enum Kind : uint32_t { K_NODE = 0, K_MAIN_CUR, K_MAIN_NXT, K_AUX_CUR, K_AUX_NXT, K_PERIODIC, K_CONST, K_PUB, K_RAND, K_COUNT, K_SEQ = K_COUNT };   // K_SEQ: value of a sequence assertion only
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;
constexpr uint32_t REF_GENERAL = 0xFFFFFFFEu;    // aux builder, `den` field: column(i+1) = num evaluated on (main row i, main row i+1, aux row i): any recurrence, built on the host
inline uint32_t ref_kind(uint32_t r) { return r >> 24; }
inline uint32_t ref_index(uint32_t r) { return r & 0xFFFFFFu; }
inline uint32_t mk_ref(uint32_t k, uint32_t i) { return (k << 24) | i; }

// ---- the device program ------------------------------------------------------------------------------------------------------
// Operand kinds of the emitted code. Frame values (D_MAIN_* / D_AUX_*) appear only in the descriptors of the LOAD instructions:
// arithmetic reads registers, scalars and periodic tables.
enum DKind : uint32_t { D_SLOT_B = 0, D_SLOT_E, D_MAIN_CUR, D_MAIN_NXT, D_AUX_CUR, D_AUX_NXT, D_PERIODIC, D_SCAL_B, D_SCAL_E };
enum DOp : uint32_t {
    OP_END = 0,
    OP_ADD_B, OP_SUB_B, OP_MUL_B,            // base x base -> base slot
    OP_ADD_E, OP_SUB_E, OP_MUL_E,            // operands of either type (base values are lifted) -> E slot
    OP_MULB_E,                               // a in E, b in the base field -> E slot
    OP_EMIT_B, OP_EMIT_E,                    // transition constraint `dst`: value a (base / E) into the group accumulators
    OP_GROUP_END,                            // degree group `dst` complete: total += acc_beta * x^adj(dst)
    OP_OUT_B, OP_OUT_E,                      // aux builder: value a -> output column dst
    OP_LOAD_MAIN,                            // LOAD_WIDTH frame values of the main segment -> base slots; dst = first descriptor
    OP_LOAD_AUX,                             // LOAD_WIDTH / 2 frame values of the auxiliary segment -> E slots
    OP_BOUND_B, OP_BOUND_E,                  // assertion `dst` on the (current-row) value a: into its divisor group's accumulators
    OP_EMIT_ADD_B, OP_EMIT_SUB_B, OP_EMIT_MUL_B,   // transition constraint `dst` = a (+, -, *) b, both base registers: the root node never
                                             // travels through the register file
    OP_EMIT3_B,                              // transition constraint = a o1 (b o2 c) or (b o2 c) o1 a over three base registers: the last TWO
                                             // nodes of the expression folded into the EMIT. op bits 16-17: o1, 18-19: o2 (1 +, 2 -, 3 *),
                                             // bit 20: inner node on the left; on the device `dst` carries register c
};
constexpr uint32_t LOAD_WIDTH = 8;           // loads issued back to back by one LOAD instruction (memory-level parallelism per lane)
// One instruction = 32 bytes. op = DOp | kind(a) << 8 | kind(b) << 12; dst / a / b = register slots or table indices. The device
// never follows an index into a table: per proof the host lays every scalar an instruction can need (scalar operands,
// composition-coefficient pairs, LOAD descriptors) out in ONE pool and writes the instruction's position in it into `poff`
// (`pk` says which table `pi` indexes); the interpreter prefetches pool[poff] one instruction ahead.
enum PoolKind : uint32_t { PK_NONE = 0, PK_SCAL_B, PK_SCAL_E, PK_TCOEF, PK_BCOEF, PK_DESC };
struct Insn { uint32_t op, dst, a, b, poff, pk, pi, pad; };
// LOAD descriptor as compiled: next-row flag << 31 | periodic flag << 30 | slot << 16 | column (or periodic column id)
inline uint32_t mk_desc(bool next, bool periodic, uint32_t slot, uint32_t col) { return (next ? 0x80000000u : 0u) | (periodic ? 0x40000000u : 0u) | (slot << 16) | col; }
inline bool dkind_is_ext(uint32_t k) { return k == D_SLOT_E || k == D_AUX_CUR || k == D_AUX_NXT || k == D_SCAL_E; }
inline bool dkind_is_frame(uint32_t k) { return k >= D_MAIN_CUR && k <= D_PERIODIC; }      // arrives through a LOAD (periodic values too)
inline bool dkind_is_scalar(uint32_t k) { return k == D_SCAL_B || k == D_SCAL_E; }

struct Node { uint32_t op, a, b; };
struct Transition { uint32_t root, base; std::vector<uint32_t> cycles; uint32_t group; };
struct Assertion { uint32_t col; int64_t first; uint32_t stride, value; };
struct Builder { uint32_t init, num, den, add_num = REF_NONE, add_den = REF_NONE; };   // column(i+1) = column(i) * num / den + add_num / add_den

struct Program {
    // ---- as written
    uint32_t W = 0, A = 0, R = 0, num_pub = 0, exemptions = 1;
    std::vector<uint64_t> consts;
    std::vector<std::vector<uint64_t>> periodic;
    std::vector<std::vector<uint64_t>> sequences;   // version 2: the value lists of `Assertion::sequence` (an assertion's value = K_SEQ | index)
    std::vector<Node> nodes;
    std::vector<Transition> trans;            // main first
    uint32_t n_main_trans = 0;
    std::vector<Assertion> masserts, aasserts;
    std::vector<Builder> builders;
    // ---- analysis
    std::vector<uint8_t> is_ext, row_dep, uses_aux;
    std::vector<int32_t> scalar_of;           // node -> index into scalB / scalE when folded, else -1
    std::vector<uint32_t> fold_order;         // folded nodes in evaluation order
    uint32_t n_scalB = 0, n_scalE = 0;        // scalB = consts | pubs | folded base nodes; scalE = rands | folded E nodes
    struct DegGroup { uint32_t base; std::vector<uint32_t> cycles; };
    std::vector<DegGroup> dgroups;
    uint32_t ce_blowup = 2;
    // ---- code
    std::vector<Insn> cons_code, aux_code;
    std::vector<uint32_t> cons_desc, aux_desc;       // LOAD descriptors
    uint32_t cons_slotsB = 0, cons_slotsE = 0, aux_slotsB = 0, aux_slotsE = 0;
    std::vector<uint8_t> has_den;             // per aux column
    std::vector<uint8_t> has_add;             // per aux column: 0 = pure running product, 1 = additive term, 3 = additive term with a denominator, 4 = general recurrence (host)
    std::vector<std::vector<uint32_t>> general_nodes;   // per aux column with a general recurrence: the nodes its expression needs, ascending
    std::vector<uint32_t> general_main_cols;  // main columns the general recurrences read (the host evaluation downloads these)
    mutable std::shared_ptr<void> jit_cache;  // code objects of the run-time compiled evaluation kernels (air_jit.hip), shared by copies
    mutable std::shared_ptr<void> seq_cache;  // interpolants of the sequence assertions per (sequence, trace length, first step)

    size_t num_transition() const { return trans.size(); }
    size_t num_assertions() const { return masserts.size() + aasserts.size(); }
    bool has_builders() const { return A > 0 && builders.size() == A; }
};

// operand reference -> device operand (kind, index); nodes that live in registers keep their node id as a virtual slot
struct DOperand { uint32_t kind, idx; bool virt; };

namespace detail {
struct Reader {
    const uint8_t* p; size_t len, off = 0;
    void need(size_t k) const { if (off + k > len) fail("air program: truncated"); }
    uint32_t u32() { need(4); uint32_t v; memcpy(&v, p + off, 4); off += 4; return v; }
    uint64_t felt() { need(8); uint64_t v; memcpy(&v, p + off, 8); off += 8; if (v >= gl::P) fail("air program: non-canonical field element"); return v; }
};
}  // namespace detail

inline DOperand device_operand(const Program& p, uint32_t ref) {
    const uint32_t k = ref_kind(ref), i = ref_index(ref);
    switch (k) {
        case K_NODE:
            if (p.scalar_of[i] >= 0) return DOperand{p.is_ext[i] ? D_SCAL_E : D_SCAL_B, (uint32_t)p.scalar_of[i], false};
            return DOperand{p.is_ext[i] ? D_SLOT_E : D_SLOT_B, i, true};
        case K_MAIN_CUR: return DOperand{D_MAIN_CUR, i, false};
        case K_MAIN_NXT: return DOperand{D_MAIN_NXT, i, false};
        case K_AUX_CUR: return DOperand{D_AUX_CUR, i, false};
        case K_AUX_NXT: return DOperand{D_AUX_NXT, i, false};
        case K_PERIODIC: return DOperand{D_PERIODIC, i, false};
        case K_CONST: return DOperand{D_SCAL_B, i, false};
        case K_PUB: return DOperand{D_SCAL_B, (uint32_t)p.consts.size() + i, false};
        default: return DOperand{D_SCAL_E, i, false};
    }
}
inline bool ref_is_ext(const Program& p, uint32_t ref) {
    const uint32_t k = ref_kind(ref);
    return k == K_AUX_CUR || k == K_AUX_NXT || k == K_RAND || (k == K_NODE && p.is_ext[ref_index(ref)]);
}
inline bool ref_row_dep(const Program& p, uint32_t ref) {
    const uint32_t k = ref_kind(ref);
    return (k >= K_MAIN_CUR && k <= K_PERIODIC) || (k == K_NODE && p.row_dep[ref_index(ref)]);
}
inline bool ref_uses_aux(const Program& p, uint32_t ref) {
    const uint32_t k = ref_kind(ref);
    return k == K_AUX_CUR || k == K_AUX_NXT || (k == K_NODE && p.uses_aux[ref_index(ref)]);
}

// Code generation. gen(root) emits the not-yet-computed nodes of an expression in post-order (common sub-expressions once);
// lower() then makes every frame access explicit: LOAD instructions fetch LOAD_WIDTH frame values at a time into registers (the
// loads of one instruction are issued back to back - a lane waits once per batch, not once per operand), looking ahead in the
// stream for the values the following instructions need; the assertions on a column are evaluated right behind the load that
// brings its current-row value (no second read of the row); a value whose next use is far away is dropped and re-loaded.
// allocate() is a linear scan over the result: every live value gets a slot of the per-lane register file, reused after its last use.
struct VOperand { uint32_t kind, idx; bool virt; };            // virt: idx is a virtual register
struct VInsn {
    uint32_t op = OP_END, dst = 0;
    VOperand a{D_SCAL_B, 0, false}, b{D_SCAL_B, 0, false}, c{D_SCAL_B, 0, false};   // c: third operand of OP_EMIT3_B
    std::vector<std::pair<uint32_t, uint32_t>> loads;          // LOAD: (frame operand key, virtual register)
};
struct CodeGen {
    const Program& p;
    std::vector<VInsn> code;
    std::vector<uint8_t> done;
    std::vector<uint8_t> vreg_ext;                              // per virtual register: E-valued?
    static constexpr uint32_t RELOAD_GAP = 96;                  // instructions a frame value may wait in a register for its next use
    explicit CodeGen(const Program& prog) : p(prog), done(prog.nodes.size(), 0) {
        vreg_ext.assign(prog.nodes.size(), 0);
        for (size_t i = 0; i < prog.nodes.size(); i++) vreg_ext[i] = prog.is_ext[i];
    }
    static VOperand vop(const DOperand& o) { return VOperand{o.kind, o.idx, o.virt}; }
    void gen(uint32_t ref) {
        if (ref_kind(ref) != K_NODE) return;
        std::vector<std::pair<uint32_t, int>> st;     // (node, stage)
        st.push_back({ref_index(ref), 0});
        while (!st.empty()) {
            auto [i, stage] = st.back();
            st.pop_back();
            if (done[i] || p.scalar_of[i] >= 0) continue;
            const Node& nd = p.nodes[i];
            if (stage == 0) {
                st.push_back({i, 1});
                for (uint32_t r : {nd.b, nd.a})
                    if (ref_kind(r) == K_NODE && !done[ref_index(r)] && p.scalar_of[ref_index(r)] < 0) st.push_back({ref_index(r), 0});
                continue;
            }
            DOperand a = device_operand(p, nd.a), b = device_operand(p, nd.b);
            uint32_t op;
            if (!p.is_ext[i]) op = nd.op == 1 ? OP_ADD_B : nd.op == 2 ? OP_SUB_B : OP_MUL_B;
            else if (nd.op != 3) op = nd.op == 1 ? OP_ADD_E : OP_SUB_E;
            else {
                const bool ea = dkind_is_ext(a.kind), eb = dkind_is_ext(b.kind);
                if (ea && eb) op = OP_MUL_E;
                else { op = OP_MULB_E; if (!ea) std::swap(a, b); }
            }
            VInsn in;
            in.op = op; in.dst = i; in.a = vop(a); in.b = vop(b);
            code.push_back(in);
            done[i] = 1;
        }
    }
    void emit(uint32_t op, uint32_t dst, uint32_t ref) {
        VInsn in;
        in.op = op; in.dst = dst; in.a = vop(device_operand(p, ref));
        code.push_back(in);
    }
    // a transition constraint whose root is a base-field node nobody else reads: its operands are generated, the root operation
    // itself is folded into the EMIT (one instruction and one register-file round trip less per constraint)
    bool emit_fused(uint32_t k, uint32_t ref, const std::vector<uint32_t>& node_uses) {
        if (ref_kind(ref) != K_NODE) return false;
        const uint32_t i = ref_index(ref);
        if (done[i] || p.scalar_of[i] >= 0 || p.is_ext[i] || node_uses[i] != 1) return false;
        const Node& nd = p.nodes[i];
        const DOperand a = device_operand(p, nd.a), b = device_operand(p, nd.b);
        if (dkind_is_scalar(a.kind) || dkind_is_scalar(b.kind)) return false;          // the pool block is taken by the coefficients
        // one level deeper: an operand that is itself a single-use base node over two registers
        for (int side = 1; side >= 0; side--) {                                        // side 1: the inner node is operand a (left)
            const uint32_t inner_ref = side ? nd.a : nd.b, other_ref = side ? nd.b : nd.a;
            if (ref_kind(inner_ref) != K_NODE) continue;
            const uint32_t j = ref_index(inner_ref);
            if (done[j] || p.scalar_of[j] >= 0 || p.is_ext[j] || node_uses[j] != 1) continue;
            const Node& in2 = p.nodes[j];
            const DOperand x = device_operand(p, in2.a), y = device_operand(p, in2.b), o = device_operand(p, other_ref);
            if (dkind_is_scalar(x.kind) || dkind_is_scalar(y.kind)) continue;
            gen(other_ref); gen(in2.a); gen(in2.b);
            VInsn in;
            in.op = OP_EMIT3_B | (nd.op << 16) | (in2.op << 18) | ((uint32_t)side << 20);
            in.dst = k; in.a = vop(o); in.b = vop(x); in.c = vop(y);
            code.push_back(in);
            done[i] = done[j] = 1;
            return true;
        }
        gen(nd.a); gen(nd.b);
        VInsn in;
        in.op = nd.op == 1 ? OP_EMIT_ADD_B : nd.op == 2 ? OP_EMIT_SUB_B : OP_EMIT_MUL_B;
        in.dst = k; in.a = vop(a); in.b = vop(b);
        code.push_back(in);
        done[i] = 1;
        return true;
    }
    void marker(uint32_t op, uint32_t dst) { VInsn in; in.op = op; in.dst = dst; code.push_back(in); }
    static bool writes_slot(uint32_t op) { op &= 0xff; return op >= OP_ADD_B && op <= OP_MULB_E; }
    static bool is_emit2(uint32_t op) { op &= 0xff; return op >= OP_EMIT_ADD_B && op <= OP_EMIT_MUL_B; }
    static bool is_emit3(uint32_t op) { return (op & 0xff) == OP_EMIT3_B; }
    static bool reads_b(uint32_t op) { return writes_slot(op) || is_emit2(op) || is_emit3(op); }
    static bool reads_c(uint32_t op) { return is_emit3(op); }
    static bool reads_a(uint32_t op) { op &= 0xff; return op != OP_END && op != OP_GROUP_END && op != OP_LOAD_MAIN && op != OP_LOAD_AUX; }
    static uint32_t frame_key(const VOperand& o) { return (o.kind << 16) | o.idx; }

    // the operands an instruction reads
    static std::vector<VOperand*> reads(VInsn& in) {
        std::vector<VOperand*> r;
        if (reads_a(in.op)) r.push_back(&in.a);
        if (reads_b(in.op)) r.push_back(&in.b);
        if (reads_c(in.op)) r.push_back(&in.c);
        return r;
    }
    // `bound[k]` = assertions (member ids) on the current-row value with frame key k
    void lower(const std::map<uint32_t, std::vector<uint32_t>>& bound) {
        std::vector<VInsn> in;
        in.swap(code);
        // use positions of every frame operand
        std::map<uint32_t, std::vector<size_t>> uses;
        for (size_t i = 0; i < in.size(); i++)
            for (VOperand* o : reads(in[i]))
                if (dkind_is_frame(o->kind)) { auto& u = uses[frame_key(*o)]; if (u.empty() || u.back() != i) u.push_back(i); }
        std::map<uint32_t, uint32_t> avail;          // frame key -> virtual register holding it
        std::map<uint32_t, uint8_t> bound_done;
        auto is_aux = [](uint32_t key) { const uint32_t k = key >> 16; return k == D_AUX_CUR || k == D_AUX_NXT; };     // periodic values travel with the main batches
        auto emit_load = [&](const std::vector<uint32_t>& keys, bool aux) {
            VInsn ld;
            ld.op = aux ? OP_LOAD_AUX : OP_LOAD_MAIN;
            for (uint32_t k : keys) {
                const uint32_t v = (uint32_t)vreg_ext.size();
                vreg_ext.push_back(aux ? 1 : 0);
                ld.loads.push_back({k, v});
                avail[k] = v;
            }
            code.push_back(ld);
            for (auto& kv : ld.loads) {
                auto it = bound.find(kv.first);
                if (it == bound.end() || bound_done[kv.first]) continue;
                bound_done[kv.first] = 1;
                for (uint32_t m : it->second) {
                    VInsn bi;
                    bi.op = aux ? OP_BOUND_E : OP_BOUND_B; bi.dst = m; bi.a = VOperand{aux ? D_SLOT_E : D_SLOT_B, kv.second, true};
                    code.push_back(bi);
                }
            }
        };
        // the frame value `key` is needed at position i: load it together with the next values the stream will ask for
        auto demand = [&](uint32_t key, size_t i) {
            const bool aux = is_aux(key);
            const uint32_t width = aux ? LOAD_WIDTH / 2 : LOAD_WIDTH;
            std::vector<uint32_t> keys{key};
            for (size_t j = i; j < in.size() && keys.size() < width && j < i + 4 * RELOAD_GAP; j++)
                for (VOperand* o : reads(in[j])) {
                    if (!dkind_is_frame(o->kind)) continue;
                    const uint32_t k = frame_key(*o);
                    if (is_aux(k) != aux || avail.count(k) || std::find(keys.begin(), keys.end(), k) != keys.end()) continue;
                    if (keys.size() < width) keys.push_back(k);
                }
            emit_load(keys, aux);
        };
        for (size_t i = 0; i < in.size(); i++) {
            VInsn cur = in[i];
            for (VOperand* o : reads(cur)) {
                if (!dkind_is_frame(o->kind)) continue;
                const uint32_t key = frame_key(*o);
                if (!avail.count(key)) demand(key, i);
                *o = VOperand{is_aux(key) ? D_SLOT_E : D_SLOT_B, avail[key], true};
            }
            code.push_back(cur);
            // retire values whose next use is far away (or never)
            for (VOperand* o : reads(in[i])) {
                if (!dkind_is_frame(o->kind)) continue;
                const uint32_t key = frame_key(*o);
                const auto& u = uses[key];
                auto nx = std::upper_bound(u.begin(), u.end(), i);
                if (nx == u.end() || *nx - i > RELOAD_GAP) avail.erase(key);
            }
        }
        // columns that are only asserted, never read by a constraint
        std::vector<uint32_t> rest_main, rest_aux;
        for (auto& kv : bound) if (!bound_done[kv.first]) (is_aux(kv.first) ? rest_aux : rest_main).push_back(kv.first);
        for (size_t o = 0; o < rest_main.size(); o += LOAD_WIDTH)
            emit_load(std::vector<uint32_t>(rest_main.begin() + o, rest_main.begin() + std::min(rest_main.size(), o + LOAD_WIDTH)), false);
        for (size_t o = 0; o < rest_aux.size(); o += LOAD_WIDTH / 2)
            emit_load(std::vector<uint32_t>(rest_aux.begin() + o, rest_aux.begin() + std::min(rest_aux.size(), o + LOAD_WIDTH / 2)), true);
    }
    // virtual registers -> slots; encodes the instructions and the LOAD descriptors
    void allocate(std::vector<Insn>& out, std::vector<uint32_t>& desc, uint32_t* slotsB, uint32_t* slotsE) {
        std::map<uint32_t, size_t> last;
        for (size_t i = 0; i < code.size(); i++)
            for (VOperand* o : reads(code[i])) if (o->virt) last[o->idx] = i;
        std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> freeB, freeE;
        uint32_t nB = 0, nE = 0;
        std::map<uint32_t, uint32_t> slot;
        auto take = [&](uint32_t v) {
            auto& fl = vreg_ext[v] ? freeE : freeB;
            uint32_t s;
            if (!fl.empty()) { s = fl.top(); fl.pop(); } else s = vreg_ext[v] ? nE++ : nB++;
            slot[v] = s;
            return s;
        };
        auto give = [&](uint32_t v) { (vreg_ext[v] ? freeE : freeB).push(slot.at(v)); };
        for (size_t i = 0; i < code.size(); i++) {
            VInsn& in = code[i];
            const uint32_t opc = in.op & 0xff;
            if (opc == OP_LOAD_MAIN || opc == OP_LOAD_AUX) {
                const uint32_t width = opc == OP_LOAD_AUX ? LOAD_WIDTH / 2 : LOAD_WIDTH;
                const uint32_t first = (uint32_t)desc.size();
                std::vector<uint32_t> dead;
                for (auto& kv : in.loads) {
                    const uint32_t k = kv.first >> 16, col = kv.first & 0xffff;
                    desc.push_back(mk_desc(k == D_MAIN_NXT || k == D_AUX_NXT, k == D_PERIODIC, take(kv.second), col));
                    if (!last.count(kv.second)) dead.push_back(kv.second);
                }
                while (desc.size() < first + width) desc.push_back(desc.back());     // padding: the last value once more
                for (uint32_t v : dead) give(v);
                out.push_back(Insn{in.op, first, (uint32_t)in.loads.size(), 0, 0, PK_DESC, first, 0});
                continue;
            }
            Insn e{in.op | (in.a.kind << 8) | (in.b.kind << 12), in.dst, in.a.idx, in.b.idx, 0, PK_NONE, 0, 0};
            // what the instruction needs from the pool: its scalar operand (at most one: scalar-only nodes were folded), or the
            // coefficient pair of the constraint / assertion it feeds
            if (opc == OP_EMIT_B || opc == OP_EMIT_E || is_emit2(in.op) || is_emit3(in.op)) { e.pk = PK_TCOEF; e.pi = in.dst; }
            else if (opc == OP_BOUND_B || opc == OP_BOUND_E) { e.pk = PK_BCOEF; e.pi = in.dst; }
            else if (reads_a(in.op) && dkind_is_scalar(in.a.kind)) { e.pk = in.a.kind == D_SCAL_E ? PK_SCAL_E : PK_SCAL_B; e.pi = in.a.idx; }
            else if (reads_b(in.op) && dkind_is_scalar(in.b.kind)) { e.pk = in.b.kind == D_SCAL_E ? PK_SCAL_E : PK_SCAL_B; e.pi = in.b.idx; }
            if ((opc == OP_EMIT_B || opc == OP_EMIT_E) && dkind_is_scalar(in.a.kind))
                fail("air program: a transition constraint that does not depend on the trace");
            if (writes_slot(in.op) && dkind_is_scalar(in.a.kind) && dkind_is_scalar(in.b.kind)) fail("air program: unfolded constant node", ST_INTERNAL);
            std::vector<uint32_t> dying;
            uint32_t* fields[3] = {&e.a, &e.b, &e.pad};       // the third operand of EMIT3 travels in `pad` (host) and in `dst` on the device
            int fi = 0;
            for (VOperand* o : {&in.a, &in.b, &in.c}) {
                const int f = fi++;
                if (f == 0 ? !reads_a(in.op) : f == 1 ? !reads_b(in.op) : !reads_c(in.op)) continue;
                if (f == 2) *fields[2] = o->idx;
                if (!o->virt) continue;
                *fields[f] = slot.at(o->idx);
                if (last[o->idx] == i && std::find(dying.begin(), dying.end(), o->idx) == dying.end()) dying.push_back(o->idx);
            }
            for (uint32_t v : dying) give(v);
            if (writes_slot(in.op)) {
                e.dst = take(in.dst);
                if (!last.count(in.dst)) give(in.dst);
            }
            out.push_back(e);
        }
        *slotsB = nB; *slotsE = nE;
        if (nB >= 4096 || nE >= 4096) fail("air program: register file too large", ST_UNSUPPORTED);
    }
};

inline Program load(const uint8_t* bytes, size_t len) {
    if (!bytes || len < 8 + 64) fail("air program: too short");
    if (memcmp(bytes, "AEROAIR", 7) != 0 || (bytes[7] != 1 && bytes[7] != 2)) fail("air program: not an AEROAIR version-1 or version-2 program");
    const int version = bytes[7];
    detail::Reader rd{bytes, len, 8};
    uint32_t h[16];
    for (auto& v : h) v = rd.u32();
    Program p;
    p.W = h[0]; p.A = h[1]; p.R = h[2]; p.num_pub = h[3]; p.exemptions = h[4];
    const uint32_t nc = h[5], np = h[6], nn = h[7], nmt = h[8], nat = h[9], nma = h[10], naa = h[11], nb = h[12];
    if (p.W < 1 || p.W > 255) fail("air program: main width must be in [1, 255]");
    if (p.A > 255 - p.W) fail("air program: main + aux width must not exceed 255");
    if ((p.A == 0) != (p.R == 0) || p.R > 255) fail("air program: an auxiliary segment needs 1..255 random elements (and none without one)");
    if (p.num_pub > 4096) fail("air program: at most 4096 public inputs");
    if (p.exemptions < 1 || p.exemptions > 64) fail("air program: 1..64 transition exemptions");
    const uint32_t nseq = version >= 2 ? h[13] : 0;
    if ((version < 2 && h[13]) || h[14] || h[15]) fail("air program: reserved header words must be zero");
    if (nseq > 4096) fail("air program: at most 4096 sequences", ST_UNSUPPORTED);
    if (nb != 0 && nb != p.A) fail("air program: one aux builder per auxiliary column, or none");
    if (nn > (1u << 22) || nc > (1u << 22) || np > 4096 || (uint64_t)nmt + nat > (1u << 20) || (uint64_t)nma + naa > (1u << 20)) fail("air program: section too large", ST_UNSUPPORTED);
    if (nmt + nat == 0) fail("air program: no transition constraints");
    if (nat && !p.A) fail("air program: auxiliary constraints without an auxiliary segment");
    if (naa && !p.A) fail("air program: auxiliary assertions without an auxiliary segment");
    p.consts.resize(nc);
    for (auto& v : p.consts) v = rd.felt();
    for (uint32_t i = 0; i < np; i++) {
        const uint32_t cl = rd.u32();
        if (cl < 2 || (cl & (cl - 1)) || cl > (1u << 24)) fail("air program: a periodic column's cycle length must be a power of two >= 2");
        std::vector<uint64_t> v(cl);
        for (auto& x : v) x = rd.felt();
        p.periodic.push_back(std::move(v));
    }
    for (uint32_t i = 0; i < nseq; i++) {
        const uint32_t cnt = rd.u32();
        if (cnt < 2 || (cnt & (cnt - 1)) || cnt > (1u << 28)) fail("air program: a sequence needs a power-of-two number of values >= 2");
        rd.need((size_t)cnt * 8);
        std::vector<uint64_t> v(cnt);
        for (auto& x : v) x = rd.felt();
        p.sequences.push_back(std::move(v));
    }
    auto check_ref = [&](uint32_t ref, uint32_t node_limit, const char* what) {
        const uint32_t k = ref_kind(ref), i = ref_index(ref);
        const uint32_t lim[K_COUNT] = {node_limit, p.W, p.W, p.A, p.A, np, nc, p.num_pub, p.R};
        if (k >= K_COUNT || i >= lim[k]) fail(std::string("air program: operand out of range in ") + what);
    };
    p.nodes.resize(nn);
    p.is_ext.assign(nn, 0); p.row_dep.assign(nn, 0); p.uses_aux.assign(nn, 0); p.scalar_of.assign(nn, -1);
    p.n_scalB = nc + p.num_pub; p.n_scalE = p.R;
    for (uint32_t i = 0; i < nn; i++) {
        Node nd{rd.u32(), rd.u32(), rd.u32()};
        if (nd.op < 1 || nd.op > 3) fail("air program: unknown node opcode");
        check_ref(nd.a, i, "a node"); check_ref(nd.b, i, "a node");
        p.nodes[i] = nd;
        p.is_ext[i] = ref_is_ext(p, nd.a) || ref_is_ext(p, nd.b);
        p.row_dep[i] = ref_row_dep(p, nd.a) || ref_row_dep(p, nd.b);
        p.uses_aux[i] = ref_uses_aux(p, nd.a) || ref_uses_aux(p, nd.b);
        if (!p.row_dep[i]) { p.scalar_of[i] = (int32_t)(p.is_ext[i] ? p.n_scalE++ : p.n_scalB++); p.fold_order.push_back(i); }
    }
    p.n_main_trans = nmt;
    std::map<std::pair<uint32_t, std::vector<uint32_t>>, uint32_t> gkey;
    for (uint32_t i = 0; i < nmt + nat; i++) {
        Transition t;
        t.root = rd.u32(); t.base = rd.u32();
        const uint32_t ncy = rd.u32();
        if (t.base < 1 || t.base > 255) fail("air program: a constraint's degree must be in [1, 255]");
        if (ncy > 32) fail("air program: more than 32 cycles in a constraint degree");
        for (uint32_t j = 0; j < ncy; j++) {
            const uint32_t c = rd.u32();
            if (c < 2 || (c & (c - 1))) fail("air program: cycle lengths must be powers of two >= 2");
            t.cycles.push_back(c);
        }
        check_ref(t.root, nn, "a transition constraint");
        if (i < nmt && ref_is_ext(p, t.root)) fail("air program: a main transition constraint may not depend on the auxiliary segment or the random elements");
        std::vector<uint32_t> sc = t.cycles;
        std::sort(sc.begin(), sc.end());
        auto key = std::make_pair(t.base, sc);
        auto it = gkey.find(key);
        if (it == gkey.end()) { it = gkey.insert({key, (uint32_t)p.dgroups.size()}).first; p.dgroups.push_back(Program::DegGroup{t.base, sc}); }
        t.group = it->second;
        size_t d = t.base + ncy, e = 2;
        while (e < d) e <<= 1;
        if (e > p.ce_blowup) p.ce_blowup = (uint32_t)e;
        p.trans.push_back(std::move(t));
    }
    if (p.ce_blowup > 128) fail("air program: constraint-evaluation blowup above 128", ST_UNSUPPORTED);
    for (uint32_t i = 0; i < nma + naa; i++) {
        Assertion s;
        s.col = rd.u32(); s.first = (int32_t)rd.u32(); s.stride = rd.u32(); s.value = rd.u32();
        const bool aux = i >= nma;
        if (s.col >= (aux ? p.A : p.W)) fail("air program: assertion column out of range");
        if (s.stride && (s.stride < 2 || (s.stride & (s.stride - 1)))) fail("air program: an assertion's stride must be 0 or a power of two >= 2");
        if (ref_kind(s.value) == K_SEQ) {
            // `Assertion::sequence(column, first_step, stride, values)`: the column equals values[i] at first_step + i * stride
            if (ref_index(s.value) >= nseq) fail("air program: sequence index out of range in an assertion");
            if (!s.stride) fail("air program: a sequence assertion needs a stride");
        } else {
            check_ref(s.value, nn, "an assertion");
            if (ref_row_dep(p, s.value)) fail("air program: an assertion's value must not depend on the trace or a periodic column");
            if (!aux && ref_is_ext(p, s.value)) fail("air program: a main assertion's value must be a base-field quantity (constant or public input)");
        }
        (aux ? p.aasserts : p.masserts).push_back(s);
    }
    for (uint32_t i = 0; i < nb; i++) {
        Builder b;
        b.init = rd.u32(); b.num = rd.u32(); b.den = rd.u32();
        if (version >= 2) { b.add_num = rd.u32(); b.add_den = rd.u32(); }      // version 2: affine recurrence (running sums, mixed forms)
        check_ref(b.init, nn, "an aux builder"); check_ref(b.num, nn, "an aux builder");
        if (b.den != REF_NONE && b.den != REF_GENERAL) check_ref(b.den, nn, "an aux builder");
        if (b.add_num != REF_NONE) check_ref(b.add_num, nn, "an aux builder");
        if (b.add_den != REF_NONE) { if (b.add_num == REF_NONE) fail("air program: an aux builder's additive denominator needs a numerator"); check_ref(b.add_den, nn, "an aux builder"); }
        if (ref_row_dep(p, b.init)) fail("air program: an aux builder's initial value must not depend on the trace");
        if (b.den == REF_GENERAL) {
            // general recurrence: `num` is the whole next value; it may read the current row of auxiliary columns up to and including its own
            if (version < 2) fail("air program: general aux recurrences need version 2");
            if (b.add_num != REF_NONE || b.add_den != REF_NONE) fail("air program: a general aux recurrence has no additive term");
        } else
        for (uint32_t r : {b.num, b.den, b.add_num, b.add_den})
            if (r != REF_NONE && ref_uses_aux(p, r)) fail("air program: an aux builder's terms may only read the main segment");
        p.builders.push_back(b);
    }
    if (rd.off != len) fail("air program: trailing bytes");

    // ---- constraint code: degree group by degree group, then the assertions woven in behind the loads of their columns
    {
        CodeGen cg(p);
        std::vector<uint32_t> node_uses(p.nodes.size(), 0);        // readers of every node: other nodes, constraint roots, builders
        auto count_use = [&](uint32_t ref) { if (ref != REF_NONE && ref_kind(ref) == K_NODE) node_uses[ref_index(ref)]++; };
        for (auto& nd : p.nodes) { count_use(nd.a); count_use(nd.b); }
        for (auto& t : p.trans) count_use(t.root);
        for (auto& b : p.builders) { count_use(b.init); count_use(b.num); if (b.den != REF_GENERAL) count_use(b.den); count_use(b.add_num); count_use(b.add_den); }
        for (auto* v : {&p.masserts, &p.aasserts}) for (auto& as : *v) count_use(as.value);
        for (uint32_t g = 0; g < p.dgroups.size(); g++) {
            for (uint32_t k = 0; k < p.trans.size(); k++) {
                if (p.trans[k].group != g) continue;
                if (cg.emit_fused(k, p.trans[k].root, node_uses)) continue;
                cg.gen(p.trans[k].root);
                cg.emit(ref_is_ext(p, p.trans[k].root) ? OP_EMIT_E : OP_EMIT_B, k, p.trans[k].root);
            }
            cg.marker(OP_GROUP_END, g);
        }
        std::map<uint32_t, std::vector<uint32_t>> bound;       // frame key of the asserted column -> member ids (main, then aux)
        for (uint32_t m = 0; m < p.masserts.size(); m++) bound[(D_MAIN_CUR << 16) | p.masserts[m].col].push_back(m);
        for (uint32_t m = 0; m < p.aasserts.size(); m++) bound[(D_AUX_CUR << 16) | p.aasserts[m].col].push_back((uint32_t)p.masserts.size() + m);
        cg.lower(bound);
        cg.marker(OP_END, 0);
        cg.allocate(p.cons_code, p.cons_desc, &p.cons_slotsB, &p.cons_slotsE);
    }
    // ---- aux builder code: output column 2c = numerator factor, 2c + 1 = denominator factor of aux column c; affine builders
    //      (version 2) additionally 2A + 2c = numerator and 2A + 2c + 1 = denominator of the additive term
    if (p.has_builders()) {
        CodeGen cg(p);
        p.has_den.assign(p.A, 0);
        p.has_add.assign(p.A, 0);
        p.general_nodes.assign(p.A, {});
        std::vector<uint8_t> gmain(p.W, 0);
        for (uint32_t c = 0; c < p.A; c++) {
            if (p.builders[c].den == REF_GENERAL) {
                // evaluated row after row on the host (air_host.hip: host_general_column): the nodes the expression reaches, in index order
                p.has_add[c] = 4;
                std::vector<uint8_t> seen(p.nodes.size(), 0);
                std::vector<uint32_t> stack;
                auto visit_ref = [&](uint32_t r) {
                    const uint32_t k = ref_kind(r), i = ref_index(r);
                    if (k == K_NODE) { if (!seen[i] && p.scalar_of[i] < 0) { seen[i] = 1; stack.push_back(i); } }
                    else if (k == K_MAIN_CUR || k == K_MAIN_NXT) gmain[i] = 1;
                    else if (k == K_AUX_NXT) fail("air program: a general aux recurrence reads the CURRENT row of the auxiliary segment only");
                    else if (k == K_AUX_CUR && i > c) fail("air program: a general aux recurrence may read auxiliary columns up to its own index only");
                };
                visit_ref(p.builders[c].num);
                while (!stack.empty()) {
                    const uint32_t i = stack.back();
                    stack.pop_back();
                    p.general_nodes[c].push_back(i);
                    visit_ref(p.nodes[i].a); visit_ref(p.nodes[i].b);
                }
                std::sort(p.general_nodes[c].begin(), p.general_nodes[c].end());
                continue;
            }
            if (p.builders[c].add_num != REF_NONE) {
                p.has_add[c] = 1;
                cg.gen(p.builders[c].add_num);
                cg.emit(ref_is_ext(p, p.builders[c].add_num) ? OP_OUT_E : OP_OUT_B, 2 * p.A + 2 * c, p.builders[c].add_num);
                if (p.builders[c].add_den != REF_NONE) {
                    p.has_add[c] = 3;
                    cg.gen(p.builders[c].add_den);
                    cg.emit(ref_is_ext(p, p.builders[c].add_den) ? OP_OUT_E : OP_OUT_B, 2 * p.A + 2 * c + 1, p.builders[c].add_den);
                }
            }
            cg.gen(p.builders[c].num);
            cg.emit(ref_is_ext(p, p.builders[c].num) ? OP_OUT_E : OP_OUT_B, 2 * c, p.builders[c].num);
            if (p.builders[c].den != REF_NONE) {
                p.has_den[c] = 1;
                cg.gen(p.builders[c].den);
                cg.emit(ref_is_ext(p, p.builders[c].den) ? OP_OUT_E : OP_OUT_B, 2 * c + 1, p.builders[c].den);
            }
        }
        for (uint32_t c = 0; c < p.W; c++) if (gmain[c]) p.general_main_cols.push_back(c);
        cg.lower({});
        cg.marker(OP_END, 0);
        cg.allocate(p.aux_code, p.aux_desc, &p.aux_slotsB, &p.aux_slotsE);
    }
    if (p.cons_slotsB + 2 * p.cons_slotsE > 1024 || p.aux_slotsB + 2 * p.aux_slotsE > 1024)
        fail("air program: more than 1024 live values at one point (the per-lane register file of the interpreter)", ST_UNSUPPORTED);
    return p;
}

// ---- instance: what depends on the trace length -------------------------------------------------------------------------------
// An assertion as the instance sees it: member id = position in the program (main assertions, then aux); `coef` = its place in
// the sorted order that hands out the composition coefficients; `group` = its divisor group (numerator column 1 + group).
struct BoundaryMember { uint32_t col, aux, coef, group, val_ext, val_idx; int32_t seq = -1; };   // value = scalB[val_idx] or scalE[val_idx]; seq >= 0: sequences[seq]
struct BoundaryGroup { uint32_t stride; uint64_t first, a, b, adj; bool has_seq = false; };
struct Instance {
    int log_n = 0;
    uint64_t n = 0, ce_n = 0;
    std::vector<uint64_t> dgroup_adj;          // per degree group
    std::vector<BoundaryGroup> bgroups;        // numerator column 1 + j
    std::vector<BoundaryMember> members;       // by member id
    size_t num_columns() const { return 1 + bgroups.size(); }
};
inline Instance instantiate(const Program& p, int log_n) {
    Instance in;
    if (log_n < 3 || log_n > 29) fail("air program: trace length must be between 2^3 and 2^29");
    in.log_n = log_n; in.n = 1ull << log_n; in.ce_n = in.n * p.ce_blowup;
    const uint64_t n = in.n;
    if (p.exemptions >= n) fail("air program: more transition exemptions than trace steps");
    for (auto& v : p.periodic) if (v.size() > n) fail("air program: a periodic column's cycle is longer than the trace");
    const uint64_t target = in.ce_n - 1 + (n - p.exemptions);
    for (auto& g : p.dgroups) {
        uint64_t ed = (uint64_t)g.base * (n - 1);
        for (uint32_t c : g.cycles) { if (c > n) fail("air program: a degree cycle is longer than the trace"); ed += (n / c) * (c - 1); }
        if (ed > target) fail("air program: a constraint's evaluation degree exceeds the composition degree");
        in.dgroup_adj.push_back(target - ed);
    }
    const uint64_t g = gl::root_of_unity(log_n);
    uint32_t coef = 0;
    in.members.resize(p.masserts.size() + p.aasserts.size());
    for (int seg = 0; seg < 2; seg++) {
        struct Item { Assertion s; uint32_t id; };
        std::vector<Item> v;
        const auto& src = seg == 0 ? p.masserts : p.aasserts;
        for (uint32_t i = 0; i < src.size(); i++) v.push_back(Item{src[i], (uint32_t)(seg == 0 ? i : p.masserts.size() + i)});
        for (auto& it : v) {
            Assertion& s = it.s;
            if (s.first < 0) s.first += (int64_t)n;
            if (s.first < 0 || (uint64_t)s.first >= n) fail("air program: assertion step outside the trace");
            if (s.stride && (s.stride >= n || (uint64_t)s.first >= s.stride)) fail("air program: a periodic assertion needs first_step < stride < trace length");
        }
        std::stable_sort(v.begin(), v.end(), [](const Item& x, const Item& y) {
            return std::make_tuple(x.s.stride, x.s.first, x.s.col) < std::make_tuple(y.s.stride, y.s.first, y.s.col);
        });
        for (size_t i = 1; i < v.size(); i++)
            if (v[i].s.col == v[i - 1].s.col && v[i].s.stride == v[i - 1].s.stride && v[i].s.first == v[i - 1].s.first) fail("air program: two assertions on the same column and step");
        for (auto& it : v) {
            const Assertion& s = it.s;
            size_t j = 0;
            for (; j < in.bgroups.size(); j++) if (in.bgroups[j].stride == s.stride && in.bgroups[j].first == (uint64_t)s.first) break;
            if (j == in.bgroups.size()) {
                BoundaryGroup bg{};
                bg.stride = s.stride; bg.first = (uint64_t)s.first;
                bg.a = s.stride ? n / s.stride : 1;
                bg.b = gl::pow(g, bg.first * bg.a);
                bg.adj = (in.ce_n - 1 + bg.a) - (n - 1);
                in.bgroups.push_back(bg);
            }
            if (ref_kind(s.value) == K_SEQ) {
                // values at first + i * stride for i < n / stride: the interpolant of the values at x * w_n^-first (winter-air 0.4, boundary constraint "poly_offset")
                if ((uint64_t)p.sequences[ref_index(s.value)].size() * s.stride != n) fail("air program: a sequence assertion needs stride * number of values = trace length");
                BoundaryMember bm{s.col, (uint32_t)seg, coef++, (uint32_t)j, 0u, 0u};
                bm.seq = (int32_t)ref_index(s.value);
                in.members[it.id] = bm;
                in.bgroups[j].has_seq = true;
                continue;
            }
            const DOperand val = device_operand(p, s.value);     // row-independent: a scalar
            in.members[it.id] = BoundaryMember{s.col, (uint32_t)seg, coef++, (uint32_t)j, val.kind == D_SCAL_E ? 1u : 0u, val.idx};
        }
    }
    if (in.bgroups.size() > 64) fail("air program: more than 64 boundary divisors", ST_UNSUPPORTED);
    return in;
}

// ---- per proof: the scalar operands (constants, public inputs, random elements, folded nodes) --------------------------------
template <class F> struct Scalars {
    std::vector<uint64_t> b;
    std::vector<typename F::T> e;
};
template <class F> Scalars<F> fold_scalars(const Program& p, const uint64_t* pub, const typename F::T* rands) {
    typedef typename F::T T;
    Scalars<F> s;
    s.b.assign(p.n_scalB, 0); s.e.assign(p.n_scalE, F::zero());
    for (size_t i = 0; i < p.consts.size(); i++) s.b[i] = p.consts[i];
    for (uint32_t i = 0; i < p.num_pub; i++) s.b[p.consts.size() + i] = pub[i];
    for (uint32_t i = 0; i < p.R; i++) s.e[i] = rands ? rands[i] : F::zero();
    auto val_b = [&](uint32_t ref) { return s.b[device_operand(p, ref).idx]; };
    auto val_e = [&](uint32_t ref) { const DOperand o = device_operand(p, ref); return o.kind == D_SCAL_E ? s.e[o.idx] : F::from(s.b[o.idx]); };
    for (uint32_t i : p.fold_order) {
        const Node& nd = p.nodes[i];
        if (!p.is_ext[i]) {
            const uint64_t a = val_b(nd.a), b = val_b(nd.b);
            s.b[p.scalar_of[i]] = nd.op == 1 ? gl::add(a, b) : nd.op == 2 ? gl::sub(a, b) : gl::mul(a, b);
        } else {
            const T a = val_e(nd.a), b = val_e(nd.b);
            s.e[p.scalar_of[i]] = nd.op == 1 ? F::add(a, b) : nd.op == 2 ? F::sub(a, b) : F::mul(a, b);
        }
    }
    return s;
}

// ---- host transforms for the periodic columns -----------------------------------------------------------------------------------
inline void host_ntt(std::vector<uint64_t>& a, bool inverse) {     // natural order in and out, size a power of two
    const size_t n = a.size();
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    for (size_t i = 0; i < n; i++) { const size_t j = gl::bitrev((uint32_t)i, lg); if (i < j) std::swap(a[i], a[j]); }
    for (int s = 1; s <= lg; s++) {
        const size_t m = (size_t)1 << s;
        uint64_t wm = gl::root_of_unity(s);
        if (inverse) wm = gl::inv(wm);
        for (size_t k = 0; k < n; k += m) {
            uint64_t w = 1;
            for (size_t j = 0; j < m / 2; j++) {
                const uint64_t t = gl::mul(w, a[k + j + m / 2]), u = a[k + j];
                a[k + j] = gl::add(u, t); a[k + j + m / 2] = gl::sub(u, t);
                w = gl::mul(w, wm);
            }
        }
    }
    if (inverse) { const uint64_t ni = gl::inv(n); for (auto& v : a) v = gl::mul(v, ni); }
}
// Values of periodic column k on a domain of `rows` points x_s = h w_rows^s (rows a multiple of n): the column's value at x is
// P_k(x^(n/c)), P_k = interpolant of one cycle; x_s^(n/c) = h^(n/c) * (root of order rows c / n)^s, so the table has
// rows * c / n entries and is indexed by s mod that. (rows = n, h = 1: the cycle itself.)
inline std::vector<uint64_t> periodic_table(const std::vector<uint64_t>& cycle, uint64_t n, uint64_t rows, uint64_t h) {
    const uint64_t c = cycle.size(), period = rows / n * c;
    std::vector<uint64_t> co = cycle;
    host_ntt(co, true);
    const uint64_t hs = gl::pow(h, n / c);
    std::vector<uint64_t> t(period, 0);
    uint64_t sc = 1;
    for (uint64_t i = 0; i < c; i++) { t[i] = gl::mul(co[i], sc); sc = gl::mul(sc, hs); }
    host_ntt(t, false);
    return t;
}

// Coefficients (natural order) of the value polynomial of sequence `k` in the variable x: P(x w_n^-first) with P the interpolant of the
// values over the subgroup of their own size - i.e. coefficient j of P times w_n^(-first j).
inline std::vector<uint64_t> sequence_poly(const Program& p, uint32_t k, int log_n, uint64_t first) {
    std::vector<uint64_t> co = p.sequences[k];
    host_ntt(co, true);
    if (first) {
        const uint64_t gi = gl::inv(gl::pow(gl::root_of_unity(log_n), first));
        uint64_t sc = 1;
        for (auto& c : co) { c = gl::mul(c, sc); sc = gl::mul(sc, gi); }
    }
    return co;
}
// the same, kept per (sequence, trace length, first step) in the program handle: a prover asks for the same polynomials proof after proof
struct SeqPolyCache {
    std::mutex mu;
    std::map<std::tuple<uint32_t, int, uint64_t>, std::shared_ptr<const std::vector<uint64_t>>> polys;
};
inline std::shared_ptr<const std::vector<uint64_t>> sequence_poly_cached(const Program& p, uint32_t k, int log_n, uint64_t first) {
    static std::mutex create_mu;
    std::shared_ptr<SeqPolyCache> cache;
    {
        std::lock_guard<std::mutex> lk(create_mu);
        if (!p.seq_cache) p.seq_cache = std::make_shared<SeqPolyCache>();
        cache = std::static_pointer_cast<SeqPolyCache>(p.seq_cache);
    }
    std::lock_guard<std::mutex> lk(cache->mu);
    auto key = std::make_tuple(k, log_n, first);
    auto it = cache->polys.find(key);
    if (it == cache->polys.end()) {
        if (cache->polys.size() >= 64) cache->polys.clear();          // bounded: a prover works with a handful of trace lengths
        it = cache->polys.emplace(key, std::make_shared<const std::vector<uint64_t>>(sequence_poly(p, k, log_n, first))).first;
    }
    return it->second;            // shared: the map may be cleared by another thread of a pool while this polynomial is in use
}
// value of every sequence assertion at the point x (0 for the other members): what host_evaluate takes as seq_at_x
template <class F> std::vector<typename F::T> sequence_values_at(const Program& p, const Instance& in, typename F::T x) {
    std::vector<typename F::T> out(in.members.size(), F::zero());
    for (size_t m = 0; m < in.members.size(); m++) {
        const BoundaryMember& bm = in.members[m];
        if (bm.seq < 0) continue;
        const std::vector<uint64_t> co = sequence_poly(p, (uint32_t)bm.seq, in.log_n, in.bgroups[bm.group].first);
        typename F::T acc = F::zero();
        for (size_t i = co.size(); i-- > 0;) acc = F::add(F::mul(acc, x), F::from(co[i]));
        out[m] = acc;
    }
    return out;
}

// ---- host execution of the compiled constraint program over E (the verifier's out-of-domain check) ------------------------------
// Frame, periodic values and x are E-valued; `xpow(e)` returns x^e. Returns the numerator of every column.
template <class F>
std::vector<typename F::T> host_evaluate(const Program& p, const Instance& in, const Scalars<F>& sc, const typename F::T* cur, const typename F::T* nxt,
                                         const std::vector<typename F::T>& per, const std::vector<typename F::T>& ta, const std::vector<typename F::T>& tb,
                                         const std::vector<typename F::T>& ba, const std::vector<typename F::T>& bb,
                                         const std::function<typename F::T(uint64_t)>& xpow, const std::vector<typename F::T>& seq_at_x = {}) {
    typedef typename F::T T;
    std::vector<T> slotB(p.cons_slotsB, F::zero()), slotE(p.cons_slotsE, F::zero());
    auto fetch = [&](uint32_t kind, uint32_t idx) -> T {
        switch (kind) {
            case D_SLOT_B: return slotB[idx];
            case D_SLOT_E: return slotE[idx];
            case D_MAIN_CUR: return cur[idx];
            case D_MAIN_NXT: return nxt[idx];
            case D_AUX_CUR: return cur[p.W + idx];
            case D_AUX_NXT: return nxt[p.W + idx];
            case D_PERIODIC: return per[idx];
            case D_SCAL_B: return F::from(sc.b[idx]);
            default: return sc.e[idx];
        }
    };
    std::vector<T> out(in.num_columns(), F::zero());
    std::vector<T> gsa(in.bgroups.size(), F::zero()), gsb(in.bgroups.size(), F::zero());
    T acc_a = F::zero(), acc_b = F::zero(), total = F::zero();
    for (const Insn& I : p.cons_code) {
        const uint32_t op = I.op & 0xff, ka = (I.op >> 8) & 0xf, kb = (I.op >> 12) & 0xf;
        if (op == OP_END) break;
        switch (op) {
            case OP_ADD_B: slotB[I.dst] = F::add(fetch(ka, I.a), fetch(kb, I.b)); break;
            case OP_SUB_B: slotB[I.dst] = F::sub(fetch(ka, I.a), fetch(kb, I.b)); break;
            case OP_MUL_B: slotB[I.dst] = F::mul(fetch(ka, I.a), fetch(kb, I.b)); break;
            case OP_ADD_E: slotE[I.dst] = F::add(fetch(ka, I.a), fetch(kb, I.b)); break;
            case OP_SUB_E: slotE[I.dst] = F::sub(fetch(ka, I.a), fetch(kb, I.b)); break;
            case OP_MUL_E: case OP_MULB_E: slotE[I.dst] = F::mul(fetch(ka, I.a), fetch(kb, I.b)); break;
            case OP_EMIT_B: case OP_EMIT_E: {
                const T v = fetch(ka, I.a);
                acc_a = F::add(acc_a, F::mul(ta[I.dst], v));
                acc_b = F::add(acc_b, F::mul(tb[I.dst], v));
                break;
            }
            case OP_EMIT_ADD_B: case OP_EMIT_SUB_B: case OP_EMIT_MUL_B: {
                const T x = fetch(ka, I.a), y = fetch(kb, I.b);
                const T v = op == OP_EMIT_ADD_B ? F::add(x, y) : op == OP_EMIT_SUB_B ? F::sub(x, y) : F::mul(x, y);
                acc_a = F::add(acc_a, F::mul(ta[I.dst], v));
                acc_b = F::add(acc_b, F::mul(tb[I.dst], v));
                break;
            }
            case OP_EMIT3_B: {
                const uint32_t o1 = (I.op >> 16) & 3, o2 = (I.op >> 18) & 3, left = (I.op >> 20) & 1;
                auto ap = [&](uint32_t o, T x, T y) { return o == 1 ? F::add(x, y) : o == 2 ? F::sub(x, y) : F::mul(x, y); };
                const T inner = ap(o2, slotB[I.b], slotB[I.pad]), other = slotB[I.a];
                const T v = left ? ap(o1, inner, other) : ap(o1, other, inner);
                acc_a = F::add(acc_a, F::mul(ta[I.dst], v));
                acc_b = F::add(acc_b, F::mul(tb[I.dst], v));
                break;
            }
            case OP_GROUP_END: total = F::add(total, F::mul(acc_b, xpow(in.dgroup_adj[I.dst]))); acc_b = F::zero(); break;
            case OP_LOAD_MAIN: case OP_LOAD_AUX:
                for (uint32_t k = 0; k < I.a; k++) {
                    const uint32_t d = p.cons_desc[I.dst + k], col = d & 0xffff, slot = (d >> 16) & 0xfff;
                    const T* row = (d >> 31) ? nxt : cur;
                    if (op == OP_LOAD_AUX) slotE[slot] = row[p.W + col];
                    else slotB[slot] = (d & 0x40000000u) ? per[col] : row[col];
                }
                break;
            case OP_BOUND_B: case OP_BOUND_E: {
                const BoundaryMember& bm = in.members[I.dst];
                const T v = fetch(ka, I.a);
                gsa[bm.group] = F::add(gsa[bm.group], F::mul(ba[bm.coef], v));
                gsb[bm.group] = F::add(gsb[bm.group], F::mul(bb[bm.coef], v));
                break;
            }
            default: fail("air program: corrupt constraint code", ST_INTERNAL);
        }
    }
    out[0] = F::add(total, acc_a);
    // the assertions' values: sum (alpha + beta x^adj)(v - value) = [sa - sum alpha value] + x^adj [sb - sum beta value]
    for (size_t m = 0; m < in.members.size(); m++) {
        const BoundaryMember& bm = in.members[m];
        // sequence assertions: the value polynomial at the point (seq_at_x[member], see sequence_value_at)
        if (bm.seq >= 0 && seq_at_x.size() != in.members.size()) fail("air program: sequence values missing", ST_INTERNAL);
        const T val = bm.seq >= 0 ? seq_at_x[m] : bm.val_ext ? sc.e[bm.val_idx] : F::from(sc.b[bm.val_idx]);
        gsa[bm.group] = F::sub(gsa[bm.group], F::mul(ba[bm.coef], val));
        gsb[bm.group] = F::sub(gsb[bm.group], F::mul(bb[bm.coef], val));
    }
    for (size_t j = 0; j < in.bgroups.size(); j++) out[1 + j] = F::add(gsa[j], F::mul(gsb[j], xpow(in.bgroups[j].adj)));
    return out;
}

// ---- built-in AIRs as programs ------------------------------------------------------------------------------------------------
struct Writer {
    std::vector<uint8_t> out;
    void u32(uint32_t v) { for (int i = 0; i < 4; i++) out.push_back((uint8_t)(v >> (8 * i))); }
    void u64(uint64_t v) { for (int i = 0; i < 8; i++) out.push_back((uint8_t)(v >> (8 * i))); }
};
// Minimal expression builder used by the emitters below (hash-consing of nodes and constants)
struct Emitter {
    uint32_t W, A, R, num_pub, exemptions;
    std::vector<uint64_t> consts;
    std::map<uint64_t, uint32_t> const_idx;
    std::vector<std::vector<uint64_t>> periodic;
    std::vector<Node> nodes;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> node_idx;
    struct T { uint32_t root, base; std::vector<uint32_t> cycles; };
    std::vector<T> mtrans, atrans;
    std::vector<Assertion> masserts, aasserts;
    std::vector<Builder> builders;
    uint32_t cst(uint64_t v) {
        v %= gl::P;
        auto it = const_idx.find(v);
        if (it == const_idx.end()) { it = const_idx.insert({v, (uint32_t)consts.size()}).first; consts.push_back(v); }
        return mk_ref(K_CONST, it->second);
    }
    uint32_t node(uint32_t op, uint32_t a, uint32_t b) {
        if (op != 2 && b < a) std::swap(a, b);
        auto key = std::make_tuple(op, a, b);
        auto it = node_idx.find(key);
        if (it == node_idx.end()) { it = node_idx.insert({key, (uint32_t)nodes.size()}).first; nodes.push_back(Node{op, a, b}); }
        return mk_ref(K_NODE, it->second);
    }
    uint32_t add(uint32_t a, uint32_t b) { return node(1, a, b); }
    uint32_t sub(uint32_t a, uint32_t b) { return node(2, a, b); }
    uint32_t mul(uint32_t a, uint32_t b) { return node(3, a, b); }
    uint32_t pow(uint32_t a, uint32_t e) {
        uint32_t r = REF_NONE, base = a;
        while (e) {
            if (e & 1) r = r == REF_NONE ? base : mul(r, base);
            e >>= 1;
            if (e) base = mul(base, base);
        }
        return r;
    }
    uint32_t per(const std::vector<uint64_t>& cycle) { periodic.push_back(cycle); return mk_ref(K_PERIODIC, (uint32_t)periodic.size() - 1); }
    std::vector<uint8_t> bytes() const {
        Writer w;
        for (char ch : std::string("AEROAIR")) w.out.push_back((uint8_t)ch);
        w.out.push_back(1);
        const uint32_t h[16] = {W, A, R, num_pub, exemptions, (uint32_t)consts.size(), (uint32_t)periodic.size(), (uint32_t)nodes.size(),
                                (uint32_t)mtrans.size(), (uint32_t)atrans.size(), (uint32_t)masserts.size(), (uint32_t)aasserts.size(),
                                (uint32_t)builders.size(), 0, 0, 0};
        for (uint32_t v : h) w.u32(v);
        for (uint64_t v : consts) w.u64(v);
        for (auto& c : periodic) { w.u32((uint32_t)c.size()); for (uint64_t v : c) w.u64(v); }
        for (auto& n : nodes) { w.u32(n.op); w.u32(n.a); w.u32(n.b); }
        for (auto* v : {&mtrans, &atrans}) for (auto& t : *v) { w.u32(t.root); w.u32(t.base); w.u32((uint32_t)t.cycles.size()); for (uint32_t c : t.cycles) w.u32(c); }
        for (auto* v : {&masserts, &aasserts}) for (auto& s : *v) { w.u32(s.col); w.u32((uint32_t)(int32_t)s.first); w.u32(s.stride); w.u32(s.value); }
        for (auto& b : builders) { w.u32(b.init); w.u32(b.num); w.u32(b.den); }
        return w.out;
    }
};
// FibAir(width) [+ auxiliary segment]: the constraint set prover.hpp's FibAir hard-wires
inline std::vector<uint8_t> fib_program(uint32_t width, uint32_t aux_width, uint32_t aux_rands, uint32_t aux_degree) {
    if (width < 2 || (width & 1) || width > 254) fail("fib_program: even width in [2, 254]");
    if (aux_width && (aux_degree < 2 || aux_degree > 8 || aux_rands < 1 || aux_rands > 255 || aux_width > 255 - width)) fail("fib_program: bad auxiliary segment");
    Emitter e{};
    e.W = width; e.A = aux_width; e.R = aux_width ? aux_rands : 0; e.num_pub = width / 2; e.exemptions = 1;
    for (uint32_t k = 0; k < width / 2; k++) {
        const uint32_t a = mk_ref(K_MAIN_CUR, 2 * k), b = mk_ref(K_MAIN_CUR, 2 * k + 1), na = mk_ref(K_MAIN_NXT, 2 * k), nb = mk_ref(K_MAIN_NXT, 2 * k + 1);
        e.mtrans.push_back({e.sub(na, e.add(a, b)), 1, {}});
        e.mtrans.push_back({e.sub(nb, e.add(b, na)), 1, {}});
    }
    for (uint32_t c = 0; c < width; c++) e.masserts.push_back(Assertion{c, 0, 0, e.cst(1 + c)});
    for (uint32_t k = 0; k < width / 2; k++) e.masserts.push_back(Assertion{2 * k + 1, -1, 0, mk_ref(K_PUB, k)});
    for (uint32_t c = 0; c < aux_width; c++) {
        const uint32_t f = e.pow(e.add(mk_ref(K_RAND, c % aux_rands), mk_ref(K_MAIN_CUR, c % width)), aux_degree - 1);
        e.atrans.push_back({e.sub(mk_ref(K_AUX_NXT, c), e.mul(mk_ref(K_AUX_CUR, c), f)), aux_degree, {}});
        e.aasserts.push_back(Assertion{c, 0, 0, e.cst(1)});
        e.builders.push_back(Builder{e.cst(1), f, REF_NONE});
    }
    return e.bytes();
}


// ---- a synthetic AIR in the SHAPE of a VM's, with a trace that satisfies it (include/aero_air.h: aero_air_synth_vm_*) -----------------
// Miden's ProcessorAir is absent from the reference mount; this is the stand-in for BASELINE configs[4] that
// goes through the program path: 20 + 2 * pairs main columns - clock, 4-bit binary counter, its low 3 bits as a number, 8 state
// columns under power maps of degree 2..7 gated by a periodic selector with periodic round constants (cycle 8), 4 accumulators of
// degree 5..8, two columns that are cyclic shifts of state columns (permutation arguments), Fibonacci pairs; `aux` auxiliary running
// products (two of them with denominators, periodic gating, boundary values that depend on the random elements); two transition
// exemptions; assertions at the first, last, an interior step and periodic ones. 24 + 2 * pairs main and `aux` auxiliary transition
// constraints in 10+ degree groups; 8 composition columns. tests/air_examples.py: synth_vm builds the SAME system in Python.
struct SynthVm {
    static constexpr uint32_t CLK = 0, BIT = 1, M8 = 5, S = 6, ACC = 14, B1 = 18, B2 = 19, FIB = 20;
    static uint64_t sel(uint32_t i) { static const uint64_t v[8] = {1, 1, 1, 0, 1, 0, 0, 1}; return v[i & 7]; }
    static uint64_t rc(uint32_t j, uint32_t i) {
        const unsigned __int128 v = (unsigned __int128)0x9E3779B97F4A7C15ull * (8 * j + (i & 7) + 1) + 12345;
        return (uint64_t)(v % gl::P);
    }
    static uint32_t deg_s(uint32_t j) { static const uint32_t d[8] = {2, 3, 4, 5, 6, 7, 2, 3}; return d[j]; }
    static uint32_t deg_acc(uint32_t k) { return 5 + k; }
    static uint32_t width(uint32_t pairs) { return 20 + 2 * pairs; }
};
// column-major width x 2^log_n; pub (pairs + 1 values): the Fibonacci results, then the last value of the degree-8 accumulator
inline void synth_vm_trace(uint32_t log_n, uint32_t pairs, uint64_t* t, uint64_t* pub) {
    typedef SynthVm V;
    const size_t n = (size_t)1 << log_n;
    auto col = [&](uint32_t c) { return t + (size_t)c * n; };
    for (uint32_t j = 0; j < 8; j++) col(V::S + j)[0] = 3 + j;
    for (uint32_t k = 0; k < 4; k++) col(V::ACC + k)[0] = 11 + k;
    for (uint32_t k = 0; k < pairs; k++) { col(V::FIB + 2 * k)[0] = 1 + 2 * k; col(V::FIB + 2 * k + 1)[0] = 2 + 2 * k; }
    for (size_t i = 0; i < n; i++) {
        col(V::CLK)[i] = i;
        for (uint32_t q = 0; q < 4; q++) col(V::BIT + q)[i] = (i >> q) & 1;
        col(V::M8)[i] = i & 7;
        if (i + 1 == n) break;
        uint64_t s[8];
        for (uint32_t j = 0; j < 8; j++) s[j] = col(V::S + j)[i];
        for (uint32_t j = 0; j < 8; j++)
            col(V::S + j)[i + 1] = V::sel((uint32_t)i) ? gl::add(gl::pow(s[j], V::deg_s(j)), V::rc(j, (uint32_t)i)) : gl::add(s[j], s[(j + 1) & 7]);
        for (uint32_t k = 0; k < 4; k++) col(V::ACC + k)[i + 1] = gl::add(gl::pow(col(V::ACC + k)[i], V::deg_acc(k)), s[k]);
        for (uint32_t k = 0; k < pairs; k++) {
            const uint64_t a = col(V::FIB + 2 * k)[i], b = col(V::FIB + 2 * k + 1)[i], na = gl::add(a, b);
            col(V::FIB + 2 * k)[i + 1] = na;
            col(V::FIB + 2 * k + 1)[i + 1] = gl::add(b, na);
        }
    }
    for (size_t i = 0; i + 1 < n; i++) {      // cyclic shift by one step over the first n - 1 rows
        col(V::B1)[i] = col(V::S)[(i + 1) % (n - 1)];
        col(V::B2)[i] = col(V::S + 5)[(i + 1) % (n - 1)];
    }
    col(V::B1)[n - 1] = 0; col(V::B2)[n - 1] = 0;
    if (pub) {
        for (uint32_t k = 0; k < pairs; k++) pub[k] = col(V::FIB + 2 * k + 1)[n - 1];
        pub[pairs] = col(V::ACC + 3)[n - 1];
    }
}
// the program for a trace of 2^log_n rows (an interior assertion sits at step n / 2, two values are read off the trace's first row)
inline std::vector<uint8_t> synth_vm_program(uint32_t log_n, uint32_t pairs, uint32_t aux, uint32_t rands) {
    typedef SynthVm V;
    if (log_n < 4 || log_n > 29 || pairs < 1 || V::width(pairs) > 254 || aux > 255 - V::width(pairs) || (aux && (rands < 3 || rands > 255)))
        fail("synth_vm_program: log_n in [4, 29], 1 <= pairs <= 117, an auxiliary segment needs at least 3 random elements");
    const uint64_t n = 1ull << log_n;
    Emitter e{};
    e.W = V::width(pairs); e.A = aux; e.R = aux ? rands : 0; e.num_pub = pairs + 1; e.exemptions = 2;
    auto cur = [](uint32_t c) { return mk_ref(K_MAIN_CUR, c); };
    auto nxt = [](uint32_t c) { return mk_ref(K_MAIN_NXT, c); };
    std::vector<uint64_t> selv(8), rcv(8);
    for (uint32_t i = 0; i < 8; i++) selv[i] = V::sel(i);
    const uint32_t selp = e.per(selv);
    uint32_t rcp[8];
    for (uint32_t j = 0; j < 8; j++) { for (uint32_t i = 0; i < 8; i++) rcv[i] = V::rc(j, i); rcp[j] = e.per(rcv); }
    const uint32_t one = e.cst(1), two = e.cst(2);
    auto T = [&](uint32_t root, uint32_t base, std::vector<uint32_t> cyc = {}) { e.mtrans.push_back({root, base, cyc}); };
    T(e.sub(e.sub(nxt(V::CLK), cur(V::CLK)), one), 1);
    uint32_t bits[4];
    for (uint32_t i = 0; i < 4; i++) { bits[i] = cur(V::BIT + i); T(e.mul(bits[i], e.sub(bits[i], one)), 2); }
    uint32_t carry = REF_NONE;
    for (uint32_t i = 0; i < 4; i++) {                        // bit_i' = bit_i XOR (bit_0 ... bit_(i-1))
        const uint32_t t = carry == REF_NONE ? one : carry;
        T(e.sub(nxt(V::BIT + i), e.sub(e.add(bits[i], t), e.mul(two, e.mul(bits[i], t)))), i ? i + 1 : 1);
        carry = carry == REF_NONE ? bits[i] : e.mul(carry, bits[i]);
    }
    T(e.sub(cur(V::M8), e.add(e.add(bits[0], e.mul(two, bits[1])), e.mul(e.cst(4), bits[2]))), 1);
    uint32_t s[8];
    for (uint32_t j = 0; j < 8; j++) s[j] = cur(V::S + j);
    const uint32_t nsel = e.sub(one, selp);
    for (uint32_t j = 0; j < 8; j++)                          // s_j' = sel (s_j^d + rc_j) + (1 - sel)(s_j + s_(j+1))
        T(e.sub(nxt(V::S + j), e.add(e.mul(selp, e.add(e.pow(s[j], V::deg_s(j)), rcp[j])), e.mul(nsel, e.add(s[j], s[(j + 1) & 7])))), V::deg_s(j), {8});
    for (uint32_t k = 0; k < 4; k++) T(e.sub(nxt(V::ACC + k), e.add(e.pow(cur(V::ACC + k), V::deg_acc(k)), s[k])), V::deg_acc(k));
    T(e.sub(cur(V::B1), nxt(V::S)), 1);
    T(e.sub(cur(V::B2), nxt(V::S + 5)), 1);
    for (uint32_t k = 0; k < pairs; k++) {
        const uint32_t a = cur(V::FIB + 2 * k), b = cur(V::FIB + 2 * k + 1), na = nxt(V::FIB + 2 * k), nb = nxt(V::FIB + 2 * k + 1);
        T(e.sub(na, e.add(a, b)), 1);
        T(e.sub(nb, e.add(b, na)), 1);
    }
    auto Am = [&](uint32_t c, int64_t step, uint32_t stride, uint32_t val) { e.masserts.push_back(Assertion{c, step, stride, val}); };
    Am(V::CLK, 0, 0, e.cst(0));
    Am(V::CLK, (int64_t)(n / 2), 0, e.cst(n / 2));
    Am(V::B1, -2, 0, e.cst(3));                               // where the cyclic shift wraps: s_0(0) = 3, s_5(0) = 8
    Am(V::B2, -2, 0, e.cst(8));
    for (uint32_t j = 0; j < 8; j++) Am(V::S + j, 0, 0, e.cst(3 + j));
    for (uint32_t k = 0; k < 4; k++) Am(V::ACC + k, 0, 0, e.cst(11 + k));
    Am(V::ACC + 3, -1, 0, mk_ref(K_PUB, pairs));
    for (uint32_t k = 0; k < pairs; k++) {
        Am(V::FIB + 2 * k, 0, 0, e.cst(1 + 2 * k));
        Am(V::FIB + 2 * k + 1, 0, 0, e.cst(2 + 2 * k));
        Am(V::FIB + 2 * k + 1, -1, 0, mk_ref(K_PUB, k));
    }
    Am(V::M8, 0, 8, e.cst(0));
    Am(V::M8, 3, 8, e.cst(3));
    Am(V::BIT, 1, 2, one);
    if (aux) {
        auto r = [&](uint32_t i) { return mk_ref(K_RAND, i % rands); };
        auto ac = [](uint32_t c) { return mk_ref(K_AUX_CUR, c); };
        auto an = [](uint32_t c) { return mk_ref(K_AUX_NXT, c); };
        // p0: permutation argument s_0 <-> b1 over the first n - 1 rows: p' (r0 + b1) = p (r0 + s_0); it returns to 1
        {
            const uint32_t num = e.add(r(0), s[0]), den = e.add(r(0), cur(V::B1));
            e.atrans.push_back({e.sub(e.mul(an(0), den), e.mul(ac(0), num)), 2, {}});
            e.builders.push_back(Builder{one, num, den});
            e.aasserts.push_back(Assertion{0, 0, 0, one});
            e.aasserts.push_back(Assertion{0, -1, 0, one});
        }
        for (uint32_t c = 1; c < aux; c++) {
            if (c == 1) {                                     // a second argument with a denominator, tuples compressed with r1, r2
                const uint32_t rc2 = e.mul(r(2), cur(V::CLK));
                const uint32_t num = e.add(e.add(r(1), s[5]), rc2), den = e.add(e.add(r(1), cur(V::B2)), rc2);
                const uint32_t init = e.add(e.mul(r(0), r(0)), one);
                e.atrans.push_back({e.sub(e.mul(an(c), den), e.mul(ac(c), num)), 2, {}});
                e.builders.push_back(Builder{init, num, den});
                e.aasserts.push_back(Assertion{c, 0, 0, init});        // a boundary value that depends on the random elements
                continue;
            }
            const uint32_t ex = 1 + c % 3;
            const uint32_t colv = e.add(s[(c + 1) & 7], e.mul(r(c + 1), cur(V::CLK)));
            uint32_t f = e.pow(e.add(r(c), colv), ex);
            if (c % 2 == 0) {                                 // gated by the periodic selector
                f = e.add(e.mul(selp, f), nsel);
                e.atrans.push_back({e.sub(an(c), e.mul(ac(c), f)), ex + 1, {8}});
            } else {
                e.atrans.push_back({e.sub(an(c), e.mul(ac(c), f)), ex + 1, {}});
            }
            e.builders.push_back(Builder{one, f, REF_NONE});
            e.aasserts.push_back(Assertion{c, 0, 0, one});
        }
    }
    return e.bytes();
}

}  // namespace air
}  // namespace aero

// the opaque handle of include/aero_air.h
struct aero_air {
    aero::air::Program prog;
    std::vector<uint8_t> bytes;
};
