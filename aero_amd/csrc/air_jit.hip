// AIR programs compiled at run time (the fast path of include/aero_air.h's constraint evaluation).
//
// The interpreter (air_kernels.hip) decodes the compiled program once per instruction and per wave and keeps the register file of
// the expression DAG in LDS; on a VM-sized constraint system that costs 2 - 3x the arithmetic itself (profiles/r3j_air_pmc_*.csv:
// 38 % of a wave's cycles issue VALU work, the rest is scalar decode, LDS round trips and their waits at 1.5 waves per SIMD).
// An AIR is fixed for the lifetime of a prover, so the program is turned into a HIP kernel instead: straight-line code over the
// expression DAG (every node a local variable, constants as literals, scalar operands and composition coefficients as uniform loads
// from the per-proof pool), compiled for gfx950 with hiprtc when the first proof of a (program, field, boundary structure) arrives
// and cached - in the aero_air handle (code object) and in the context (loaded module). The compiler does what the interpreter
// cannot: register allocation over the whole DAG, load scheduling, common sub-expressions across constraints.
//
// The arithmetic is the interpreter's, in another order: all of it is exact field arithmetic, so H(x) - and with it every proof
// byte - is identical (tests/test_gpu_air.py runs both paths against the oracle). AERO_AIR_JIT=0 selects the interpreter; it also
// takes over when hiprtc is not available on the machine or the generated kernel does not compile (the reason is kept in
// air_jit_last_error()). Reference seam: `ConstraintEvaluator::new(&air, aux_rand_elements, &constraint_coeffs)` ->
// `evaluate_fragment` (aero-sdk/miden-wasm/src/constraints_worker.rs:32-59, proving_worker.rs:374-437): Winterfell monomorphises
// the evaluator over the AIR at compile time; this is the same specialisation, made when the AIR arrives as data.
#include "air_host.hpp"
#include "air_kernels.hpp"

#include <hip/hiprtc.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <set>
#include <sstream>

namespace aero {
namespace {

const char* const GL_FIELD_SRC =
#include "gl_field_src.inc"
    ;

struct AirJitArgs {      // mirrored in the generated source; every member 8 bytes
    const uint64_t *lde, *aux;
    uint64_t N, blowup, ce_step, split_log, rows, first, count;
    const uint64_t* pool;
    uint64_t oSE, oT, oB;
    const uint64_t *ptab, *pdesc, *dg_exp;
    const AirBGroupDev* bgroups;
    const uint64_t *gA, *gB, *tw_lo, *tw_hi;
    uint64_t tw_h, offset;
    const uint64_t* zn_inv;
    uint64_t xmask;
    const uint64_t* exempt;
    uint64_t *out_cols, *out_h0, *out_h1;
    const uint64_t* seq_tab;
};
const char* const ARGS_SRC = R"SRC(
struct BG { uint64_t a_exp, ha, b, adj_exp, seq; };
struct Args {
    const uint64_t *lde, *aux;
    uint64_t N, blowup, ce_step, split_log, rows, first, count;
    const uint64_t* pool;
    uint64_t oSE, oT, oB;
    const uint64_t *ptab, *pdesc, *dg_exp;
    const BG* bgroups;
    const uint64_t *gA, *gB, *tw_lo, *tw_hi;
    uint64_t tw_h, offset;
    const uint64_t* zn_inv;
    uint64_t xmask;
    const uint64_t* exempt;
    uint64_t *out_cols, *out_h0, *out_h1;
    const uint64_t* seq_tab;
};
__device__ __forceinline__ uint64_t tw(const Args& a, uint64_t e) { return gl::mul(a.tw_lo[e & ((1ull << a.tw_h) - 1)], a.tw_hi[e >> a.tw_h]); }
using gl::Wide; using gl::wzero; using gl::wmac; using gl::wreduce;      // 160-bit sums of products (gl_field.hpp)
)SRC";

struct Blob {
    uint64_t id;
    std::vector<char> code;
    std::string error;       // non-empty: compilation failed, the interpreter runs instead
    std::string cache_file;  // non-empty: the code object was read from this cache entry
};
struct Cache {
    std::mutex mu;
    std::map<std::string, std::shared_ptr<Blob>> blobs;
};
std::mutex g_mu;
std::mutex g_rtc_mu;          // hiprtc / comgr and hipModuleLoadData are entered by one thread at a time (pool workers and thread ranks meet their
                              // first proof of a program together; neither library promises re-entrancy)
std::string g_last_error;
std::atomic<uint64_t> g_next_id{1};

using namespace air;

// Code-generation switches (AERO_AIR_JIT_TUNE="key=value,..." for experiments). Defaults = what measured best on MI355X over FibAir(72),
// FibAir(72) + 9 aux columns of degree 8 and the VM-shaped program (tools/air_jit_variants.sh; profiles/r3m_air_jit_variants.jsonl,
// r3l_* with strided rows): on the VM-shaped program 2^20 x (72 + 9) the kernel takes 5.7 ms with the defaults, 11.0 ms with early = 0
// (all 72 current-row values stay live until the assertions: 336 VGPRs, one wave per SIMD), 9.3 ms with 4 rows and no barrier.
struct Tune {
    int barrier = 0;       // 1: per row, hide the pool pointer behind an empty asm so that the uniform loads of the coefficients are not hoisted out of the row loop
    int early = 1;         // a column's assertions are evaluated where its current-row value is first loaded, not after all transition constraints
    int minblocks = 2;     // __launch_bounds__(256, minblocks)
    int rows = 0;          // rows per thread of the fused form; 0 = by program size: 4 for small programs (the inversion dominates), 2 otherwise
    int wide = 1;          // products coefficient x base-field value are summed as 160-bit integers and reduced once per sum (4 multiply-adds + 5
                           // carry adds per term instead of a whole field multiplication + addition)
};
Tune read_tune() {
    Tune t;
    const char* e = getenv("AERO_AIR_JIT_TUNE");
    if (!e) return t;
    std::string str(e);
    size_t pos = 0;
    while (pos < str.size()) {
        const size_t c = str.find(',', pos), eq = str.find('=', pos);
        const size_t end = c == std::string::npos ? str.size() : c;
        if (eq != std::string::npos && eq < end) {
            const std::string k = str.substr(pos, eq - pos);
            const int v = atoi(str.substr(eq + 1, end - eq - 1).c_str());
            if (k == "barrier") t.barrier = v; else if (k == "early") t.early = v; else if (k == "minblocks") t.minblocks = v; else if (k == "rows") t.rows = v; else if (k == "wide") t.wide = v;
        }
        pos = end + 1;
    }
    return t;
}
const Tune g_tune = read_tune();

// ---- source generation -----------------------------------------------------------------------------------------------------------
struct Gen {
    const Program& p;
    const Instance& in;
    int DEG, mode;
    std::ostringstream o;
    std::vector<uint8_t> node_done;
    std::set<std::string> defined;
    std::vector<std::vector<uint32_t>> asserts_main, asserts_aux;      // member ids by column
    std::vector<uint8_t> member_done;

    Gen(const Program& p_, const Instance& in_, int deg, int m) : p(p_), in(in_), DEG(deg), mode(m), node_done(p_.nodes.size(), 0) {
        asserts_main.resize(p.W); asserts_aux.resize(p.A); member_done.assign(in.members.size(), 0);
        for (uint32_t i = 0; i < in.members.size(); i++) (in.members[i].aux ? asserts_aux : asserts_main)[in.members[i].col].push_back(i);
    }
    void member(uint32_t m, const std::string& v) {
        if (member_done[m]) return;
        member_done[m] = 1;
        const BoundaryMember& bm = in.members[m];
        if (!bm.aux && use_wide()) {
            for (int d = 0; d < DEG; d++) {
                o << "    wmac(wsa" << bm.group << "_" << d << ", pool[oB + " << (2 * m) * DEG + d << "], " << v << ");\n";
                o << "    wmac(wsb" << bm.group << "_" << d << ", pool[oB + " << (2 * m + 1) * DEG + d << "], " << v << ");\n";
            }
            return;
        }
        const char* mf = bm.aux ? "F::mul" : "F::mulb";
        o << "    sa" << bm.group << " = F::add(sa" << bm.group << ", " << mf << "(" << pool_e("oB", 2 * m) << ", " << v << "));\n";
        o << "    sb_" << bm.group << " = F::add(sb_" << bm.group << ", " << mf << "(" << pool_e("oB", 2 * m + 1) << ", " << v << "));\n";
    }
    bool use_wide() const { return g_tune.wide && mode <= 1; }
    std::string wide_value(const std::string& name) const {       // the E element whose components were summed in <name>_0 (, <name>_1)
        return "F::make(wreduce(" + name + "_0), " + (DEG > 1 ? "wreduce(" + name + "_1))" : std::string("0)"));
    }

    static std::string lit(uint64_t v) { char b[40]; snprintf(b, sizeof b, "0x%llxull", (unsigned long long)v); return b; }
    std::string pool_e(const std::string& base, uint64_t idx) {      // an element of E at pool[base + idx * DEG]
        std::ostringstream s;
        s << "F::make(pool[" << base << " + " << idx * DEG << "], " << (DEG > 1 ? "pool[" + base + " + " + std::to_string(idx * DEG + 1) + "]" : std::string("0")) << ")";
        return s.str();
    }
    void define(const std::string& name, const std::string& type, const std::string& expr) {
        if (defined.insert(name).second) o << "    const " << type << " " << name << " = " << expr << ";\n";
    }
    // frame values and scalars are named after what they are and defined where they are first needed
    std::string main_val(uint32_t c, bool next) {
        const std::string n = (next ? "mn" : "mc") + std::to_string(c);
        const bool fresh = !defined.count(n);
        define(n, "uint64_t", "a.lde[" + std::to_string(c) + "ull * a.N + " + (next ? "rn" : "r") + "]");
        if (fresh && !next && g_tune.early) for (uint32_t m : asserts_main[c]) member(m, n);
        return n;
    }
    std::string aux_val(uint32_t c, bool next) {
        const std::string n = (next ? "an" : "ac") + std::to_string(c), row = next ? "rn" : "r";
        std::string e = "F::make(a.aux[" + std::to_string((uint64_t)c * DEG) + "ull * a.N + " + row + "], ";
        e += DEG > 1 ? "a.aux[" + std::to_string((uint64_t)c * DEG + 1) + "ull * a.N + " + row + "])" : "0)";
        const bool fresh = !defined.count(n);
        define(n, "T", e);
        if (fresh && !next && g_tune.early) for (uint32_t m : asserts_aux[c]) member(m, n);
        return n;
    }
    std::string periodic_val(uint32_t k) {
        const std::string n = "pv" + std::to_string(k);
        if (!defined.count(n)) {
            o << "    const uint64_t pd" << k << " = a.pdesc[" << k << "];\n";
            define(n, "uint64_t", "a.ptab[(pd" + std::to_string(k) + " & 0xffffffffull) + (s & (pd" + std::to_string(k) + " >> 32))]");
        }
        return n;
    }
    std::string scal_b(uint32_t idx) {
        const std::string n = "sb" + std::to_string(idx);
        define(n, "uint64_t", "pool[" + std::to_string(idx) + "]");
        return n;
    }
    std::string scal_e(uint32_t idx) {
        const std::string n = "se" + std::to_string(idx);
        define(n, "T", pool_e("oSE", idx));
        return n;
    }
    // operand as (expression, is_ext)
    std::pair<std::string, bool> operand(uint32_t ref) {
        const uint32_t k = ref_kind(ref), i = ref_index(ref);
        switch (k) {
            case K_NODE:
                if (p.scalar_of[i] >= 0) return p.is_ext[i] ? std::make_pair(scal_e((uint32_t)p.scalar_of[i]), true) : std::make_pair(scal_b((uint32_t)p.scalar_of[i]), false);
                return {(p.is_ext[i] ? "e" : "t") + std::to_string(i), p.is_ext[i] != 0};
            case K_MAIN_CUR: return {main_val(i, false), false};
            case K_MAIN_NXT: return {main_val(i, true), false};
            case K_AUX_CUR: return {aux_val(i, false), true};
            case K_AUX_NXT: return {aux_val(i, true), true};
            case K_PERIODIC: return {periodic_val(i), false};
            case K_CONST: return {lit(p.consts[i]), false};
            case K_PUB: return {scal_b((uint32_t)p.consts.size() + i), false};
            default: return {scal_e(i), true};
        }
    }
    // the nodes an expression needs, not yet emitted, in index order (operands precede their users in a program)
    void need(uint32_t root) {
        if (ref_kind(root) != K_NODE) return;
        std::vector<uint32_t> stack{ref_index(root)}, todo;
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            if (node_done[i] || p.scalar_of[i] >= 0) continue;
            node_done[i] = 1;
            todo.push_back(i);
            for (uint32_t r : {p.nodes[i].a, p.nodes[i].b})
                if (ref_kind(r) == K_NODE) stack.push_back(ref_index(r));
        }
        std::sort(todo.begin(), todo.end());
        for (uint32_t i : todo) {
            const Node& nd = p.nodes[i];
            const auto x = operand(nd.a), y = operand(nd.b);
            const char* fn = nd.op == 1 ? "add" : nd.op == 2 ? "sub" : "mul";
            if (!p.is_ext[i]) {
                o << "    const uint64_t t" << i << " = gl::" << fn << "(" << x.first << ", " << y.first << ");\n";
            } else if (nd.op == 3 && x.second != y.second) {       // E x base
                o << "    const T e" << i << " = F::mulb(" << (x.second ? x.first : y.first) << ", " << (x.second ? y.first : x.first) << ");\n";
            } else {
                const std::string xe = x.second ? x.first : "F::from(" + x.first + ")", ye = y.second ? y.first : "F::from(" + y.first + ")";
                o << "    const T e" << i << " = F::" << fn << "(" << xe << ", " << ye << ");\n";
            }
        }
    }

    // R rows per thread (R chunks of 256 consecutive rows per workgroup): the boundary divisors of a row are folded into ONE fraction N / D
    // (N = sum_j num_j prod_(i != j) d_i, D = prod_j d_j), and a thread inverts the D of its R rows with one field inversion
    // (~95 multiplications, as much as a few hundred constraint terms). The rows run through a real loop - the body is emitted
    // once - and wait for the inversion in LDS (2 DEG + 1 words per row and thread).
    // mode 2: `Trace::validate(&air)` of a debug-mode Winterfell prover (the "(debug) validate" of commit_to_trace_and_validate,
    // proving_worker.rs:323-332) on the TRACE domain: every transition constraint on every row but the exempted last ones, every
    // assertion on the steps it names; the smallest (row << 24 | id) that fails is left in out_h0[0] (id = transition index, or
    // 0x800000 | assertion index in the program's order, main first; the loader admits at most 2^20 of either, rows < 2^29).
    std::string validate_source(size_t args_size) {
        o << GL_FIELD_SRC << "\n" << ARGS_SRC;
        o << "static_assert(sizeof(Args) == " << args_size << ", \"argument block\");\n";
        o << "typedef gl::" << (DEG == 1 ? "FB" : "FQ") << " F;\ntypedef F::T T;\nconstexpr int DEG = " << DEG << ";\n";
        o << "__device__ __forceinline__ bool nz(uint64_t v) { return v != 0; }\n__device__ __forceinline__ bool nz(gl::E2 v) { return (v.a0 | v.a1) != 0; }\n";
        o << "__device__ __forceinline__ bool ne(uint64_t a, uint64_t b) { return a != b; }\n__device__ __forceinline__ bool ne(gl::E2 a, gl::E2 b) { return a.a0 != b.a0 || a.a1 != b.a1; }\n";
        o << "extern \"C\" __global__ __launch_bounds__(256) void air_jit_kernel(Args a) {\n";
        o << "    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;\n    if (s >= a.count) return;\n";
        o << "    const uint64_t* pool = a.pool;\n    const uint64_t oSE = a.oSE;\n";
        o << "    const size_t r = s, rn = (s + 1) & (a.N - 1);\n";
        o << "    unsigned long long bad = ~0ull;\n";
        const bool early = g_tune.early;
        for (auto& v : asserts_main) v.clear();          // assertions are checked explicitly below, not accumulated
        for (auto& v : asserts_aux) v.clear();
        (void)early;
        o << "    if (s + " << p.exemptions << " < a.count) {\n";
        for (size_t k = 0; k < (mode == 3 ? (size_t)p.n_main_trans : p.num_transition()); k++) {
            need(p.trans[k].root);
            const auto v = operand(p.trans[k].root);
            o << "    if (nz(" << v.first << ")) { const unsigned long long e = (s << 24) | " << k << "ull; bad = e < bad ? e : bad; }\n";
        }
        o << "    }\n";
        for (uint32_t m = 0; m < in.members.size(); m++) {
            const BoundaryMember& bm = in.members[m];
            if (mode == 3 && bm.aux) continue;
            const BoundaryGroup& bg = in.bgroups[bm.group];
            std::string cond = bg.stride ? "(s & " + std::to_string(bg.stride - 1) + ") == " + std::to_string(bg.first) : "s == " + std::to_string(bg.first);
            o << "    if (" << cond << ") {\n";
            const std::string col = bm.aux ? "F::make(a.aux[" + std::to_string((uint64_t)bm.col * DEG) + "ull * a.N + r], " +
                                                 (DEG > 1 ? "a.aux[" + std::to_string((uint64_t)bm.col * DEG + 1) + "ull * a.N + r])" : std::string("0)"))
                                           : "a.lde[" + std::to_string(bm.col) + "ull * a.N + r]";
            std::string want;
            if (bm.seq >= 0) {
                // raw values behind the periodic tables (air_validate_trace): entry (s - first) / stride
                int lgs = 0; while ((1u << lgs) < bg.stride) lgs++;
                const size_t pd = std::max<size_t>(1, p.periodic.size()) + (size_t)bm.seq;
                want = "a.ptab[(a.pdesc[" + std::to_string(pd) + "] & 0xffffffffull) + ((s - " + std::to_string(bg.first) + "ull) >> " + std::to_string(lgs) + ")]";
                if (bm.aux) want = "F::from(" + want + ")";
            }
            else if (bm.val_ext) want = pool_e("oSE", bm.val_idx);
            else want = bm.aux ? "F::from(pool[" + std::to_string(bm.val_idx) + "])" : "pool[" + std::to_string(bm.val_idx) + "]";
            o << "        if (ne(" << col << ", " << want << ")) { const unsigned long long e = (s << 24) | " << (0x800000u | m) << "ull; bad = e < bad ? e : bad; }\n    }\n";
        }
        o << "    if (bad != ~0ull) atomicMin((unsigned long long*)a.out_h0, bad);\n}\n";
        return o.str();
    }

    std::string source(size_t args_size, int R) {
        if (mode >= 2) return validate_source(args_size);
        const size_t nt = p.num_transition(), ng = in.bgroups.size(), ndg = p.dgroups.size();
        const bool batch = mode == 1 && ng > 0;
        if (!batch) R = 1;
        o << GL_FIELD_SRC << "\n" << ARGS_SRC;
        o << "static_assert(sizeof(Args) == " << args_size << ", \"argument block\");\n";
        o << "typedef gl::" << (DEG == 1 ? "FB" : "FQ") << " F;\ntypedef F::T T;\nconstexpr int R = " << R << ", DEG = " << DEG << ";\n";
        o << "extern \"C\" __global__ __launch_bounds__(256, " << g_tune.minblocks << ") void air_jit_kernel(Args a) {\n";
        if (batch) o << "    __shared__ uint64_t sh_h[R][DEG][256], sh_n[R][DEG][256], sh_d[R][256];\n";
        // a workgroup's R x 256 rows are consecutive (R chunks of 256): its iterations stay on the same pages of every column
        o << "    const size_t t0 = (size_t)blockIdx.x * (256 * R) + threadIdx.x;\n    if (t0 >= a.count) return;\n";
        o << "    const uint64_t rmask = a.rows - 1;\n";
        o << "#pragma unroll 1\n    for (int q = 0; q < R; q++) {\n";
        o << "    const uint64_t s = a.first + t0 + (uint64_t)q * 256;\n";
        o << "    const uint64_t* pool = a.pool;\n    uint64_t oSE = a.oSE, oT = a.oT, oB = a.oB;\n";
        if (g_tune.barrier) o << "    asm volatile(\"\" : \"+s\"(pool), \"+s\"(oSE), \"+s\"(oT), \"+s\"(oB));\n";
        o << "    size_t r = (size_t)s * a.ce_step, rn = (r + a.blowup) & (a.N - 1);\n";
        o << "    if (a.split_log) { const size_t pl = a.N >> a.split_log, pm = ((size_t)1 << a.split_log) - 1; r = (r & pm) * pl + (r >> a.split_log); rn = (rn & pm) * pl + (rn >> a.split_log); }\n";
        o << "    T acc = F::zero();\n";
        for (size_t g = 0; g < ndg; g++) o << "    T gb" << g << " = F::zero();\n";
        for (size_t j = 0; j < ng; j++) o << "    T sa" << j << " = F::zero(), sb_" << j << " = F::zero();\n";
        const bool wide = use_wide();
        if (wide) {
            for (int d = 0; d < DEG; d++) o << "    Wide wacc_" << d << " = wzero();\n";
            for (size_t j = 0; j < ng; j++)
                for (int d = 0; d < DEG; d++) o << "    Wide wsa" << j << "_" << d << " = wzero(), wsb" << j << "_" << d << " = wzero();\n";
        }
        // transition constraints: acc += alpha c, gb[group] += beta' c; one degree group after the other (a single wide gb is live)
        std::vector<size_t> order(nt);
        for (size_t k = 0; k < nt; k++) order[k] = k;
        if (wide) std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return p.trans[x].group < p.trans[y].group; });
        for (size_t q = 0; q < nt; q++) {
            const size_t k = order[q];
            const uint32_t g = p.trans[k].group;
            need(p.trans[k].root);
            const auto v = operand(p.trans[k].root);
            if (wide && (q == 0 || p.trans[order[q - 1]].group != g))
                for (int d = 0; d < DEG; d++) o << "    Wide wgb" << g << "_" << d << " = wzero();\n";
            if (wide && !v.second) {
                for (int d = 0; d < DEG; d++) {
                    o << "    wmac(wacc_" << d << ", pool[oT + " << (2 * k) * DEG + d << "], " << v.first << ");\n";
                    o << "    wmac(wgb" << g << "_" << d << ", pool[oT + " << (2 * k + 1) * DEG + d << "], " << v.first << ");\n";
                }
            } else {
                const char* m = v.second ? "F::mul" : "F::mulb";
                o << "    acc = F::add(acc, " << m << "(" << pool_e("oT", 2 * k) << ", " << v.first << "));\n";
                o << "    gb" << g << " = F::add(gb" << g << ", " << m << "(" << pool_e("oT", 2 * k + 1) << ", " << v.first << "));\n";
            }
            if (wide && (q + 1 == nt || p.trans[order[q + 1]].group != g))
                o << "    gb" << g << " = F::add(gb" << g << ", " << wide_value("wgb" + std::to_string(g)) << ");\n";
        }
        if (wide) o << "    acc = F::add(acc, " << wide_value("wacc") << ");\n";
        for (size_t g = 0; g < ndg; g++) o << "    acc = F::add(acc, F::mulb(gb" << g << ", tw(a, (s * a.dg_exp[" << g << "]) & rmask)));\n";
        // assertions not met on the way (their column is in no transition constraint, or early = 0)
        for (uint32_t m = 0; m < in.members.size(); m++) {
            const BoundaryMember& bm = in.members[m];
            member(m, bm.aux ? aux_val(bm.col, false) : main_val(bm.col, false));
        }
        auto gpair = [&](const char* arr, size_t j) {
            std::ostringstream s2;
            s2 << "F::make(a." << arr << "[" << j * DEG << "], " << (DEG > 1 ? std::string("a.") + arr + "[" + std::to_string(j * DEG + 1) + "]" : std::string("0")) << ")";
            return s2.str();
        };
        if (wide)
            for (size_t j = 0; j < ng; j++) {
                o << "    sa" << j << " = F::add(sa" << j << ", " << wide_value("wsa" + std::to_string(j)) << ");\n";
                o << "    sb_" << j << " = F::add(sb_" << j << ", " << wide_value("wsb" + std::to_string(j)) << ");\n";
            }
        for (size_t j = 0, t = 0; j < ng; j++) {
            std::ostringstream e;
            e << "F::add(F::sub(sa" << j << ", " << gpair("gA", j) << "), F::mulb(F::sub(sb_" << j << ", " << gpair("gB", j) << "), tw(a, (s * a.bgroups[" << j << "].adj_exp) & rmask)))";
            if (in.bgroups[j].has_seq) {
                // sequence assertions of this group: minus the row's entry of the group's value table (air_host.hip: seq_tables)
                std::ostringstream v;
                v << "F::make(a.seq_tab[" << t * DEG << "ull * a.rows + s], " << (DEG > 1 ? "a.seq_tab[" + std::to_string(t * DEG + 1) + "ull * a.rows + s]" : std::string("0")) << ")";
                o << "    const T num" << j << " = F::sub(" << e.str() << ", " << v.str() << ");\n";
                t++;
            } else o << "    const T num" << j << " = " << e.str() << ";\n";
        }
        if (mode == 0) {
            for (int d = 0; d < DEG; d++) o << "    a.out_cols[" << d << "ull * a.count + (s - a.first)] = F::comp(acc, " << d << ");\n";
            for (size_t j = 0; j < ng; j++)
                for (int d = 0; d < DEG; d++) o << "    a.out_cols[" << ((1 + j) * DEG + d) << "ull * a.count + (s - a.first)] = F::comp(num" << j << ", " << d << ");\n";
            o << "    }\n}\n";
            return o.str();
        }
        // transition divisor (x^n - 1) / prod (x - w^(n-i)): its inverse comes from the table of (x^n - 1)^-1
        o << "    const uint64_t x = gl::mul(a.offset, tw(a, s & rmask));\n    uint64_t tdiv = a.zn_inv[s & a.xmask];\n";
        for (uint32_t i = 0; i < p.exemptions; i++) o << "    tdiv = gl::mul(tdiv, gl::sub(x, a.exempt[" << i << "]));\n";
        o << "    const T h = F::mulb(acc, tdiv);\n";
        if (!batch) {
            o << "    a.out_h0[s] = F::comp(h, 0);\n";
            if (DEG > 1) o << "    a.out_h1[s] = F::comp(h, 1);\n";
            o << "    }\n}\n";
            return o.str();
        }
        // boundary divisors x^a - b of this row as one fraction
        o << "    uint64_t run = 1;\n";
        for (size_t j = 0; j < ng; j++) {
            o << "    const uint64_t d" << j << " = gl::sub(gl::mul(a.bgroups[" << j << "].ha, tw(a, (s * a.bgroups[" << j << "].a_exp) & rmask)), a.bgroups[" << j << "].b);\n";
            if (j) o << "    const uint64_t pre" << j << " = run;\n";
            o << "    run = " << (j ? "gl::mul(run, d" + std::to_string(j) + ")" : std::string("d0")) << ";\n";
        }
        o << "    T nsum = F::zero();\n    uint64_t suf = 1;\n";
        for (size_t j = ng; j-- > 0;) {
            const std::string others = j == 0 ? "suf" : (j + 1 == ng ? "pre" + std::to_string(j) : "gl::mul(pre" + std::to_string(j) + ", suf)");
            o << "    nsum = F::add(nsum, F::mulb(num" << j << ", " << others << "));\n";
            if (j) o << "    suf = " << (j + 1 == ng ? "d" + std::to_string(j) : "gl::mul(suf, d" + std::to_string(j) + ")") << ";\n";
        }
        o << "    for (int d = 0; d < DEG; d++) { sh_h[q][d][threadIdx.x] = F::comp(h, d); sh_n[q][d][threadIdx.x] = F::comp(nsum, d); }\n";
        o << "    sh_d[q][threadIdx.x] = run;\n    }\n";
        // one inversion for the thread's R rows
        o << "    uint64_t pr[R], tot = 1;\n#pragma unroll\n    for (int q = 0; q < R; q++) { pr[q] = tot; tot = gl::mul(tot, sh_d[q][threadIdx.x]); }\n";
        o << "    uint64_t ia = gl::inv(tot);\n#pragma unroll\n    for (int q = R - 1; q >= 0; q--) {\n";
        o << "        const uint64_t di = gl::mul(ia, pr[q]);\n        ia = gl::mul(ia, sh_d[q][threadIdx.x]);\n";
        o << "        const T hq = F::add(F::make(sh_h[q][0][threadIdx.x], DEG > 1 ? sh_h[q][DEG - 1][threadIdx.x] : 0), F::mulb(F::make(sh_n[q][0][threadIdx.x], DEG > 1 ? sh_n[q][DEG - 1][threadIdx.x] : 0), di));\n";
        o << "        const uint64_t s = a.first + t0 + (uint64_t)q * 256;\n        a.out_h0[s] = F::comp(hq, 0);\n";
        if (DEG > 1) o << "        a.out_h1[s] = F::comp(hq, 1);\n";
        o << "    }\n}\n";
        return o.str();
    }
};

// BLAKE2s-256 of a byte string (host): integrity of the on-disk code-object cache
b2s::Digest hash_bytes(const void* data, size_t len) {
    b2s::State st; b2s::init(st);
    const unsigned char* p = (const unsigned char*)data;
    size_t off = 0;
    do {
        uint32_t m[16] = {0};
        const size_t take = std::min<size_t>(64, len - off);
        memcpy(m, p + off, take);
        off += take;
        b2s::compress(st, m, (uint32_t)off, (uint32_t)((uint64_t)off >> 32), off == len);
    } while (off < len);
    b2s::Digest d;
    for (int i = 0; i < 8; i++) d.w[i] = st.h[i];
    return d;
}
// Cache file = header + code object. The header binds the file to the generated source (a strong hash, not the file name's) and
// to its own payload, so a truncated, foreign or tampered file is recompiled instead of being handed to hipModuleLoadData.
struct CacheHeader {
    char magic[8];            // "AEROJIT1"
    uint64_t src_len, code_len;
    b2s::Digest src_hash, code_hash;
};
// A cache directory that anybody else can write to is a way to inject kernels into the prover (the header's digests are unkeyed: whoever can
// write the directory can forge a valid entry). Trusted = a real directory (not a symbolic link: lstat), owned by this process's user, not
// writable by group or others. Entries are opened with O_NOFOLLOW and must be regular files of the same owner (open_cache_entry).
bool cache_dir_trusted(const char* dir) {
    struct stat st;
    if (lstat(dir, &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
static FILE* open_cache_entry(const std::string& path) {
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH)) != 0) { close(fd); return nullptr; }
    FILE* f = fdopen(fd, "rb");
    if (!f) close(fd);
    return f;
}

std::string structure_key(const Program& p, const Instance& in, int DEG, int mode, int R) {
    std::ostringstream k;
    k << DEG << ":" << mode << ":" << R << ":" << in.bgroups.size() << ":";
    for (auto& m : in.members) k << m.group << (m.seq >= 0 ? "s" : "") << ",";
    // the validation kernels carry each boundary group's (stride, first step) as literals, and `first` follows the trace length for
    // steps counted from the end: a second aero_air_validate_trace with another length must not meet the first one's kernel
    if (mode >= 2) for (auto& g : in.bgroups) k << "|" << g.stride << "@" << g.first;
    return k.str();
}

std::shared_ptr<Blob> compile(const Program& p, const Instance& in, int DEG, int mode, int R) {
    auto blob = std::make_shared<Blob>();
    blob->id = g_next_id++;
    Gen gen(p, in, DEG, mode);
    const std::string src = gen.source(sizeof(AirJitArgs), R);
    if (const char* dump = getenv("AERO_AIR_JIT_DUMP")) {
        if (FILE* f = fopen(dump, "w")) { fwrite(src.data(), 1, src.size(), f); fclose(f); }
    }
    // optional on-disk cache of code objects (AERO_AIR_JIT_CACHE=<directory>): a prover process that restarts does not pay the
    // compilation again. Keyed by the generated source (which contains the field arithmetic and every switch) + the hiprtc version.
    std::string cache_path;
    const char* dir = getenv("AERO_AIR_JIT_CACHE");
    if (dir && *dir && !cache_dir_trusted(dir)) {
        std::lock_guard<std::mutex> lk(g_mu);
        g_last_error = std::string("AERO_AIR_JIT_CACHE ignored: ") + dir + " is not a directory of this user, is a symbolic link, or is group/world-writable";
        dir = nullptr;
    }
    if (dir && *dir) {
        int major = 0, minor = 0;
        (void)hiprtcVersion(&major, &minor);
        uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull ^ ((uint64_t)major << 32 | (uint32_t)minor);
        for (unsigned char ch : src) { h1 = (h1 ^ ch) * 0x100000001b3ull; h2 = (h2 + ch) * 0x9E3779B97F4A7C15ull; h2 ^= h2 >> 29; }
        char name[64];
        snprintf(name, sizeof name, "/aero_air_%016llx%016llx.co", (unsigned long long)h1, (unsigned long long)h2);
        cache_path = std::string(dir) + name;
        if (FILE* f = open_cache_entry(cache_path)) {
            fseek(f, 0, SEEK_END);
            const long sz = ftell(f);
            fseek(f, 0, SEEK_SET);
            CacheHeader hd;
            bool ok = sz > (long)sizeof hd && fread(&hd, 1, sizeof hd, f) == sizeof hd && memcmp(hd.magic, "AEROJIT1", 8) == 0 &&
                      hd.src_len == src.size() && hd.code_len == (uint64_t)sz - sizeof hd;
            if (ok) {
                blob->code.resize((size_t)hd.code_len);
                ok = fread(blob->code.data(), 1, blob->code.size(), f) == blob->code.size();
            }
            fclose(f);
            if (ok) {
                const b2s::Digest hs = hash_bytes(src.data(), src.size()), hc = hash_bytes(blob->code.data(), blob->code.size());
                ok = memcmp(&hs, &hd.src_hash, sizeof hs) == 0 && memcmp(&hc, &hd.code_hash, sizeof hc) == 0 &&
                     blob->code.size() > 4 && memcmp(blob->code.data(), "\x7f" "ELF", 4) == 0;
            }
            if (ok) { blob->cache_file = cache_path; return blob; }
            blob->code.clear();
            remove(cache_path.c_str());          // unusable entry: compiled again below and rewritten
        }
    }
    std::lock_guard<std::mutex> rtc_lock(g_rtc_mu);
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "air_jit_kernel.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        blob->error = "hiprtcCreateProgram failed";
        return blob;
    }
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pragma-once-outside-header"};
    const hiprtcResult rc = hiprtcCompileProgram(prog, 4, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, 0);
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        blob->error = std::string("hiprtc: ") + hiprtcGetErrorString(rc) + ": " + log.substr(0, 600);
    } else {
        size_t cs = 0;
        hiprtcGetCodeSize(prog, &cs);
        blob->code.resize(cs);
        if (cs) hiprtcGetCode(prog, blob->code.data());
        if (!cs) blob->error = "hiprtc produced no code object";
    }
    hiprtcDestroyProgram(&prog);
    if (!cache_path.empty() && blob->error.empty()) {         // written under a temporary name, then renamed: readers never see a partial file
        const std::string tmp = cache_path + ".tmp" + std::to_string((unsigned long long)blob->id) + "_" + std::to_string((long)getpid());
        const int tfd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
        FILE* f = tfd >= 0 ? fdopen(tfd, "wb") : nullptr;
        if (!f && tfd >= 0) close(tfd);
        if (f) {
            CacheHeader hd;
            memcpy(hd.magic, "AEROJIT1", 8);
            hd.src_len = src.size(); hd.code_len = blob->code.size();
            hd.src_hash = hash_bytes(src.data(), src.size()); hd.code_hash = hash_bytes(blob->code.data(), blob->code.size());
            const bool ok = fwrite(&hd, 1, sizeof hd, f) == sizeof hd && fwrite(blob->code.data(), 1, blob->code.size(), f) == blob->code.size();
            fclose(f);
            if (!ok || rename(tmp.c_str(), cache_path.c_str()) != 0) remove(tmp.c_str());
        }
    }
    return blob;
}

std::shared_ptr<Blob> get_blob(const Program& p, const Instance& in, int DEG, int mode, int R) {
    std::shared_ptr<Cache> cache;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!p.jit_cache) p.jit_cache = std::make_shared<Cache>();
        cache = std::static_pointer_cast<Cache>(p.jit_cache);
    }
    const std::string key = structure_key(p, in, DEG, mode, R);
    std::lock_guard<std::mutex> lk(cache->mu);          // one compilation per key; other threads of a pool wait for it
    auto it = cache->blobs.find(key);
    if (it != cache->blobs.end()) return it->second;
    auto blob = compile(p, in, DEG, mode, R);
    if (!blob->error.empty()) { std::lock_guard<std::mutex> lk2(g_mu); g_last_error = blob->error; }
    cache->blobs[key] = blob;
    return blob;
}

}  // namespace

std::string air_jit_last_error() {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_last_error;
}
// rows per thread of the fused form: 4 when the launch's row count allows it (whole domains and their power-of-two shards do)
static int jit_rows(const air::Program& p, size_t count, int mode, size_t n_bgroups) {
    if (mode != 1 || n_bgroups == 0) return 1;
    const int want = g_tune.rows ? g_tune.rows : (p.num_transition() + p.num_assertions() <= 16 ? 4 : 2);
    for (int r = want; r > 1; r >>= 1) if (count % ((size_t)256 * r) == 0) return r;
    return 1;
}
std::string air_jit_source(const air::Program& p, const air::Instance& in, int deg, int mode) {
    Gen gen(p, in, deg, mode);
    return gen.source(sizeof(AirJitArgs), jit_rows(p, (size_t)1 << 20, mode, in.bgroups.size()));
}
bool air_jit_compile_only(const air::Program& p, const air::Instance& in, int deg, int mode, std::string* err) {
    auto blob = get_blob(p, in, deg, mode, jit_rows(p, (size_t)1 << 20, mode, in.bgroups.size()));
    if (err) *err = blob->error;
    return blob->error.empty();
}

// The kernel a PROOF of 2^log_n rows will ask for (mode 1, rows per thread chosen from the rows one launch evaluates), built ahead of
// the proof: pools and sharded entry points call this once before their workers / ranks start, so no worker meets a compilation.
bool air_jit_prepare(const air::Program& p, int log_n, int deg, size_t rows_per_launch, std::string* err) {
    if (p.nodes.size() + p.num_transition() + p.num_assertions() > 20000) { if (err) *err = "program too large for the compiled evaluator (interpreter)"; return false; }
    const air::Instance in = air::instantiate(p, log_n);
    auto blob = get_blob(p, in, deg, 1, jit_rows(p, rows_per_launch, 1, in.bgroups.size()));
    if (err) *err = blob->error;
    return blob->error.empty();
}

void Context::unload_jit_modules(Context* ctx) {
    if (ctx->jit_modules.empty()) return;
    std::lock_guard<std::mutex> rtc_lock(g_rtc_mu);
    for (hipModule_t m : ctx->jit_modules) (void)hipModuleUnload(m);
    ctx->jit_modules.clear();
    ctx->jit_funcs.clear();
}

template <class F> bool launch_air_jit(Context* ctx, const air::Program& p, const air::Instance& in, const AirConsArgs<F>& c, const uint64_t* pdesc, uint64_t oSE,
                                       uint64_t oT, uint64_t oB, int mode) {
    if (!ctx->air_jit) return false;
    // a kernel body grows with the program: beyond this many nodes + constraints + assertions (minutes of compilation, a code object of
    // several MB) the interpreter, whose cost per proof does not depend on a compiler, is the better evaluator
    if (p.nodes.size() + p.num_transition() + p.num_assertions() > 20000) return false;
    const int R = mode >= 2 ? 1 : jit_rows(p, c.count, mode, in.bgroups.size());
    auto blob = get_blob(p, in, F::DEG, mode, R);
    if (!blob->error.empty()) return false;
    hipFunction_t fn;
    auto it = ctx->jit_funcs.find(blob->id);
    if (it != ctx->jit_funcs.end()) {
        fn = (hipFunction_t)it->second;
    } else {
        // Lock order everywhere: a program's cache mutex (get_blob) BEFORE g_rtc_mu (compile, module load) - so the retry after a
        // cache entry that does not load re-enters get_blob with g_rtc_mu released.
        hipModule_t mod = nullptr;
        auto load = [&](const std::shared_ptr<Blob>& b) {
            std::lock_guard<std::mutex> rtc_lock(g_rtc_mu);
            const bool ok = hipModuleLoadData(&mod, b->code.data()) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            return ok;
        };
        if (!load(blob)) {
            // not retried on every launch; a cache entry that does not load (built for another chip or runtime) is dropped and the
            // program compiled afresh, once (the entry is gone, so this cannot loop)
            // A load failure is a fact about THIS context at this moment (out of memory, a device in a bad state), not about the program: the
            // shared blob is not marked failed - other contexts of a pool read blob->error without a lock, and a transient failure here must
            // not switch every later proof of the process to the interpreter. Only a compile error (set once, before the blob is published) is
            // permanent. This context gets the interpreter for this launch and tries again at its next one.
            const bool from_cache = !blob->cache_file.empty();
            if (from_cache) remove(blob->cache_file.c_str());
            { std::lock_guard<std::mutex> lk(g_mu); g_last_error = "hipModuleLoadData refused the code object"; }
            if (!from_cache) return false;
            {
                auto cache = std::static_pointer_cast<Cache>(p.jit_cache);
                std::lock_guard<std::mutex> lk(cache->mu);
                cache->blobs.erase(structure_key(p, in, F::DEG, mode, R));
            }
            blob = get_blob(p, in, F::DEG, mode, R);
            if (!blob->error.empty() || !load(blob)) return false;
        }
        std::lock_guard<std::mutex> rtc_lock(g_rtc_mu);
        ctx->jit_modules.push_back(mod);
        ctx->jit_blobs.push_back(blob);          // the image stays alive as long as the module that was loaded from it
        if (hipModuleGetFunction(&fn, mod, "air_jit_kernel") != hipSuccess) { (void)hipGetLastError(); return false; }
        ctx->jit_funcs[blob->id] = (void*)fn;
    }
    AirJitArgs a{};
    a.lde = c.lde; a.aux = c.aux; a.N = c.N; a.blowup = c.blowup; a.ce_step = c.ce_step; a.split_log = c.split_log;
    a.rows = c.rows; a.first = c.first; a.count = c.count;
    a.pool = c.pool; a.oSE = oSE; a.oT = oT; a.oB = oB;
    a.ptab = c.ptab; a.pdesc = pdesc; a.dg_exp = c.dg_exp; a.bgroups = c.bgroups;
    a.gA = reinterpret_cast<const uint64_t*>(c.gA); a.gB = reinterpret_cast<const uint64_t*>(c.gB);
    a.tw_lo = c.tw_lo; a.tw_hi = c.tw_hi; a.tw_h = (uint64_t)c.tw_h; a.offset = c.offset;
    a.zn_inv = c.zn_inv; a.xmask = c.xmask; a.exempt = c.exempt;
    a.out_cols = c.out_cols; a.out_h0 = c.out_h[0]; a.out_h1 = c.out_h[1];
    a.seq_tab = c.seq_tab;
    size_t size = sizeof(a);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    const size_t in_cols = (size_t)c.W + (size_t)c.A * F::DEG;
    const size_t abytes = c.count * 8 * (in_cols + (mode == 0 ? (1 + c.n_bgroups) * F::DEG : F::DEG));
    const bool timed = ctx->kernel_timing && ctx->kt_match("air_jit_kernel");
    if (timed) ctx->kt_begin("air_jit_kernel", abytes);
    AERO_HIP(hipModuleLaunchKernel(fn, (unsigned)((c.count + 256 * R - 1) / (256 * R)), 1, 1, 256, 1, 1, 0, ctx->stream, nullptr, config));
    if (timed) ctx->kt_end();
    ctx->check_launch("air_jit");
    return true;
}
template bool launch_air_jit<gl::FB>(Context*, const air::Program&, const air::Instance&, const AirConsArgs<gl::FB>&, const uint64_t*, uint64_t, uint64_t, uint64_t, int);
template bool launch_air_jit<gl::FQ>(Context*, const air::Program&, const air::Instance&, const AirConsArgs<gl::FQ>&, const uint64_t*, uint64_t, uint64_t, uint64_t, int);

}  // namespace aero
