// Per-proof device tables + launches of a program AIR (see air_host.hpp).
#include "air_host.hpp"
#include <atomic>
#include <map>
#include <mutex>
#include <thread>

namespace aero {

using gl::FB;
using gl::FQ;
using namespace air;

static int ilog2z(uint64_t x) { int r = 0; while ((1ull << r) < x) r++; return r; }

namespace {
struct PeriodicTables {
    std::vector<uint64_t> tab;
    std::vector<uint32_t> off, mask;
};
PeriodicTables periodic_tables(const Program& p, uint64_t n, uint64_t rows, uint64_t h) {
    PeriodicTables t;
    for (auto& cyc : p.periodic) {
        const std::vector<uint64_t> v = periodic_table(cyc, n, rows, h);
        if (t.tab.size() + v.size() > (1ull << 28)) fail("air program: periodic tables too large for this domain", ST_UNSUPPORTED);
        t.off.push_back((uint32_t)t.tab.size());
        t.mask.push_back((uint32_t)v.size() - 1);
        t.tab.insert(t.tab.end(), v.begin(), v.end());
    }
    if (t.tab.empty()) { t.tab.push_back(0); t.off.push_back(0); t.mask.push_back(0); }
    return t;
}
// The per-proof pool (u64 words): scalar operands | (alpha, beta') per transition constraint | (alpha, beta') per assertion |
// LOAD descriptors | one spare block; and the code with every instruction's position in it written into `poff`.
template <class F> struct Pool {
    std::vector<uint64_t> words;
    std::vector<Insn> code;
    uint32_t oSE = 0, oT = 0, oB = 0;         // where the E scalars, the transition and the boundary coefficient pairs start
};
template <class F>
Pool<F> build_pool(const Program& p, const std::vector<Insn>& code, const std::vector<uint32_t>& desc, const Scalars<F>& sc, const PeriodicTables& pt,
                   const std::vector<typename F::T>& ta, const std::vector<typename F::T>& tb, const std::vector<typename F::T>& ba,
                   const std::vector<typename F::T>& bb, const std::vector<uint32_t>& mem_group) {
    Pool<F> pool;
    auto& w = pool.words;
    auto put = [&](typename F::T v) { for (int d = 0; d < F::DEG; d++) w.push_back(F::comp(v, d)); };
    const uint32_t oSB = 0;
    w.insert(w.end(), sc.b.begin(), sc.b.end());
    const uint32_t oSE = (uint32_t)w.size();
    for (auto& v : sc.e) put(v);
    const uint32_t oT = (uint32_t)w.size();
    for (size_t k = 0; k < ta.size(); k++) { put(ta[k]); put(tb[k]); }
    const uint32_t oB = (uint32_t)w.size();
    for (size_t m = 0; m < ba.size(); m++) { put(ba[m]); put(bb[m]); }
    const uint32_t oD = (uint32_t)w.size();
    for (uint32_t d : desc) {
        const bool next = d >> 31, periodic = (d >> 30) & 1;
        const uint32_t slot = (d >> 16) & 0xfff, col = d & 0xffff;
        if (periodic) { int lg = 0; while ((1u << lg) <= pt.mask[col]) lg++; w.push_back(air_dev_desc(false, true, slot, pt.off[col], (uint32_t)lg)); }
        else w.push_back(air_dev_desc(next, false, slot, col, 0));
    }
    for (int i = 0; i < 8; i++) w.push_back(0);
    if (w.size() >= (1ull << 31)) fail("air program: scalar pool too large", ST_UNSUPPORTED);
    pool.oSE = oSE; pool.oT = oT; pool.oB = oB;
    pool.code = code;
    for (Insn& I : pool.code) {
        switch (I.pk) {
            case PK_SCAL_B: I.poff = oSB + I.pi; break;
            case PK_SCAL_E: I.poff = oSE + I.pi * F::DEG; break;
            case PK_TCOEF: I.poff = oT + I.pi * 2 * F::DEG; if ((I.op & 0xff) == OP_EMIT3_B) I.dst = I.pad; break;      // device: dst = third register
            case PK_BCOEF: I.poff = oB + I.pi * 2 * F::DEG; I.dst = mem_group[I.pi]; break;
            case PK_DESC: I.poff = oD + I.pi; break;
            default: I.poff = 0;
        }
    }
    for (int i = 0; i < 3; i++) pool.code.push_back(Insn{OP_END, 0, 0, 0, 0, PK_NONE, 0, 0});      // the interpreter fetches two instructions ahead
    return pool;
}
struct DivisorTables {
    std::vector<AirBGroupDev> groups;
    std::vector<uint64_t> zn_inv, exempt;
};
// divisors on the domain offset * <w_rows>: x^n takes rows / n values (indexed by s mod that)
DivisorTables divisor_tables(const Program& p, const Instance& in, uint64_t rows, uint64_t h) {
    DivisorTables d;
    const uint64_t n = in.n, xcount = rows / n;
    const uint64_t hn = gl::pow(h, n), wx = gl::root_of_unity(ilog2z(xcount));
    for (uint64_t k = 0; k < xcount; k++) d.zn_inv.push_back(gl::inv(gl::sub(gl::mul(hn, gl::pow(wx, k)), 1)));
    const uint64_t g = gl::root_of_unity(in.log_n);
    for (uint32_t i = 1; i <= p.exemptions; i++) d.exempt.push_back(gl::pow(g, n - i));
    for (auto& bg : in.bgroups)
        d.groups.push_back(AirBGroupDev{bg.a & (rows - 1), gl::pow(h, bg.a), bg.b, bg.adj & (rows - 1)});
    if (d.groups.empty()) d.groups.push_back(AirBGroupDev{});
    return d;
}
// Sequence assertions (`Assertion::sequence`): the value an asserted column must take varies with the row - it is the polynomial
// P_m(x w_n^-first) (P_m = interpolant of the assertion's values; winter-air 0.4 BoundaryConstraint, "poly_offset"). For every divisor
// group that holds such assertions ONE E-valued table over the evaluation domain is built per proof,
//     T_g(x) = sum_m (alpha_m + beta_m x^adj_g) P_m(x w_n^-first_g),
// which the evaluation kernels subtract from the group's numerator: deg T_g < ce_n, so it is the transform of a coefficient vector
// with at most 2 (n / stride) non-zero entries per component (folded modulo x^rows - offset^rows when a rank evaluates a domain
// smaller than ce_n). groups[j].seq = 1 + index of the group's table (0 = none); returns DEG * (#tables) columns of `rows` entries.
template <class F>
const uint64_t* seq_tables(Context* ctx, const Program& p, const Instance& in, uint64_t rows, uint64_t h, const std::vector<typename F::T>& ba,
                           const std::vector<typename F::T>& bb, std::vector<AirBGroupDev>& groups) {
    typedef typename F::T T;
    uint32_t ntab = 0;
    for (size_t j = 0; j < in.bgroups.size(); j++) if (in.bgroups[j].has_seq) groups[j].seq = ++ntab;
    if (!ntab) return nullptr;
    const int lg = ilog2z(rows);
    std::vector<uint64_t> pos, val;                    // scatter list: destination (bit-reversed index within its column) and value
    for (size_t j = 0; j < in.bgroups.size(); j++) {
        if (!in.bgroups[j].has_seq) continue;
        const BoundaryGroup& bg = in.bgroups[j];
        // (coefficient index mod rows, c_k offset^k) pairs of every member, then sorted and merged: at most 2 n / stride entries per member
        std::vector<std::pair<uint64_t, T>> terms;
        const uint64_t hadj = gl::pow(h, bg.adj);
        for (size_t m = 0; m < in.members.size(); m++) {
            const BoundaryMember& bm = in.members[m];
            if (bm.seq < 0 || bm.group != j) continue;
            const std::shared_ptr<const std::vector<uint64_t>> co_keep = sequence_poly_cached(p, (uint32_t)bm.seq, in.log_n, bg.first);
            const std::vector<uint64_t>& co = *co_keep;
            terms.reserve(terms.size() + 2 * co.size());
            uint64_t hk = 1;                                           // offset^k
            for (size_t k = 0; k < co.size(); k++) {
                if (co[k]) {
                    const uint64_t c = gl::mul(co[k], hk);
                    terms.push_back({k & (rows - 1), F::mulb(ba[bm.coef], c)});
                    terms.push_back({(k + bg.adj) & (rows - 1), F::mulb(bb[bm.coef], gl::mul(c, hadj))});
                }
                hk = gl::mul(hk, h);
            }
        }
        std::sort(terms.begin(), terms.end(), [](const std::pair<uint64_t, T>& x, const std::pair<uint64_t, T>& y) { return x.first < y.first; });
        const uint64_t col0 = (uint64_t)(groups[j].seq - 1) * F::DEG;
        for (size_t i = 0; i < terms.size();) {
            T acc = terms[i].second;
            size_t e = i + 1;
            while (e < terms.size() && terms[e].first == terms[i].first) acc = F::add(acc, terms[e++].second);
            for (int d = 0; d < F::DEG; d++) {
                pos.push_back((col0 + d) * rows + gl::bitrev((uint32_t)terms[i].first, lg));
                val.push_back(F::comp(acc, d));
            }
            i = e;
        }
    }
    const size_t ncols = (size_t)ntab * F::DEG;
    uint64_t* tab = (uint64_t*)ctx->scratch_alloc(ncols * rows * 8);
    AERO_HIP(hipMemsetAsync(tab, 0, ncols * rows * 8, ctx->stream));
    if (!pos.empty()) {
        // the lists travel through a staging copy of their own (they can exceed the parameter staging block)
        uint64_t* d_pos = (uint64_t*)ctx->scratch_alloc(pos.size() * 16);
        uint64_t* d_val = d_pos + pos.size();
        AERO_HIP(hipMemcpyAsync(d_pos, pos.data(), pos.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        AERO_HIP(hipMemcpyAsync(d_val, val.data(), val.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        launch_air_scatter(ctx, tab, d_pos, d_val, pos.size());
        ctx->sync();                                   // pos / val are pageable host vectors that die with this frame
    }
    ctx->ntt_forward(tab, rows, tab, rows, (int)ncols, lg, 0);       // plain transform: bit-reversed coefficients in, natural order out
    return tab;
}
}  // namespace

template <class F>
void air_eval_constraints(Context* ctx, const Program& p, const Instance& in, const AirGeometry& g, const AirCoeffs<F>& cc, const uint64_t* pub,
                          const typename F::T* rands, int mode, uint64_t* out_cols, uint64_t* const out_h[2]) {
    typedef typename F::T T;
    if (cc.ta.size() != p.num_transition() || cc.tb.size() != p.num_transition() || cc.ba.size() != p.num_assertions() || cc.bb.size() != p.num_assertions())
        fail("air program: wrong number of composition coefficients", ST_INTERNAL);
    if (g.rows % in.n || (g.rows & (g.rows - 1)) || g.frame_rows % g.rows) fail("air program: evaluation domain does not fit the frame matrix", ST_INTERNAL);
    const uint64_t rows = g.rows, h = g.offset;
    const Scalars<F> sc = fold_scalars<F>(p, pub, rands);
    // offset^adj goes into the beta coefficients; the device multiplies by w^(s adj) from the twiddle table
    std::vector<T> tb(cc.tb);
    std::vector<uint64_t> dg_exp;
    {
        std::vector<uint64_t> hadj;
        for (uint64_t adj : in.dgroup_adj) { hadj.push_back(gl::pow(h, adj)); dg_exp.push_back(adj & (rows - 1)); }
        for (size_t k = 0; k < tb.size(); k++) tb[k] = F::mulb(tb[k], hadj[p.trans[k].group]);
    }
    // assertions by member id; their constant parts sum alpha value, sum beta' value per divisor group
    const size_t nm = in.members.size(), ng = in.bgroups.size();
    std::vector<T> ba(nm ? nm : 1, F::zero()), bb(nm ? nm : 1, F::zero()), gA(ng ? ng : 1, F::zero()), gB(ng ? ng : 1, F::zero());
    std::vector<uint32_t> mem_group(nm ? nm : 1, 0);
    {
        std::vector<uint64_t> hadj;
        for (auto& bg : in.bgroups) hadj.push_back(gl::pow(h, bg.adj));
        for (size_t m = 0; m < nm; m++) {
            const BoundaryMember& bm = in.members[m];
            ba[m] = cc.ba[bm.coef];
            bb[m] = F::mulb(cc.bb[bm.coef], hadj[bm.group]);
            mem_group[m] = bm.group;
            if (bm.seq >= 0) continue;                   // a sequence assertion's value varies with the row: seq_tables() below
            const T val = bm.val_ext ? sc.e[bm.val_idx] : F::from(sc.b[bm.val_idx]);
            gA[bm.group] = F::add(gA[bm.group], F::mul(ba[m], val));
            gB[bm.group] = F::add(gB[bm.group], F::mul(bb[m], val));
        }
    }
    const PeriodicTables pt = periodic_tables(p, in.n, rows, h);
    DivisorTables dv = divisor_tables(p, in, rows, h);
    const uint64_t* seq_tab = seq_tables<F>(ctx, p, in, rows, h, cc.ba, cc.bb, dv.groups);
    const Pool<F> pool = build_pool<F>(p, p.cons_code, p.cons_desc, sc, pt, cc.ta, tb, nm ? ba : std::vector<T>(), nm ? bb : std::vector<T>(), mem_group);
    std::vector<uint64_t> pdesc;               // the run-time compiled kernel's view of the periodic tables: offset | mask << 32
    for (size_t k = 0; k < pt.off.size(); k++) pdesc.push_back((uint64_t)pt.off[k] | ((uint64_t)pt.mask[k] << 32));
    ParamPack pp(ctx);
    const size_t i_code = pp.add(pool.code), i_pool = pp.add(pool.words), i_pt = pp.add(pt.tab), i_dg = pp.add(dg_exp), i_bg = pp.add(dv.groups),
                 i_ga = pp.add(gA), i_gb = pp.add(gB), i_zn = pp.add(dv.zn_inv), i_ex = pp.add(dv.exempt), i_pd = pp.add(pdesc);
    pp.commit();
    NttTables* tw = ctx->ntt_tables(ilog2z(rows));
    AirConsArgs<F> a{};
    a.lde = g.lde; a.aux = g.aux; a.N = g.frame_rows; a.W = p.W; a.A = p.A;
    a.blowup = (uint32_t)(g.frame_rows / in.n); a.ce_step = (uint32_t)(g.frame_rows / rows); a.split_log = g.split_log;
    a.rows = rows; a.first = g.first; a.count = g.count;
    a.code = pp.ptr<Insn>(i_code); a.pool = pp.ptr<uint64_t>(i_pool); a.slotsB = p.cons_slotsB; a.slotsE = p.cons_slotsE;
    a.ptab = pp.ptr<uint64_t>(i_pt); a.dg_exp = pp.ptr<uint64_t>(i_dg);
    a.bgroups = pp.ptr<AirBGroupDev>(i_bg); a.n_bgroups = (uint32_t)in.bgroups.size();
    a.gA = pp.ptr<T>(i_ga); a.gB = pp.ptr<T>(i_gb); a.seq_tab = seq_tab;
    a.tw_lo = tw->lo_fwd; a.tw_hi = tw->hi_fwd; a.tw_h = tw->h; a.offset = h;
    a.zn_inv = pp.ptr<uint64_t>(i_zn); a.xmask = (uint32_t)(rows / in.n) - 1;
    a.exempt = pp.ptr<uint64_t>(i_ex); a.n_exempt = p.exemptions;
    a.out_cols = out_cols;
    if (out_h) { a.out_h[0] = out_h[0]; a.out_h[1] = out_h[1]; }
    // the program as a kernel of its own, compiled when it is first met (air_jit.hip); the interpreter otherwise
    if (launch_air_jit<F>(ctx, p, in, a, pp.ptr<uint64_t>(i_pd), pool.oSE, pool.oT, pool.oB, mode)) return;
    if (mode == 0) { launch_air_constraints<F>(ctx, a, 0); return; }
    if (launch_air_constraints<F>(ctx, a, 1)) return;
    // too many boundary divisors for the fused form: numerator columns, then the division
    if (g.first != 0 || g.count != rows) fail("air program: unfused division needs the whole domain", ST_INTERNAL);
    const size_t ncols = in.num_columns() * F::DEG;
    uint64_t* cols = (uint64_t*)ctx->scratch_alloc(ncols * rows * 8);
    a.out_cols = cols;
    launch_air_constraints<F>(ctx, a, 0);
    AirDivideArgs<F> d{};
    d.cols = cols; d.rows = rows; d.bgroups = a.bgroups; d.n_bgroups = a.n_bgroups; d.tw_lo = a.tw_lo; d.tw_hi = a.tw_hi; d.tw_h = a.tw_h;
    d.offset = h; d.zn_inv = a.zn_inv; d.xmask = a.xmask; d.exempt = a.exempt; d.n_exempt = a.n_exempt;
    d.out_h[0] = out_h[0]; d.out_h[1] = out_h[1];
    launch_air_divide<F>(ctx, d);
}
template void air_eval_constraints<FB>(Context*, const Program&, const Instance&, const AirGeometry&, const AirCoeffs<FB>&, const uint64_t*, const uint64_t*, int, uint64_t*, uint64_t* const[2]);
template void air_eval_constraints<FQ>(Context*, const Program&, const Instance&, const AirGeometry&, const AirCoeffs<FQ>&, const uint64_t*, const gl::E2*, int, uint64_t*, uint64_t* const[2]);

template <class F>
void air_divide_columns(Context* ctx, const Program& p, const Instance& in, const uint64_t* cols_dev, size_t rows, uint64_t offset, uint64_t* const out_h[2]) {
    const DivisorTables dv = divisor_tables(p, in, rows, offset);
    ParamPack pp(ctx);
    const size_t i_bg = pp.add(dv.groups), i_zn = pp.add(dv.zn_inv), i_ex = pp.add(dv.exempt);
    pp.commit();
    NttTables* tw = ctx->ntt_tables(ilog2z(rows));
    AirDivideArgs<F> d{};
    d.cols = cols_dev; d.rows = rows; d.bgroups = pp.ptr<AirBGroupDev>(i_bg); d.n_bgroups = (uint32_t)in.bgroups.size();
    d.tw_lo = tw->lo_fwd; d.tw_hi = tw->hi_fwd; d.tw_h = tw->h; d.offset = offset;
    d.zn_inv = pp.ptr<uint64_t>(i_zn); d.xmask = (uint32_t)(rows / in.n) - 1; d.exempt = pp.ptr<uint64_t>(i_ex); d.n_exempt = p.exemptions;
    d.out_h[0] = out_h[0]; d.out_h[1] = out_h[1];
    launch_air_divide<F>(ctx, d);
}
template void air_divide_columns<FB>(Context*, const Program&, const Instance&, const uint64_t*, size_t, uint64_t, uint64_t* const[2]);
template void air_divide_columns<FQ>(Context*, const Program&, const Instance&, const uint64_t*, size_t, uint64_t, uint64_t* const[2]);

template <class F>
void air_build_aux(Context* ctx, const Program& p, const uint64_t* trace_dev, int log_n, const uint64_t* pub, const typename F::T* rands, uint64_t* out) {
    typedef typename F::T T;
    if (!p.has_builders()) fail("air program: the program does not describe how its auxiliary columns are built (no aux builders)", ST_UNSUPPORTED);
    const uint64_t n = 1ull << log_n;
    for (auto& v : p.periodic) if (v.size() > n) fail("air program: a periodic column's cycle is longer than the trace");
    const Scalars<F> sc = fold_scalars<F>(p, pub, rands);
    const PeriodicTables pt = periodic_tables(p, n, n, 1);      // on the trace domain the tables are the cycles themselves
    std::vector<T> init(p.A);
    for (uint32_t c = 0; c < p.A; c++) {
        const DOperand o = device_operand(p, p.builders[c].init);
        init[c] = o.kind == D_SCAL_E ? sc.e[o.idx] : F::from(sc.b[o.idx]);
    }
    const Pool<F> pool = build_pool<F>(p, p.aux_code, p.aux_desc, sc, pt, {}, {}, {}, {}, {});
    ParamPack pp(ctx);
    const size_t i_code = pp.add(pool.code), i_pool = pp.add(pool.words), i_pt = pp.add(pt.tab), i_hd = pp.add(p.has_den), i_in = pp.add(init), i_ha = pp.add(p.has_add);
    pp.commit();
    AirAuxArgs<F> a{};
    a.trace = trace_dev; a.n = n; a.W = p.W; a.A = p.A;
    a.code = pp.ptr<Insn>(i_code); a.pool = pp.ptr<uint64_t>(i_pool); a.slotsB = p.aux_slotsB; a.slotsE = p.aux_slotsE;
    a.ptab = pp.ptr<uint64_t>(i_pt);
    a.has_den = pp.ptr<uint8_t>(i_hd); a.has_add = pp.ptr<uint8_t>(i_ha); a.init = pp.ptr<T>(i_in); a.out = out;
    bool any_general_col = false;
    for (uint32_t c = 0; c < p.A; c++) any_general_col |= p.has_add[c] == 4;
    hipEvent_t ev_trace_ready = nullptr;
    if (any_general_col) {      // the main columns a general recurrence reads can leave for the host while the affine columns are scanned
        ev_trace_ready = ctx->sync_event(48);
        AERO_HIP(hipEventRecord(ev_trace_ready, ctx->stream));
    }
    launch_air_aux<F>(ctx, a, p.has_den, p.has_add);
    // General recurrences (den = REF_GENERAL): column(i + 1) = expr(main row i, main row i + 1, aux row i of the columns up to its own).
    // Nothing about such a recurrence can be scanned, so it is evaluated row after row on the HOST, after the columns the device built:
    // the main columns it reads and the auxiliary columns so far come down, the finished columns go back up. For exotic
    // AIRs (about 0.1 us per row and node), not a fast path: running products / sums / mixed affine forms never take it.
    bool any_general = false;
    for (uint32_t c = 0; c < p.A; c++) any_general |= p.has_add[c] == 4;
    if (!any_general) return;
    // AERO_AIR_GENERAL_DEVICE=1 (round 5): one wavefront per general column on the proving stream, no copy of the columns to the host and back, no
    // stream synchronisation (air_kernels.hpp: AirGeneralArgs). Measured on MI355X against the host evaluation below (v2 example, 4 nodes per row,
    // profiles/r5_general_recurrence.md): 2^14 rows 20.2 ms against 0.37 ms, 2^18 rows 322 ms against 3.8 ms, 2^20 rows 1286 ms against 22 ms -
    // a dependent chain of field multiplications costs a lone wavefront hundreds of cycles per link (about 1.2 us per row) and a host core a few
    // nanoseconds; the two copies and the synchronisation the host step needs are noise next to that. A serial chain is the host's job: the host
    // evaluation stays the default, the kernel stays as the tested alternative for a host that must not be interrupted.
    static const bool general_host = !(getenv("AERO_AIR_GENERAL_DEVICE") && getenv("AERO_AIR_GENERAL_DEVICE")[0] == '1');
    // Both forms run the same compiled recurrence: per general column the leaf operands it reads (loads -> slots), the nodes that do not
    // depend on the column's own value (par: slot = op(slot | const, slot | const)) and the dependent chain (ser), in topological order.
    struct Compiled { std::vector<GenLoad> loads; std::vector<GenInsn> par, ser; std::vector<T> consts; uint32_t res_kind = 0, res_idx = 0, n_slots = 0, n_serial = 0; };
    std::vector<Compiled> progs(p.A);
    for (uint32_t c = 0; c < p.A; c++) {
        if (p.has_add[c] != 4) continue;
        Compiled& g = progs[c];
        std::map<uint64_t, uint32_t> load_slot, const_idx;       // (kind, index) -> slot / pool index
        std::vector<int> node_slot(p.nodes.size(), -1), node_ser(p.nodes.size(), -1);
        uint32_t n_slots = 0, n_serial = 0;
        struct Op { uint32_t kind, idx; bool serial; };
        auto scalar = [&](T v) {
            g.consts.push_back(v);
            return (uint32_t)g.consts.size() - 1;
        };
        auto resolve = [&](uint32_t ref) -> Op {
            const uint32_t k = ref_kind(ref), j = ref_index(ref);
            auto load = [&](uint32_t kind, uint32_t col, uint32_t mask) {
                const uint64_t key = ((uint64_t)kind << 56) | ((uint64_t)mask << 28) | col;
                auto it = load_slot.find(key);
                if (it == load_slot.end()) { it = load_slot.emplace(key, n_slots++).first; g.loads.push_back(GenLoad{kind, col, it->second, mask}); }
                return Op{GOP_SLOT, it->second, false};
            };
            auto constant = [&](uint32_t space, uint32_t idx, T v) {
                const uint64_t key = ((uint64_t)space << 32) | idx;
                auto it = const_idx.find(key);
                if (it == const_idx.end()) it = const_idx.emplace(key, scalar(v)).first;
                return Op{GOP_CONST, it->second, false};
            };
            switch (k) {
                case K_NODE:
                    if (p.scalar_of[j] >= 0) return constant(0, j, p.is_ext[j] ? sc.e[p.scalar_of[j]] : F::from(sc.b[p.scalar_of[j]]));
                    if (node_ser[j] >= 0) return Op{GOP_SER, (uint32_t)node_ser[j], true};
                    return Op{GOP_SLOT, (uint32_t)node_slot[j], false};
                case K_MAIN_CUR: return load(GLD_MAIN_CUR, j, 0);
                case K_MAIN_NXT: return load(GLD_MAIN_NXT, j, 0);
                case K_AUX_CUR: return j == c ? Op{GOP_X, 0, true} : load(GLD_AUX_CUR, j, 0);
                case K_PERIODIC: return load(GLD_PERIODIC, pt.off[j], pt.mask[j]);
                case K_CONST: return constant(1, j, F::from(p.consts[j]));
                case K_PUB: return constant(2, j, F::from(sc.b[p.consts.size() + j]));
                case K_RAND: return constant(3, j, sc.e[j]);
                default: fail("air program: operand not available to a general aux recurrence", ST_INTERNAL); return Op{0, 0, false};
            }
        };
        for (uint32_t j : p.general_nodes[c]) {          // ascending = topological
            const Node& nd = p.nodes[j];
            const Op x = resolve(nd.a), y = resolve(nd.b);
            if (x.serial || y.serial) { node_ser[j] = (int)n_serial++; g.ser.push_back(GenInsn{nd.op, (uint32_t)node_ser[j], x.kind, x.idx, y.kind, y.idx}); }
            else { node_slot[j] = (int)n_slots++; g.par.push_back(GenInsn{nd.op, (uint32_t)node_slot[j], x.kind, x.idx, y.kind, y.idx}); }
        }
        const Op r = resolve(p.builders[c].num);
        g.res_kind = r.kind; g.res_idx = r.idx;
        g.n_slots = n_slots; g.n_serial = n_serial;
    }
    if (!general_host) {
        bool fits = true;
        for (uint32_t c = 0; c < p.A; c++) {
            if (p.has_add[c] != 4) continue;
            Compiled& g = progs[c];
            if (g.n_slots > GEN_MAX_SLOTS - 1 || g.n_serial > GEN_MAX_SERIAL) fits = false;      // the last slot is the dummies' (below)
            // ParamPack keeps POINTERS to its sources until commit(): no section may be empty, none may be a temporary
            if (g.consts.empty()) g.consts.push_back(F::zero());
            if (g.loads.empty()) g.loads.push_back(GenLoad{GLD_MAIN_CUR, 0, GEN_MAX_SLOTS - 1, 0});
            if (g.par.empty()) g.par.push_back(GenInsn{1, GEN_MAX_SLOTS - 1, GOP_CONST, 0, GOP_CONST, 0});
        }
        const std::vector<uint64_t> ptab_or_dummy = pt.tab.empty() ? std::vector<uint64_t>(1, 0) : pt.tab;
        if (fits) {
            for (uint32_t c = 0; c < p.A; c++) {
                if (p.has_add[c] != 4) continue;
                const Compiled& g = progs[c];
                ParamPack gp(ctx);
                const size_t i_ld = gp.add(g.loads), i_par = gp.add(g.par), i_ser = gp.add(g.ser), i_c = gp.add(g.consts), i_pt2 = gp.add(ptab_or_dummy);
                gp.commit();
                AirGeneralArgs<F> ga{};
                ga.trace = trace_dev; ga.aux = out; ga.n = n; ga.col = c;
                ga.loads = gp.ptr<GenLoad>(i_ld); ga.n_loads = (uint32_t)g.loads.size();
                ga.par = gp.ptr<GenInsn>(i_par); ga.n_par = (uint32_t)g.par.size();
                ga.ser = gp.ptr<GenInsn>(i_ser); ga.n_ser = (uint32_t)g.ser.size();
                ga.res_kind = g.res_kind; ga.res_idx = g.res_idx;
                ga.consts = gp.ptr<T>(i_c); ga.ptab = gp.ptr<uint64_t>(i_pt2); ga.init = init[c];
                launch_air_general_column<F>(ctx, ga);
            }
            return;
        }
    }
    // ---- the host step (round 6: two phases per chunk of rows, one thread per general column, copies on their own stream) ----
    // Sums are exact in the field, so a dependent chain (((x x) + a) + b) is re-associated to (x x) + (a + b): everything that does not
    // depend on the column's own value leaves the chain (v2 example: three dependent operations per row -> two).
    for (uint32_t c = 0; c < p.A; c++) {
        if (p.has_add[c] != 4) continue;
        Compiled& g = progs[c];
        for (bool changed = true; changed;) {
            changed = false;
            for (size_t k = 0; k < g.ser.size() && !changed; k++) {
                GenInsn& s2 = g.ser[k];
                if ((s2.op != 1 && s2.op != 2) || s2.ka != GOP_SER || s2.kb == GOP_SER || s2.kb == GOP_X) continue;       // s2 = s1 +- q, q free of the chain
                size_t k1 = g.ser.size();
                for (size_t m = 0; m < k; m++) if (g.ser[m].dst == s2.ia) k1 = m;
                if (k1 == g.ser.size()) continue;
                const GenInsn s1 = g.ser[k1];
                if ((s1.op != 1 && s1.op != 2) || s1.kb == GOP_SER || s1.kb == GOP_X || !(s1.ka == GOP_SER || s1.ka == GOP_X)) continue;   // s1 = s0 +- p
                uint32_t uses = (g.res_kind == GOP_SER && g.res_idx == s1.dst) ? 1 : 0;
                for (const GenInsn& u : g.ser) uses += (u.ka == GOP_SER && u.ia == s1.dst) + (u.kb == GOP_SER && u.ib == s1.dst);
                if (uses != 1) continue;
                // s2 = (s0 +- p) +- q = s0 + r,  r = (+-p) +- q as a chain-free node: r = p + q | p - q | q - p | -(p + q) -> s0 - (p + q)
                const bool np = s1.op == 2, nq = s2.op == 2;
                GenInsn r{1, g.n_slots++, s1.kb, s1.ib, s2.kb, s2.ib};
                uint32_t top = 1;
                if (!np && nq) r.op = 2;                                   // p - q
                else if (np && !nq) { r.op = 2; std::swap(r.ka, r.kb); std::swap(r.ia, r.ib); }      // q - p
                else if (np && nq) top = 2;                                // s0 - (p + q)
                g.par.push_back(r);
                s2 = GenInsn{top, s2.dst, s1.ka, s1.ia, GOP_SLOT, r.dst};
                g.ser.erase(g.ser.begin() + (long)k1);
                changed = true;
            }
        }
    }
    // what travels: the main columns and the device-built auxiliary columns the general builders read come down (the main columns on the
    // copy stream, behind nothing but the trace itself - they do not wait for the kernel above), the general columns go up, each as soon
    // as it is finished; everything through ONE pinned block kept by the context (true asynchronous copies, no allocation per proof)
    std::vector<uint8_t> need_aux(p.A, 0);
    for (uint32_t c = 0; c < p.A; c++) if (p.has_add[c] == 4) for (const GenLoad& l : progs[c].loads) if (l.kind == GLD_AUX_CUR) need_aux[l.col] = 1;
    const size_t n_main = p.general_main_cols.size();
    uint64_t* const arena = ctx->host_arena((n_main + (size_t)p.A * F::DEG) * n * 8);
    std::vector<const uint64_t*> mcols(p.W, nullptr);
    uint64_t* const aux = arena + n_main * n;
    hipStream_t cs = ctx->get_copy_stream();
    AERO_HIP(hipStreamWaitEvent(cs, ev_trace_ready, 0));
    for (size_t k = 0; k < n_main; k++) {
        const uint32_t c = p.general_main_cols[k];
        mcols[c] = arena + k * n;
        AERO_HIP(hipMemcpyAsync(arena + k * n, trace_dev + (size_t)c * n, n * 8, hipMemcpyDeviceToHost, cs));
    }
    for (uint32_t c = 0; c < p.A; c++)
        if (need_aux[c] && p.has_add[c] != 4)
            AERO_HIP(hipMemcpyAsync(aux + (size_t)c * F::DEG * n, out + (size_t)c * F::DEG * n, (size_t)F::DEG * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    AERO_HIP(hipStreamSynchronize(cs));
    ctx->sync();
    // done[c] = rows of auxiliary column c that exist on the host (device-built columns: all of them)
    std::unique_ptr<std::atomic<uint64_t>[]> done(new std::atomic<uint64_t>[p.A]);
    for (uint32_t c = 0; c < p.A; c++) done[c].store(p.has_add[c] == 4 ? 0 : n, std::memory_order_relaxed);
    std::atomic<bool> failed{false};
    constexpr uint64_t CH = 4096;          // rows per chunk: its slot arrays stay in the core's cache
    auto run_column = [&](uint32_t c) {
        const Compiled& g = progs[c];
        std::vector<T> slots((size_t)(g.n_slots ? g.n_slots : 1) * CH), regs(g.n_serial ? g.n_serial : 1, F::zero());
        auto comp = [&](uint32_t col, int d) { return aux + ((size_t)col * F::DEG + d) * n; };
        auto put = [&](uint64_t i, T v) { for (int d = 0; d < F::DEG; d++) comp(c, d)[i] = F::comp(v, d); };
        put(0, init[c]);
        T x = init[c];
        done[c].store(1, std::memory_order_release);
        for (uint64_t i0 = 0; i0 + 1 < n; i0 += CH) {
            const uint64_t cnt = std::min<uint64_t>(CH, n - 1 - i0);
            // leaf operands of rows [i0, i0 + cnt) -> slot arrays; a column another thread is still building is waited for
            for (const GenLoad& l : g.loads) {
                T* s = slots.data() + (size_t)l.slot * CH;
                switch (l.kind) {
                    case GLD_MAIN_CUR: { const uint64_t* m = mcols[l.col] + i0; for (uint64_t k = 0; k < cnt; k++) s[k] = F::from(m[k]); break; }
                    case GLD_MAIN_NXT: { const uint64_t* m = mcols[l.col] + i0 + 1; for (uint64_t k = 0; k < cnt; k++) s[k] = F::from(m[k]); break; }
                    case GLD_AUX_CUR: {
                        while (done[l.col].load(std::memory_order_acquire) < i0 + cnt) { if (failed.load(std::memory_order_relaxed)) return; std::this_thread::yield(); }
                        const uint64_t* a0 = comp(l.col, 0) + i0;
                        const uint64_t* a1 = comp(l.col, F::DEG - 1) + i0;
                        for (uint64_t k = 0; k < cnt; k++) s[k] = F::make(a0[k], F::DEG > 1 ? a1[k] : 0);
                        break;
                    }
                    default: { const uint64_t* t = pt.tab.data() + l.col; for (uint64_t k = 0; k < cnt; k++) s[k] = F::from(t[(i0 + k) & l.mask]); }
                }
            }
            for (const GenInsn& I : g.par) {
                T* d = slots.data() + (size_t)I.dst * CH;
                const T* a = I.ka == GOP_SLOT ? slots.data() + (size_t)I.ia * CH : nullptr;
                const T* b = I.kb == GOP_SLOT ? slots.data() + (size_t)I.ib * CH : nullptr;
                const T ca = a ? F::zero() : g.consts[I.ia], cb = b ? F::zero() : g.consts[I.ib];
                if (I.op == 1) for (uint64_t k = 0; k < cnt; k++) d[k] = F::add(a ? a[k] : ca, b ? b[k] : cb);
                else if (I.op == 2) for (uint64_t k = 0; k < cnt; k++) d[k] = F::sub(a ? a[k] : ca, b ? b[k] : cb);
                else for (uint64_t k = 0; k < cnt; k++) d[k] = F::mul(a ? a[k] : ca, b ? b[k] : cb);
            }
            // the chain, row after row
            uint64_t* o0 = comp(c, 0) + i0 + 1;
            uint64_t* o1 = comp(c, F::DEG - 1) + i0 + 1;
            const GenInsn* ser = g.ser.data();
            const size_t ns = g.ser.size();
            for (uint64_t k = 0; k < cnt; k++) {
                auto val = [&](uint32_t kind, uint32_t idx) -> T {
                    return kind == GOP_SER ? regs[idx] : kind == GOP_X ? x : kind == GOP_SLOT ? slots[(size_t)idx * CH + k] : g.consts[idx];
                };
                for (size_t q = 0; q < ns; q++) {
                    const GenInsn& I = ser[q];
                    const T a = val(I.ka, I.ia), b = val(I.kb, I.ib);
                    regs[I.dst] = I.op == 1 ? F::add(a, b) : I.op == 2 ? F::sub(a, b) : F::mul(a, b);
                }
                x = val(g.res_kind, g.res_idx);
                o0[k] = F::comp(x, 0);
                if (F::DEG > 1) o1[k] = F::comp(x, F::DEG - 1);
            }
            done[c].store(i0 + cnt + 1, std::memory_order_release);
        }
    };
    std::vector<uint32_t> gcols;
    for (uint32_t c = 0; c < p.A; c++) if (p.has_add[c] == 4) gcols.push_back(c);
    auto upload = [&](uint32_t c) {
        AERO_HIP(hipMemcpyAsync(out + (size_t)c * F::DEG * n, aux + (size_t)c * F::DEG * n, (size_t)F::DEG * n * 8, hipMemcpyHostToDevice, ctx->stream));
    };
    if (gcols.size() == 1) { run_column(gcols[0]); upload(gcols[0]); }
    else {
        // a column reads only columns before it: the threads form a pipeline, each a chunk behind the ones it reads
        std::vector<std::thread> th;
        std::exception_ptr err;
        std::mutex err_mu;
        for (uint32_t c : gcols)
            th.emplace_back([&, c] {
                try { run_column(c); }
                catch (...) { failed.store(true); std::lock_guard<std::mutex> lk(err_mu); if (!err) err = std::current_exception(); }
            });
        for (size_t k = 0; k < th.size(); k++) {
            th[k].join();
            if (!failed.load()) upload(gcols[k]);       // goes up while the later columns are still being computed
        }
        if (err) std::rethrow_exception(err);
    }
    // no synchronisation: the pinned block stays the context's, and its next use is ordered behind these copies (host_arena)
}
template void air_build_aux<FB>(Context*, const Program&, const uint64_t*, int, const uint64_t*, const uint64_t*, uint64_t*);
template void air_build_aux<FQ>(Context*, const Program&, const uint64_t*, int, const uint64_t*, const gl::E2*, uint64_t*);

template <class F>
uint64_t air_validate_trace(Context* ctx, const Program& p, const Instance& in, const uint64_t* trace_dev, const uint64_t* aux_dev, const uint64_t* pub,
                            const typename F::T* rands) {
    const uint64_t n = in.n;
    const Scalars<F> sc = fold_scalars<F>(p, pub, rands);
    const PeriodicTables pt = periodic_tables(p, n, n, 1);       // on the trace domain the tables are the cycles themselves
    std::vector<uint64_t> words(sc.b.begin(), sc.b.end()), pdesc;
    const uint64_t oSE = words.size();
    for (auto& v : sc.e) for (int d = 0; d < F::DEG; d++) words.push_back(F::comp(v, d));
    words.push_back(0);
    for (size_t k = 0; k < pt.off.size(); k++) pdesc.push_back((uint64_t)pt.off[k] | ((uint64_t)pt.mask[k] << 32));
    // the values of the sequence assertions, raw, behind the periodic tables (the validation kernel reads entry (step - first) / stride)
    std::vector<uint64_t> ptab = pt.tab;
    for (auto& seq : p.sequences) { pdesc.push_back(ptab.size()); ptab.insert(ptab.end(), seq.begin(), seq.end()); }
    if (ptab.size() >= (1ull << 32)) fail("air program: tables too large", ST_UNSUPPORTED);
    const std::vector<uint64_t> init{~0ull};
    ParamPack pp(ctx);
    const size_t i_pool = pp.add(words), i_pt = pp.add(ptab), i_pd = pp.add(pdesc), i_flag = pp.add(init);
    pp.commit();
    AirConsArgs<F> a{};
    a.lde = trace_dev; a.aux = aux_dev; a.N = n; a.W = p.W; a.A = aux_dev ? p.A : 0; a.blowup = 1; a.ce_step = 1; a.split_log = 0;
    a.rows = n; a.first = 0; a.count = n;
    a.pool = pp.ptr<uint64_t>(i_pool); a.ptab = pp.ptr<uint64_t>(i_pt);
    a.out_h[0] = const_cast<uint64_t*>(pp.ptr<uint64_t>(i_flag));
    // mode 2: every constraint; mode 3: without auxiliary columns only the main segment's constraints and assertions
    if (!launch_air_jit<F>(ctx, p, in, a, pp.ptr<uint64_t>(i_pd), oSE, 0, 0, (aux_dev || !p.A) ? 2 : 3))
        fail("air program: trace validation needs the run-time compiled kernel (" + air_jit_last_error() + ")", ST_UNSUPPORTED);
    uint64_t flag = 0;
    ctx->fetch(&flag, a.out_h[0], 8);
    return flag;
}
template uint64_t air_validate_trace<FB>(Context*, const Program&, const Instance&, const uint64_t*, const uint64_t*, const uint64_t*, const uint64_t*);
template uint64_t air_validate_trace<FQ>(Context*, const Program&, const Instance&, const uint64_t*, const uint64_t*, const uint64_t*, const gl::E2*);

}  // namespace aero
