// Per-proof device tables + launches of a program AIR (see air_host.hpp).
#include "air_host.hpp"

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
std::vector<Insn> padded(const std::vector<Insn>& code) {
    std::vector<Insn> c = code;
    c.push_back(Insn{OP_END, 0, 0, 0});      // the interpreter reads one instruction ahead
    c.push_back(Insn{OP_END, 0, 0, 0});
    return c;
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
        d.groups.push_back(AirBGroupDev{bg.a & (rows - 1), gl::pow(h, bg.a), bg.b, bg.adj & (rows - 1), bg.m0, bg.count});
    if (d.groups.empty()) d.groups.push_back(AirBGroupDev{});
    return d;
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
    std::vector<T> tb(cc.tb), bb(cc.bb);
    std::vector<uint64_t> dg_exp;
    {
        std::vector<uint64_t> hadj;
        for (uint64_t adj : in.dgroup_adj) { hadj.push_back(gl::pow(h, adj)); dg_exp.push_back(adj & (rows - 1)); }
        for (size_t k = 0; k < tb.size(); k++) tb[k] = F::mulb(tb[k], hadj[p.trans[k].group]);
        for (auto& bg : in.bgroups) {
            const uint64_t ha = gl::pow(h, bg.adj);
            for (uint32_t m = bg.m0; m < bg.m0 + bg.count; m++) bb[in.members[m].coef] = F::mulb(bb[in.members[m].coef], ha);
        }
    }
    const PeriodicTables pt = periodic_tables(p, in.n, rows, h);
    const DivisorTables dv = divisor_tables(p, in, rows, h);
    const std::vector<Insn> code = padded(p.cons_code);
    std::vector<BoundaryMember> members = in.members;
    if (members.empty()) members.push_back(BoundaryMember{});
    std::vector<uint64_t> scb = sc.b;
    std::vector<T> sce = sc.e;
    if (scb.empty()) scb.push_back(0);
    if (sce.empty()) sce.push_back(F::zero());
    if (bb.empty()) bb.push_back(F::zero());
    std::vector<T> ba(cc.ba);
    if (ba.empty()) ba.push_back(F::zero());
    ParamPack pp(ctx);
    const size_t i_code = pp.add(code), i_scb = pp.add(scb), i_sce = pp.add(sce), i_pt = pp.add(pt.tab), i_po = pp.add(pt.off), i_pm = pp.add(pt.mask),
                 i_ta = pp.add(cc.ta), i_tb = pp.add(tb), i_dg = pp.add(dg_exp), i_bg = pp.add(dv.groups), i_mem = pp.add(members), i_ba = pp.add(ba),
                 i_bb = pp.add(bb), i_zn = pp.add(dv.zn_inv), i_ex = pp.add(dv.exempt);
    pp.commit();
    NttTables* tw = ctx->ntt_tables(ilog2z(rows));
    AirConsArgs<F> a{};
    a.lde = g.lde; a.aux = g.aux; a.N = g.frame_rows; a.W = p.W; a.A = p.A;
    a.blowup = (uint32_t)(g.frame_rows / in.n); a.ce_step = (uint32_t)(g.frame_rows / rows); a.split_log = g.split_log;
    a.rows = rows; a.first = g.first; a.count = g.count;
    a.code = pp.ptr<Insn>(i_code); a.slotsB = p.cons_slotsB; a.slotsE = p.cons_slotsE;
    a.scalB = pp.ptr<uint64_t>(i_scb); a.scalE = pp.ptr<T>(i_sce);
    a.ptab = pp.ptr<uint64_t>(i_pt); a.p_off = pp.ptr<uint32_t>(i_po); a.p_mask = pp.ptr<uint32_t>(i_pm);
    a.ta = pp.ptr<T>(i_ta); a.tb = pp.ptr<T>(i_tb); a.dg_exp = pp.ptr<uint64_t>(i_dg);
    a.bgroups = pp.ptr<AirBGroupDev>(i_bg); a.n_bgroups = (uint32_t)in.bgroups.size(); a.members = pp.ptr<BoundaryMember>(i_mem);
    a.ba = pp.ptr<T>(i_ba); a.bb = pp.ptr<T>(i_bb);
    a.tw_lo = tw->lo_fwd; a.tw_hi = tw->hi_fwd; a.tw_h = tw->h; a.offset = h;
    a.zn_inv = pp.ptr<uint64_t>(i_zn); a.xmask = (uint32_t)(rows / in.n) - 1;
    a.exempt = pp.ptr<uint64_t>(i_ex); a.n_exempt = p.exemptions;
    a.out_cols = out_cols;
    if (out_h) { a.out_h[0] = out_h[0]; a.out_h[1] = out_h[1]; }
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
    const std::vector<Insn> code = padded(p.aux_code);
    std::vector<T> init(p.A);
    for (uint32_t c = 0; c < p.A; c++) {
        const DOperand o = device_operand(p, p.builders[c].init);
        init[c] = o.kind == D_SCAL_E ? sc.e[o.idx] : F::from(sc.b[o.idx]);
    }
    std::vector<uint64_t> scb = sc.b;
    std::vector<T> sce = sc.e;
    if (scb.empty()) scb.push_back(0);
    if (sce.empty()) sce.push_back(F::zero());
    ParamPack pp(ctx);
    const size_t i_code = pp.add(code), i_scb = pp.add(scb), i_sce = pp.add(sce), i_pt = pp.add(pt.tab), i_po = pp.add(pt.off), i_pm = pp.add(pt.mask),
                 i_hd = pp.add(p.has_den), i_in = pp.add(init);
    pp.commit();
    AirAuxArgs<F> a{};
    a.trace = trace_dev; a.n = n; a.W = p.W; a.A = p.A;
    a.code = pp.ptr<Insn>(i_code); a.slotsB = p.aux_slotsB; a.slotsE = p.aux_slotsE;
    a.scalB = pp.ptr<uint64_t>(i_scb); a.scalE = pp.ptr<T>(i_sce);
    a.ptab = pp.ptr<uint64_t>(i_pt); a.p_off = pp.ptr<uint32_t>(i_po); a.p_mask = pp.ptr<uint32_t>(i_pm);
    a.has_den = pp.ptr<uint8_t>(i_hd); a.init = pp.ptr<T>(i_in); a.out = out;
    launch_air_aux<F>(ctx, a, p.has_den);
}
template void air_build_aux<FB>(Context*, const Program&, const uint64_t*, int, const uint64_t*, const uint64_t*, uint64_t*);
template void air_build_aux<FQ>(Context*, const Program&, const uint64_t*, int, const uint64_t*, const gl::E2*, uint64_t*);

}  // namespace aero
