// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/gl.hpp header). Never linked into libaero_stark.so.
//
// CPU restatement of the AIR-as-data path: an AEROAIR program (format: include/aero_air.h) evaluated the way winter-prover /
// winter-air 0.4 evaluate an `Air` - the generic seam of the reference:
//   aero-sdk/miden-wasm/src/constraints_worker.rs:32-59   ProcessorAir::new -> ConstraintEvaluator::new -> evaluate_fragment
//   aero-sdk/miden-wasm/src/proving_worker.rs:374-437     divisors, ConstraintEvaluationTable, fragment stitching
//   aero-sdk/miden-wasm/src/proving_worker.rs:323-332     commit_to_trace_and_validate (build_aux_segment)
// The winter-air bodies are absent from the mount; what is restated (and mirrored for the verifier in the dead code of
// src/stark_verifier/air/transitions/evaluator.cairo:79-86,131-150,216-218 and air/boundary.cairo):
//   * TransitionConstraintDegree::get_evaluation_degree = base (n - 1) + sum (n / cycle)(cycle - 1); min_blowup_factor =
//     max(next_pow2(base + #cycles), 2); ce_blowup = the maximum over all constraints;
//   * transition constraint k is merged as (alpha_k + beta_k x^adj) t_k(x), adj = (ce_n - 1 + deg(divisor)) - evaluation degree,
//     divisor = (x^n - 1) / prod_{i=1..e} (x - w^(n-i));
//   * assertions sorted by (stride, first_step, column) take the boundary coefficients in that order, main then aux; one divisor per
//     (stride, first_step): x - w^step or x^(n/stride) - w^(step n/stride); adj = (ce_n - 1 + deg(divisor)) - (n - 1); aux
//     assertions join the main group with the same divisor, otherwise form a new column behind the main ones;
//   * periodic columns are the interpolants of one cycle evaluated at x^(n/cycle);
//   * `Assertion::sequence(column, first_step, stride, values)` (AEROAIR version 2): winter-air 0.4 `BoundaryConstraint::new` interpolates
//     the values over the subgroup of their own size and evaluates the polynomial at x * g^-first_step ("poly_offset"): the
//     constraint's numerator is column(x) - P(x g^-first), its divisor the periodic one for (stride, first_step);
//   * auxiliary builders of version 2 are affine recurrences column(i+1) = column(i) * num / den + add_num / add_den.
// This interpreter is deliberately naive (every node evaluated in E per row, exponentiations by square-and-multiply): it shares no
// code and no evaluation strategy with aero_amd/csrc/air_program.hpp / air_kernels.hip.
#pragma once
#include <cstring>
#include <functional>
#include "stark.hpp"

namespace orc {

struct ProgramAir {
    enum { KNOWN = 1 };
    enum { K_NODE = 0, K_MAIN_CUR, K_MAIN_NXT, K_AUX_CUR, K_AUX_NXT, K_PERIODIC, K_CONST, K_PUB, K_RAND, K_SEQ };
    static const uint32_t NONE = 0xFFFFFFFFu;
    static const uint32_t GENERAL = 0xFFFFFFFEu;     // builder.den: column(i+1) = num on (main row i, i+1; aux row i of columns <= its own)
    // ---- the program as written
    uint32_t W = 0, A = 0, R = 0, num_pub = 0, exemptions = 1;
    Col consts;
    std::vector<Col> periodic;
    std::vector<Col> sequences;        // version 2: value lists of `Assertion::sequence`
    struct Node { uint32_t op, a, b; };
    std::vector<Node> nodes;
    std::vector<uint8_t> node_deps;    // bit 0: depends on the auxiliary frame, bit 1: on the random elements
    struct Trans { uint32_t root, base; std::vector<uint32_t> cycles; };
    std::vector<Trans> trans;          // main constraints first, then aux
    size_t n_main_trans = 0;
    struct Assertion { uint32_t col; int64_t first; uint32_t stride, value; };
    std::vector<Assertion> masserts, aasserts;
    struct Builder { uint32_t init, num, den, add_num, add_den; };
    std::vector<Builder> builders;
    // ---- the instance (bind)
    int log_n = 0;
    Col pub;
    size_t Cb = 2;
    std::vector<uint64_t> tadj;        // degree adjustment exponent per transition constraint
    struct Member { uint32_t col, value, coef; bool aux; };
    struct Group { uint32_t stride; uint64_t first; uint64_t a, b, adj; std::vector<Member> members; };
    std::vector<Group> groups;         // one per distinct boundary divisor, in column order
    std::vector<Col> ppoly;            // periodic columns: interpolant coefficients
    std::vector<Col> spoly;            // sequences: interpolant of the values over the subgroup of their size

    size_t n() const { return (size_t)1 << log_n; }
    size_t num_transition() const { return trans.size(); }
    size_t num_assertions() const { return masserts.size() + aasserts.size(); }
    size_t ce_blowup() const { return Cb; }
    size_t num_columns() const { return 1 + groups.size(); }
    const Col& pub_elements() const { return pub; }

    static uint32_t kind(uint32_t ref) { return ref >> 24; }
    static uint32_t index(uint32_t ref) { return ref & 0xFFFFFFu; }

    static ProgramAir parse(const uint8_t* p, size_t len) {
        size_t off = 0;
        auto need = [&](size_t k) { if (off + k > len) throw Err("air program: truncated"); };
        auto u32 = [&]() { need(4); uint32_t v; memcpy(&v, p + off, 4); off += 4; return v; };
        auto u64 = [&]() { need(8); uint64_t v; memcpy(&v, p + off, 8); off += 8; if (v >= P) throw Err("air program: non-canonical element"); return v; };
        need(8);
        if (memcmp(p, "AEROAIR", 7) != 0 || (p[7] != 1 && p[7] != 2)) throw Err("air program: bad magic");
        const int version = p[7];
        off = 8;
        uint32_t h[16];
        for (auto& v : h) v = u32();
        ProgramAir a;
        a.W = h[0]; a.A = h[1]; a.R = h[2]; a.num_pub = h[3]; a.exemptions = h[4];
        const uint32_t nc = h[5], np = h[6], nn = h[7], nmt = h[8], nat = h[9], nma = h[10], naa = h[11], nb = h[12];
        if (a.W < 1 || a.W > 255 || a.A > 255 - a.W || (a.A == 0) != (a.R == 0) || a.R > 255 || a.exemptions < 1) throw Err("air program: bad header");
        if (nb != 0 && nb != a.A) throw Err("air program: one builder per aux column or none");
        const uint32_t nseq = version >= 2 ? h[13] : 0;
        if ((version < 2 && h[13]) || h[14] || h[15]) throw Err("air program: reserved header words");
        if ((size_t)nc * 8 + (size_t)nn * 12 > len) throw Err("air program: truncated");
        for (uint32_t i = 0; i < nc; i++) a.consts.push_back(u64());
        for (uint32_t i = 0; i < np; i++) {
            uint32_t cl = u32();
            if (cl < 2 || (cl & (cl - 1)) || (size_t)cl * 8 > len) throw Err("air program: bad periodic column");
            Col v(cl);
            for (auto& x : v) x = u64();
            a.periodic.push_back(v);
        }
        for (uint32_t i = 0; i < nseq; i++) {
            uint32_t cnt = u32();
            if (cnt < 2 || (cnt & (cnt - 1)) || (size_t)cnt * 8 > len) throw Err("air program: bad sequence");
            Col v(cnt);
            for (auto& x : v) x = u64();
            a.sequences.push_back(v);
        }
        auto check_ref = [&](uint32_t ref, uint32_t node_limit) {
            const uint32_t k = kind(ref), i = index(ref);
            const uint32_t lim[9] = {node_limit, a.W, a.W, a.A, a.A, (uint32_t)a.periodic.size(), (uint32_t)a.consts.size(), a.num_pub, a.R};
            if (k > 8 || i >= lim[k]) throw Err("air program: operand out of range");
        };
        for (uint32_t i = 0; i < nn; i++) {
            Node nd{u32(), u32(), u32()};
            if (nd.op < 1 || nd.op > 3) throw Err("air program: bad opcode");
            check_ref(nd.a, i); check_ref(nd.b, i);
            uint8_t dep = 0;
            for (uint32_t ref : {nd.a, nd.b}) {
                const uint32_t k = kind(ref);
                if (k == K_AUX_CUR || k == K_AUX_NXT) dep |= 1;
                if (k == K_RAND) dep |= 2;
                if (k == K_NODE) dep |= a.node_deps[index(ref)];
            }
            a.nodes.push_back(nd);
            a.node_deps.push_back(dep);
        }
        a.n_main_trans = nmt;
        for (uint32_t i = 0; i < nmt + nat; i++) {
            Trans t; t.root = u32(); t.base = u32();
            uint32_t ncy = u32();
            if (ncy > 64) throw Err("air program: too many cycles");
            for (uint32_t j = 0; j < ncy; j++) t.cycles.push_back(u32());
            check_ref(t.root, nn);
            a.trans.push_back(t);
        }
        for (uint32_t i = 0; i < nma + naa; i++) {
            Assertion s; s.col = u32(); s.first = (int32_t)u32(); s.stride = u32(); s.value = u32();
            if (kind(s.value) == K_SEQ) { if (index(s.value) >= nseq || !s.stride) throw Err("air program: bad sequence assertion"); }
            else check_ref(s.value, nn);
            if (s.col >= (i < nma ? a.W : a.A)) throw Err("air program: assertion column out of range");
            (i < nma ? a.masserts : a.aasserts).push_back(s);
        }
        for (uint32_t i = 0; i < nb; i++) {
            Builder b{u32(), u32(), u32(), NONE, NONE};
            if (version >= 2) { b.add_num = u32(); b.add_den = u32(); }
            check_ref(b.init, nn); check_ref(b.num, nn);
            if (b.den != NONE && b.den != GENERAL) check_ref(b.den, nn);
            if (b.den == GENERAL && (version < 2 || b.add_num != NONE || b.add_den != NONE)) throw Err("air program: bad general builder");
            if (b.add_num != NONE) check_ref(b.add_num, nn);
            if (b.add_den != NONE) { if (b.add_num == NONE) throw Err("air program: builder denominator without a numerator"); check_ref(b.add_den, nn); }
            a.builders.push_back(b);
        }
        if (off != len) throw Err("air program: trailing bytes");
        return a;
    }

    // everything that depends on the trace length and the public inputs
    void bind(int log_n_, const Col& pub_) {
        log_n = log_n_; pub = pub_;
        if (pub.size() != num_pub) throw Err("air program: wrong number of public inputs");
        const uint64_t n_ = n();
        Cb = 2;
        for (auto& t : trans) { size_t d = t.base + t.cycles.size(), e = 2; while (e < d) e <<= 1; if (e > Cb) Cb = e; }
        const uint64_t ce_n = Cb * n_;
        if (exemptions >= n_) throw Err("air program: more exemptions than trace steps");
        tadj.clear();
        for (auto& t : trans) {
            uint64_t ed = (uint64_t)t.base * (n_ - 1);
            for (uint32_t c : t.cycles) { if (c < 2 || (c & (c - 1)) || c > n_) throw Err("air program: bad cycle length"); ed += (n_ / c) * (c - 1); }
            const uint64_t target = ce_n - 1 + (n_ - exemptions);
            if (ed > target) throw Err("air program: constraint degree exceeds the composition degree");
            tadj.push_back(target - ed);
        }
        for (auto& v : periodic) if (v.size() > n_) throw Err("air program: periodic cycle longer than the trace");
        ppoly.clear();
        for (auto& v : periodic) { Col c = v; intt(c.data(), c.size()); ppoly.push_back(c); }
        spoly.clear();
        for (auto& v : sequences) { Col c = v; intt(c.data(), c.size()); spoly.push_back(c); }
        // assertions: resolve steps, sort, hand out coefficients, group by divisor
        const uint64_t g = gl_root_of_unity(log_n);
        groups.clear();
        uint32_t coef = 0;
        for (int seg = 0; seg < 2; seg++) {
            std::vector<Assertion> v = seg == 0 ? masserts : aasserts;
            for (auto& s : v) {
                if (s.first < 0) s.first += (int64_t)n_;
                if (s.first < 0 || (uint64_t)s.first >= n_) throw Err("air program: assertion step out of range");
                if (s.stride && (s.stride < 2 || (s.stride & (s.stride - 1)) || s.stride >= n_ || (uint64_t)s.first >= s.stride)) throw Err("air program: bad assertion stride");
                if (kind(s.value) == K_SEQ && (uint64_t)sequences[index(s.value)].size() * s.stride != n_) throw Err("air program: a sequence assertion needs stride * values = trace length");
            }
            std::stable_sort(v.begin(), v.end(), [](const Assertion& x, const Assertion& y) {
                if (x.stride != y.stride) return x.stride < y.stride;
                if (x.first != y.first) return x.first < y.first;
                return x.col < y.col;
            });
            for (size_t i = 1; i < v.size(); i++)
                if (v[i].col == v[i - 1].col && v[i].stride == v[i - 1].stride && v[i].first == v[i - 1].first) throw Err("air program: duplicate assertion");
            for (auto& s : v) {
                Group* gp = nullptr;
                // aux assertions join a main group with the same divisor; within one segment groups appear in sorted key order
                for (size_t j = 0; j < groups.size(); j++)
                    if (groups[j].stride == s.stride && groups[j].first == (uint64_t)s.first) { gp = &groups[j]; break; }
                if (!gp) {
                    Group ng; ng.stride = s.stride; ng.first = (uint64_t)s.first;
                    ng.a = s.stride ? n_ / s.stride : 1;
                    ng.b = gl_pow(g, ng.first * ng.a);
                    ng.adj = (ce_n - 1 + ng.a) - (n_ - 1);
                    groups.push_back(ng);
                    gp = &groups.back();
                }
                gp->members.push_back(Member{s.col, s.value, coef++, seg == 1});
            }
        }
    }

    // ---- evaluation over E --------------------------------------------------------------------------------------------
    template <class F> struct Frame {
        typedef typename F::T T;
        const T *mcur, *mnxt, *acur, *anxt, *rands, *per;
    };
    template <class F> typename F::T operand(uint32_t ref, const Frame<F>& f, const std::vector<typename F::T>& vals) const {
        const uint32_t i = index(ref);
        switch (kind(ref)) {
            case K_NODE: return vals[i];
            case K_MAIN_CUR: return f.mcur[i];
            case K_MAIN_NXT: return f.mnxt[i];
            case K_AUX_CUR: return f.acur[i];
            case K_AUX_NXT: return f.anxt[i];
            case K_PERIODIC: return f.per[i];
            case K_CONST: return F::from(consts[i]);
            case K_PUB: return F::from(pub[i]);
            default: return f.rands[i];
        }
    }
    template <class F> void run_nodes(const Frame<F>& f, std::vector<typename F::T>& vals) const {
        vals.resize(nodes.size());
        const uint8_t skip = (f.acur ? 0 : 1) | (f.rands ? 0 : 2);   // nodes whose inputs this frame does not carry
        for (size_t i = 0; i < nodes.size(); i++) {
            if (node_deps[i] & skip) continue;
            const typename F::T a = operand<F>(nodes[i].a, f, vals), b = operand<F>(nodes[i].b, f, vals);
            vals[i] = nodes[i].op == 1 ? F::add(a, b) : nodes[i].op == 2 ? F::sub(a, b) : F::mul(a, b);
        }
    }
    template <class F> void periodic_at(typename F::T x, std::vector<typename F::T>& out) const {
        out.resize(ppoly.size());
        for (size_t k = 0; k < ppoly.size(); k++) {
            const typename F::T y = f_pow<F>(x, n() / ppoly[k].size());
            typename F::T acc = F::zero();
            for (size_t i = ppoly[k].size(); i-- > 0;) acc = F::add(F::mul(acc, y), F::from(ppoly[k][i]));
            out[k] = acc;
        }
    }
    // numerators of every column at the point x, frame given in E
    template <class F, class CC> void eval_common(const CC& cc, const typename F::T* mcur, const typename F::T* mnxt, const typename F::T* acur,
                                                  const typename F::T* anxt, const typename F::T* rands, typename F::T x, typename F::T* out) const {
        typedef typename F::T T;
        std::vector<T> per, vals;
        periodic_at<F>(x, per);
        Frame<F> f{mcur, mnxt, acur, anxt, rands, per.data()};
        run_nodes<F>(f, vals);
        T acc = F::zero();
        for (size_t k = 0; k < trans.size(); k++) {
            const T t = operand<F>(trans[k].root, f, vals);
            acc = F::add(acc, F::mul(F::add(cc.ta[k], F::mul(cc.tb[k], f_pow<F>(x, tadj[k]))), t));
        }
        out[0] = acc;
        for (size_t j = 0; j < groups.size(); j++) {
            const T xp = f_pow<F>(x, groups[j].adj);
            T s = F::zero();
            for (auto& m : groups[j].members) {
                const T v = m.aux ? acur[m.col] : mcur[m.col];
                T want;
                if (kind(m.value) == K_SEQ) {
                    // P(x g^-first): P interpolates the values over the subgroup of their own size (w_n^stride generates it)
                    const Col& pc = spoly[index(m.value)];
                    const T y = F::mulb(x, gl_inv(gl_pow(gl_root_of_unity(log_n), groups[j].first)));
                    want = F::zero();
                    for (size_t i = pc.size(); i-- > 0;) want = F::add(F::mul(want, y), F::from(pc[i]));
                } else want = operand<F>(m.value, f, vals);
                s = F::add(s, F::mul(F::add(cc.ba[m.coef], F::mul(cc.bb[m.coef], xp)), F::sub(v, want)));
            }
            out[1 + j] = s;
        }
    }
    // divisor of column j at x as a fraction num / den
    template <class F> void divisor_at(size_t j, typename F::T x, typename F::T& num, typename F::T& den) const {
        typedef typename F::T T;
        if (j == 0) {
            num = F::sub(f_pow<F>(x, n()), F::one());
            den = F::one();
            const uint64_t g = gl_root_of_unity(log_n);
            for (uint32_t i = 1; i <= exemptions; i++) den = F::mul(den, F::sub(x, F::from(gl_pow(g, n() - i))));
        } else {
            const Group& gr = groups[j - 1];
            num = F::sub(f_pow<F>(x, gr.a), F::from(gr.b));
            den = F::one();
        }
        (void)sizeof(T);
    }

    // ---- the model interface (see FibAir in oracle/stark.hpp) -----------------------------------------------------------
    template <class F, class CC> void eval_row(const CC& cc, const uint64_t* cur, const uint64_t* nxt, const typename F::T* acur,
                                               const typename F::T* anxt, const typename F::T* rands, uint64_t x, typename F::T* out) const {
        typedef typename F::T T;
        std::vector<T> mc(W), mn(W);
        for (uint32_t c = 0; c < W; c++) { mc[c] = F::from(cur[c]); mn[c] = F::from(nxt[c]); }
        eval_common<F>(cc, mc.data(), mn.data(), acur, anxt, rands, F::from(x), out);
    }
    template <class F, class CC> typename F::T ood_lhs(const CC& cc, const typename F::T* ood_cur, const typename F::T* ood_next,
                                                       const typename F::T* rands, typename F::T z) const {
        typedef typename F::T T;
        std::vector<T> num(num_columns());
        eval_common<F>(cc, ood_cur, ood_next, ood_cur + W, ood_next + W, rands, z, num.data());
        T lhs = F::zero();
        for (size_t j = 0; j < num.size(); j++) {
            T dn, dd;
            divisor_at<F>(j, z, dn, dd);
            lhs = F::add(lhs, F::mul(num[j], F::mul(dd, F::inv(dn))));
        }
        return lhs;
    }
    template <class F> void divide(const std::vector<std::vector<typename F::T>>& ce, std::vector<Col>& hcomp) const {
        typedef typename F::T T;
        const size_t ceN = Cb * n(), NC = num_columns();
        const uint64_t gce = gl_root_of_unity(ilog2(ceN));
#pragma omp parallel for schedule(static)
        for (size_t s = 0; s < ceN; s++) {
            const uint64_t x = gl_mul(GEN, gl_pow(gce, s));
            T h = F::zero();
            for (size_t j = 0; j < NC; j++) {
                uint64_t dn, dd;
                divisor_at<FB>(j, x, dn, dd);
                h = F::add(h, F::mulb(ce[j][s], gl_mul(dd, gl_inv(dn))));
            }
            for (int k = 0; k < F::DEG; k++) hcomp[k][s] = F::comp(h, k);
        }
    }
    // `build_aux_segment`: column(0) = init, column(i+1) = column(i) * num(i) / den(i) on the frame (row i, row i+1 mod n)
    template <class F> void build_aux(const std::vector<Col>& trace, const std::vector<typename F::T>& rands, std::vector<Col>& acols) const {
        typedef typename F::T T;
        if (builders.size() != A) throw Err("air program: no aux builders, the auxiliary columns cannot be constructed");
        const size_t n_ = n();
        std::vector<std::vector<T>> mult(A, std::vector<T>(n_)), addv(A);
        for (uint32_t c = 0; c < A; c++) if (builders[c].add_num != NONE) addv[c].assign(n_, F::zero());
        std::vector<T> init(A);
#pragma omp parallel
        {
            std::vector<T> mc(W), mn(W), per(periodic.size()), vals;
#pragma omp for schedule(static)
            for (size_t i = 0; i < n_; i++) {
                const size_t ni = (i + 1) % n_;
                for (uint32_t c = 0; c < W; c++) { mc[c] = F::from(trace[c][i]); mn[c] = F::from(trace[c][ni]); }
                for (size_t k = 0; k < periodic.size(); k++) per[k] = F::from(periodic[k][i % periodic[k].size()]);
                Frame<F> f{mc.data(), mn.data(), nullptr, nullptr, rands.data(), per.data()};
                run_nodes<F>(f, vals);
                for (uint32_t c = 0; c < A; c++) {
                    if (builders[c].den == GENERAL) { if (i == 0) init[c] = operand<F>(builders[c].init, f, vals); continue; }
                    T m = operand<F>(builders[c].num, f, vals);
                    if (builders[c].den != NONE) m = F::mul(m, F::inv(operand<F>(builders[c].den, f, vals)));
                    mult[c][i] = m;
                    if (builders[c].add_num != NONE) {
                        T t = operand<F>(builders[c].add_num, f, vals);
                        if (builders[c].add_den != NONE) t = F::mul(t, F::inv(operand<F>(builders[c].add_den, f, vals)));
                        addv[c][i] = t;
                    }
                    if (i == 0) init[c] = operand<F>(builders[c].init, f, vals);
                }
            }
        }
#pragma omp parallel for schedule(dynamic, 1)
        for (uint32_t c = 0; c < A; c++) {
            if (builders[c].den == GENERAL) continue;
            T p = init[c];
            for (size_t i = 0; i < n_; i++) {
                for (int k = 0; k < F::DEG; k++) acols[c * F::DEG + k][i] = F::comp(p, k);
                p = F::mul(p, mult[c][i]);
                if (!addv[c].empty()) p = F::add(p, addv[c][i]);
            }
        }
        // general recurrences, in column order, row after row: the expression sees the main frame and the current row of the
        // auxiliary columns up to its own (every node is evaluated recursively - no sharing, no ordering tricks)
        for (uint32_t c = 0; c < A; c++) {
            if (builders[c].den != GENERAL) continue;
            T p = init[c];
            for (size_t i = 0; i < n_; i++) {
                for (int k = 0; k < F::DEG; k++) acols[c * F::DEG + k][i] = F::comp(p, k);
                if (i + 1 == n_) break;
                std::function<T(uint32_t)> ev = [&](uint32_t ref) -> T {
                    const uint32_t j = index(ref);
                    switch (kind(ref)) {
                        case K_NODE: { const T a = ev(nodes[j].a), b = ev(nodes[j].b); return nodes[j].op == 1 ? F::add(a, b) : nodes[j].op == 2 ? F::sub(a, b) : F::mul(a, b); }
                        case K_MAIN_CUR: return F::from(trace[j][i]);
                        case K_MAIN_NXT: return F::from(trace[j][i + 1]);
                        case K_AUX_CUR: {
                            if (j > c) throw Err("air program: a general builder reads a later auxiliary column");
                            const uint64_t comp[2] = {acols[j * F::DEG][i], F::DEG > 1 ? acols[j * F::DEG + F::DEG - 1][i] : 0};
                            return F::make(comp);
                        }
                        case K_PERIODIC: return F::from(periodic[j][i % periodic[j].size()]);
                        case K_CONST: return F::from(consts[j]);
                        case K_PUB: return F::from(pub[j]);
                        case K_RAND: return rands[j];
                        default: throw Err("air program: operand not available to a general builder");
                    }
                };
                p = ev(builders[c].num);
            }
        }
    }
    // does the trace satisfy the program? (test helper: main transition constraints and main assertions, over the base field)
    std::string check_main_trace(const std::vector<Col>& trace) const {
        const size_t n_ = n();
        std::vector<uint64_t> mc(W), mn(W), per(periodic.size()), vals;
        for (size_t i = 0; i + exemptions < n_; i++) {
            for (uint32_t c = 0; c < W; c++) { mc[c] = trace[c][i]; mn[c] = trace[c][i + 1]; }
            for (size_t k = 0; k < periodic.size(); k++) per[k] = periodic[k][i % periodic[k].size()];
            Frame<FB> f{mc.data(), mn.data(), nullptr, nullptr, nullptr, per.data()};
            run_nodes<FB>(f, vals);
            for (size_t k = 0; k < n_main_trans; k++)
                if (operand<FB>(trans[k].root, f, vals) != 0) return "transition constraint " + std::to_string(k) + " fails at step " + std::to_string(i);
        }
        return "";
    }
};

}  // namespace orc
