// Host-side proof verifier of libaero_stark.so (no GPU work): the counterpart of `winter_verifier::verify` for proofs in the
// byte layout this backend emits, written against the in-tree specification of the reference — the Cairo verifier:
//   /root/reference/src/stark_verifier/stark_verifier.cairo:105-304   (transcript order, OOD frame, PoW, queries)
//   crypto/random.cairo (coin), channel.cairo:80-424 (openings, BatchMerkleProof), composer.cairo:17-316 (DEEP),
//   fri/fri_verifier.cairo:56-82,208-340,396-461 (layer openings, folding, remainder)
// Unlike the Cairo code (stark_verifier.cairo:151-159,183-187 leave it commented out) the OOD constraint consistency check is
// performed when the AIR is known (the built-in FibAir with its optional auxiliary segment); with an unknown AIR (`air` = NULL,
// e.g. the reference's golden Miden proof proofs/fib.bin) everything else is verified, which is exactly what the Cairo verifier does.
// This file shares no code with oracle/ (the test oracle has its own verifier); the two are compared in tests/test_verifier_cpu.py.
#include <algorithm>
#include <cstring>
#include <map>

#include "../../include/aero_air.h"
#include "air_program.hpp"
#include "prover.hpp"
#include "proof_format.hpp"

namespace aero {
namespace {

using gl::FB;
using gl::FQ;

[[noreturn]] void reject(const std::string& why) { throw Error(AERO_E_VERIFY, "verify: " + why); }

using fmt::Parsed;
using fmt::parse;
using fmt::rd64;
using fmt::batch_root;

template <class F> typename F::T rd_elem(const Bytes& b, size_t idx) {
    return F::make(rd64(b, idx * F::DEG), F::DEG > 1 ? rd64(b, idx * F::DEG + 1) : 0);
}
template <class F> Digest hash_elems(const std::vector<typename F::T>& v) {
    std::vector<uint64_t> flat;
    for (auto& e : v) for (int d = 0; d < F::DEG; d++) flat.push_back(F::comp(e, d));
    return b2s::hash_elements(flat.data(), (uint32_t)flat.size());
}
bool same(const Digest& a, const Digest& b) { return memcmp(a.w, b.w, sizeof a.w) == 0; }
template <class F> bool feq(typename F::T a, typename F::T b) {
    for (int d = 0; d < F::DEG; d++) if (F::comp(a, d) != F::comp(b, d)) return false;
    return true;
}
int leading_zero_bits(const Digest& d) {      // MSB-first from digest byte 0 (random.cairo:282-316)
    const uint8_t* b = reinterpret_cast<const uint8_t*>(d.w);
    int z = 0;
    for (int i = 0; i < 32; i++) {
        if (b[i] == 0) { z += 8; continue; }
        for (int k = 7; k >= 0 && !((b[i] >> k) & 1); k--) z++;
        break;
    }
    return z;
}

// in-place inverse transform over <w_n> (values in natural order -> coefficients in natural order), n <= 2^16
void intt_host(std::vector<uint64_t>& a) {
    const size_t n = a.size();
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    for (size_t i = 0; i < n; i++) { size_t j = gl::bitrev((uint32_t)i, lg); if (i < j) std::swap(a[i], a[j]); }
    for (int s = 1; s <= lg; s++) {
        const size_t m = (size_t)1 << s;
        const uint64_t wm = gl::inv(gl::root_of_unity(s));
        for (size_t k = 0; k < n; k += m) {
            uint64_t w = 1;
            for (size_t j = 0; j < m / 2; j++) {
                const uint64_t t = gl::mul(w, a[k + j + m / 2]), u = a[k + j];
                a[k + j] = gl::add(u, t); a[k + j + m / 2] = gl::sub(u, t);
                w = gl::mul(w, wm);
            }
        }
    }
    const uint64_t ninv = gl::inv(n);
    for (auto& v : a) v = gl::mul(v, ninv);
}

template <class F> void verify_impl(const Parsed& pr, const std::vector<uint64_t>& pub, const aero_fib_air* air_desc, const aero_verify_policy& policy,
                                    const air::Program* prog = nullptr) {
    typedef typename F::T T;
    const size_t n = (size_t)1 << pr.log_n, B = pr.opt.blowup_factor, N = n * B, Fd = pr.opt.fri_folding_factor;
    const uint32_t W = pr.W, A = pr.A, TW = W + A;
    const size_t EB = 8 * F::DEG;
    if (pr.log_n < 1 || pr.log_n > 28 || N > ((size_t)1 << gl::TWO_ADICITY)) reject("unsupported trace length");
    // ---- acceptance policy: what the CALLER requires of the proof's self-declared parameters. Without it a forger picks
    // num_queries = 1, grinding = 0, blowup = 2 and an arbitrary trace length, and "verification" means ~1 bit of soundness
    // (Winterfell 0.4 had the same gap; upstream later added AcceptableOptions).
    {
        int log_b = 0;
        while ((1u << log_b) < B) log_b++;
        const uint32_t query_bits = (uint32_t)pr.opt.num_queries * (uint32_t)log_b + pr.opt.grinding_factor;
        if (query_bits < policy.min_query_security_bits)
            reject("proof options give " + std::to_string(query_bits) + " query-security bits (num_queries * log2(blowup) + grinding), the policy requires " +
                   std::to_string(policy.min_query_security_bits));
        // Winterfell's conjectured security is min(query term, field term): 64 * extension degree - log2(LDE domain) caps it
        const uint32_t field_bits = 64u * (uint32_t)F::DEG - (uint32_t)(pr.log_n + log_b);
        if (policy.min_conjectured_security_bits && std::min(query_bits, field_bits) < policy.min_conjectured_security_bits)
            reject("conjectured security is min(" + std::to_string(query_bits) + " query bits, " + std::to_string(field_bits) + " field bits), the policy requires " +
                   std::to_string(policy.min_conjectured_security_bits) + (field_bits < policy.min_conjectured_security_bits ? " (an extension field is needed)" : ""));
        if (policy.expected_log_n && (int)policy.expected_log_n != pr.log_n)
            reject("trace length 2^" + std::to_string(pr.log_n) + " is not the expected 2^" + std::to_string(policy.expected_log_n));
        if (policy.require_options) {
            const aero_proof_options& q = policy.options;
            const ProofOptions& o = pr.opt;
            if (q.num_queries != o.num_queries || q.blowup_factor != o.blowup_factor || q.grinding_factor != o.grinding_factor || q.hash_fn != o.hash_fn ||
                q.field_extension != o.field_extension || q.fri_folding_factor != o.fri_folding_factor || q.fri_log_max_remainder != o.fri_log_max_remainder)
                reject("proof options differ from the options the policy requires");
        }
        if (policy.cairo_compat) {
            // the shape src/stark_verifier hard-codes (fri_verifier.cairo:22-25, channel.cairo:28,360-370, composer.cairo:24,159-311,
            // air_instance.cairo:96-104): only such proofs can be accepted by the reference's Cairo verifier
            if (W != 72 || A != 9) reject("cairo-compat: the Cairo verifier hard-codes 72 main and 9 auxiliary columns");
            if (pr.ood_evaluations.size() != 8 * 8) reject("cairo-compat: the Cairo verifier hard-codes 8 composition columns");
            if (pr.opt.num_queries != 27) reject("cairo-compat: the Cairo verifier hard-codes 27 queries");
            if (Fd != 8) reject("cairo-compat: the Cairo verifier hard-codes FRI folding factor 8");
            if (pr.opt.field_extension != EXT_NONE) reject("cairo-compat: the Cairo verifier has no extension field");
            if (B != 8) reject("cairo-compat: the Cairo verifier hard-codes blowup 8 (log_blowup 3)");
        }
    }
    if (pr.ood_evaluations.size() % EB || pr.ood_evaluations.empty()) reject("bad OOD evaluations length");
    const size_t C = pr.ood_evaluations.size() / EB;
    const int layers = num_fri_layers(N, Fd, 1ull << pr.opt.fri_log_max_remainder);
    const size_t nroots = (A ? 2 : 1) + 1 + layers + 1;
    if (pr.commitments.size() != 32 * nroots) reject("wrong number of commitments");
    std::vector<Digest> roots(nroots);
    for (size_t i = 0; i < nroots; i++) memcpy(roots[i].w, pr.commitments.data() + 32 * i, 32);
    const uint64_t g = gl::root_of_unity(pr.log_n);
    int log_N = 0;
    while (((size_t)1 << log_N) < N) log_N++;
    const uint64_t gN = gl::root_of_unity(log_N);

    // the AIR, when known: a program, or the built-in FibAir
    FibAir air;
    air::Instance pinst;
    if (prog) {
        if (W != prog->W || A != prog->A || pr.R != prog->R) reject("proof shape does not match the program's trace layout");
        if (pub.size() != prog->num_pub) reject("wrong number of public inputs for this program");
        if (C != prog->ce_blowup) reject("wrong number of composition columns for this program");
        try { pinst = air::instantiate(*prog, pr.log_n); } catch (const Error& e) { reject(e.what()); }
    }
    const bool have_air = air_desc != nullptr && !prog;
    if (have_air) {
        air.width = W; air.log_n = pr.log_n; air.results = pub;
        air.aux_width = air_desc->aux_width; air.aux_rands = air_desc->aux_width ? air_desc->aux_rands : 0;
        air.aux_degree = air_desc->aux_width ? air_desc->aux_degree : 2;
        if (W < 2 || (W & 1) || pub.size() != W / 2) reject("public inputs do not match the trace width");
        if (air.aux_width != A || air.aux_rands != pr.R) reject("proof shape does not match the AIR's auxiliary segment");
        if (A && (air.aux_degree < 2 || air.aux_degree > 8)) reject("auxiliary constraint degree must be in [2, 8]");
        if (C != air.ce_blowup_factor()) reject("wrong number of composition columns for this AIR");
    }

    // transcript (stark_verifier.cairo:83-144)
    HostCoin coin = HostCoin::from_elements(pub.data(), (uint32_t)pub.size());
    size_t ri = 0;
    coin.reseed(roots[ri++]);
    std::vector<T> rands;
    if (A) { for (uint32_t i = 0; i < pr.R; i++) rands.push_back(coin.draw<F>()); coin.reseed(roots[ri++]); }
    std::vector<T> ta, tb, ba, bb;
    if (have_air || prog) {
        const size_t nt = prog ? prog->num_transition() : air.num_transition_constraints(), na = prog ? prog->num_assertions() : air.num_assertions();
        for (size_t i = 0; i < nt; i++) { ta.push_back(coin.draw<F>()); tb.push_back(coin.draw<F>()); }
        for (size_t i = 0; i < na; i++) { ba.push_back(coin.draw<F>()); bb.push_back(coin.draw<F>()); }
    }
    const Digest croot = roots[ri++];
    coin.reseed(croot);
    const T z = coin.draw<F>();
    // OOD frame (stark_verifier.cairo:149-181)
    if (pr.ood_trace_states.size() != 2 * (size_t)TW * EB) reject("bad OOD frame length");
    std::vector<T> ood_cur(TW), ood_next(TW), ood_h(C);
    for (uint32_t i = 0; i < TW; i++) { ood_cur[i] = rd_elem<F>(pr.ood_trace_states, i); ood_next[i] = rd_elem<F>(pr.ood_trace_states, TW + i); }
    for (size_t i = 0; i < C; i++) ood_h[i] = rd_elem<F>(pr.ood_evaluations, i);
    coin.reseed(hash_elems<F>(ood_cur));
    coin.reseed(hash_elems<F>(ood_next));
    coin.reseed(hash_elems<F>(ood_h));
    if (prog) {
        // out-of-domain consistency for a program AIR: the compiled constraint program runs on the host over E, periodic columns
        // through their interpolants at z^(n / cycle), degree adjustments as plain powers of z
        std::vector<T> per;
        for (auto& cyc : prog->periodic) {
            std::vector<uint64_t> co = cyc;
            air::host_ntt(co, true);
            const T y = gl::fpow<F>(z, n / cyc.size());
            T acc = F::zero();
            for (size_t i = co.size(); i-- > 0;) acc = F::add(F::mul(acc, y), F::from(co[i]));
            per.push_back(acc);
        }
        const air::Scalars<F> sc = air::fold_scalars<F>(*prog, pub.data(), rands.data());
        const std::vector<T> num = air::host_evaluate<F>(*prog, pinst, sc, ood_cur.data(), ood_next.data(), per, ta, tb, ba, bb,
                                                         [&](uint64_t e) { return gl::fpow<F>(z, e); }, air::sequence_values_at<F>(*prog, pinst, z));
        T tdiv = F::inv(F::sub(gl::fpow<F>(z, n), F::one()));
        for (uint32_t i = 1; i <= prog->exemptions; i++) tdiv = F::mul(tdiv, F::sub(z, F::from(gl::pow(g, n - i))));
        T lhs = F::mul(num[0], tdiv);
        for (size_t j = 0; j < pinst.bgroups.size(); j++)
            lhs = F::add(lhs, F::mul(num[1 + j], F::inv(F::sub(gl::fpow<F>(z, pinst.bgroups[j].a), F::from(pinst.bgroups[j].b)))));
        T rhs = F::zero(), zp = F::one();
        for (size_t c = 0; c < C; c++) { rhs = F::add(rhs, F::mul(zp, ood_h[c])); zp = F::mul(zp, z); }
        if (!feq<F>(lhs, rhs)) reject("out-of-domain constraint evaluations are inconsistent");
    }
    if (have_air) {
        // sum over divisor groups of numerator(z) / divisor(z)  ==  sum_c z^c H_c(z^C)
        const uint64_t ce_n = (uint64_t)C * n;
        const T z_t = gl::fpow<F>(z, ce_n - 1), z_b = gl::fpow<F>(z, ce_n - n + 1);
        T acc = F::zero(), g0 = F::zero(), g1 = F::zero();
        for (uint32_t k = 0; k < W / 2; k++) {
            const T a = ood_cur[2 * k], b = ood_cur[2 * k + 1], na = ood_next[2 * k], nb = ood_next[2 * k + 1];
            acc = F::add(acc, F::mul(F::add(ta[2 * k], F::mul(tb[2 * k], z_t)), F::sub(na, F::add(a, b))));
            acc = F::add(acc, F::mul(F::add(ta[2 * k + 1], F::mul(tb[2 * k + 1], z_t)), F::sub(nb, F::add(b, na))));
            g0 = F::add(g0, F::mul(F::add(ba[2 * k], F::mul(bb[2 * k], z_b)), F::sub(a, F::from(1 + 2 * (uint64_t)k))));
            g0 = F::add(g0, F::mul(F::add(ba[2 * k + 1], F::mul(bb[2 * k + 1], z_b)), F::sub(b, F::from(2 + 2 * (uint64_t)k))));
            g1 = F::add(g1, F::mul(F::add(ba[W + k], F::mul(bb[W + k], z_b)), F::sub(b, F::from(pub[k]))));
        }
        if (A) {
            const uint32_t D = air.aux_degree;
            const T z_x = gl::fpow<F>(z, ce_n - 1 + (n - 1) - (uint64_t)D * (n - 1));
            for (uint32_t c = 0; c < A; c++) {
                const T f = gl::fpow<F>(F::add(rands[c % pr.R], ood_cur[c % W]), D - 1);
                const T t = F::sub(ood_next[W + c], F::mul(ood_cur[W + c], f));
                acc = F::add(acc, F::mul(F::add(ta[W + c], F::mul(tb[W + c], z_x)), t));
                const uint32_t bi = W + W / 2 + c;
                g0 = F::add(g0, F::mul(F::add(ba[bi], F::mul(bb[bi], z_b)), F::sub(ood_cur[W + c], F::one())));
            }
        }
        const T wl = F::from(gl::pow(g, n - 1));
        T lhs = F::mul(acc, F::mul(F::sub(z, wl), F::inv(F::sub(gl::fpow<F>(z, n), F::one()))));
        lhs = F::add(lhs, F::mul(g0, F::inv(F::sub(z, F::one()))));
        lhs = F::add(lhs, F::mul(g1, F::inv(F::sub(z, wl))));
        T rhs = F::zero(), zp = F::one();
        for (size_t c = 0; c < C; c++) { rhs = F::add(rhs, F::mul(zp, ood_h[c])); zp = F::mul(zp, z); }
        if (!feq<F>(lhs, rhs)) reject("out-of-domain constraint evaluations are inconsistent");
    }
    // DEEP coefficients, FRI alphas (air_instance.cairo:145-166, fri_verifier.cairo:56-82)
    std::vector<T> da(TW), db(TW), dg(TW), dc(C);
    for (uint32_t i = 0; i < TW; i++) { da[i] = coin.draw<F>(); db[i] = coin.draw<F>(); dg[i] = coin.draw<F>(); }
    for (size_t i = 0; i < C; i++) dc[i] = coin.draw<F>();
    const T lambda = coin.draw<F>(), mu = coin.draw<F>();
    std::vector<T> alphas;
    for (int l = 0; l <= layers; l++) { coin.reseed(roots[ri + l]); alphas.push_back(coin.draw<F>()); }
    // proof of work, query positions (stark_verifier.cairo:204-221)
    coin.reseed_with_int(pr.nonce);
    if (leading_zero_bits(coin.seed) < (int)pr.opt.grinding_factor) reject("insufficient proof of work");
    const std::vector<uint64_t> pos = coin.draw_integers(pr.opt.num_queries, N);
    const size_t Q = pos.size();

    // trace and constraint openings (channel.cairo:206-424); every path is authenticated
    std::vector<std::vector<T>> rows(Q, std::vector<T>(TW));
    {
        const uint32_t widths[2] = {W, A};
        for (size_t s = 0; s < pr.trace_queries.size(); s++) {
            const uint32_t w = widths[s];
            const size_t eb = s == 0 ? 8 : EB;               // main segment: base field; aux segment: E
            if (pr.trace_queries[s].values.size() != Q * w * eb) reject("bad trace query length");
            std::vector<Digest> leaves(Q);
            for (size_t i = 0; i < Q; i++) {
                std::vector<uint64_t> flat(w * eb / 8);
                for (size_t c = 0; c < flat.size(); c++) flat[c] = rd64(pr.trace_queries[s].values, i * flat.size() + c);
                leaves[i] = b2s::hash_elements(flat.data(), (uint32_t)flat.size());
                for (uint32_t c = 0; c < w; c++)
                    rows[i][s == 0 ? c : W + c] = s == 0 ? F::from(flat[c]) : F::make(flat[c * F::DEG], F::DEG > 1 ? flat[c * F::DEG + 1] : 0);
            }
            if (!same(batch_root(N, pos, leaves, pr.trace_queries[s].paths), roots[s])) reject("trace opening does not match its commitment");
        }
    }
    std::vector<std::vector<T>> crows(Q, std::vector<T>(C));
    {
        if (pr.constraint_queries.values.size() != Q * C * EB) reject("bad constraint query length");
        std::vector<Digest> leaves(Q);
        for (size_t i = 0; i < Q; i++) {
            for (size_t c = 0; c < C; c++) crows[i][c] = rd_elem<F>(pr.constraint_queries.values, i * C + c);
            leaves[i] = hash_elems<F>(crows[i]);
        }
        if (!same(batch_root(N, pos, leaves, pr.constraint_queries.paths), croot)) reject("constraint opening does not match its commitment");
    }
    // DEEP composition at the queried points (composer.cairo:48-316)
    const T z_next = F::mulb(z, g), z_c = gl::fpow<F>(z, C), z_conj = F::conj(z);
    std::vector<T> evals(Q);
    for (size_t i = 0; i < Q; i++) {
        const uint64_t x = gl::mul(gl::GEN, gl::pow(gN, pos[i]));
        const T xe = F::from(x);
        T s1 = F::zero(), s2 = F::zero(), s3 = F::zero();
        for (uint32_t c = 0; c < TW; c++) {
            const T v = rows[i][c];
            s1 = F::add(s1, F::mul(F::sub(v, ood_cur[c]), da[c]));
            s2 = F::add(s2, F::mul(F::sub(v, ood_next[c]), db[c]));
            if (F::DEG > 1 && c < W) s3 = F::add(s3, F::mul(F::sub(v, F::conj(ood_cur[c])), dg[c]));   // base-field columns only
        }
        T t = F::add(F::mul(s1, F::inv(F::sub(xe, z))), F::mul(s2, F::inv(F::sub(xe, z_next))));
        if (F::DEG > 1) t = F::add(t, F::mul(s3, F::inv(F::sub(xe, z_conj))));
        T sc = F::zero();
        for (size_t c = 0; c < C; c++) sc = F::add(sc, F::mul(F::sub(crows[i][c], ood_h[c]), dc[c]));
        t = F::add(t, F::mul(sc, F::inv(F::sub(xe, z_c))));
        evals[i] = F::mul(t, F::add(lambda, F::mulb(mu, x)));
    }
    // FRI (fri_verifier.cairo:243-451): the offset stays 7 at every layer
    if ((int)pr.fri_layers.size() != layers) reject("wrong number of FRI layers");
    std::vector<uint64_t> cur_pos = pos;
    std::vector<T> cur_eval = evals;
    uint64_t dom = N, omega = gN;
    const uint64_t gen_inv = gl::inv(gl::GEN), f_inv = gl::inv(Fd);
    for (int l = 0; l < layers; l++) {
        const uint64_t nrows = dom / Fd;
        const std::vector<uint64_t> fpos = fold_positions(cur_pos, dom, Fd);
        const QueriesBytes& q = pr.fri_layers[l];
        if (q.values.size() != fpos.size() * Fd * EB) reject("bad FRI layer length");
        std::vector<std::vector<T>> vals(fpos.size(), std::vector<T>(Fd));
        std::vector<Digest> leaves(fpos.size());
        for (size_t k = 0; k < fpos.size(); k++) {
            for (size_t j = 0; j < Fd; j++) vals[k][j] = rd_elem<F>(q.values, k * Fd + j);
            leaves[k] = hash_elems<F>(vals[k]);
        }
        if (!same(batch_root(nrows, fpos, leaves, q.paths), roots[ri + l])) reject("FRI layer opening does not match its commitment");
        for (size_t i = 0; i < cur_pos.size(); i++) {
            const uint64_t fp = cur_pos[i] % nrows, jj = cur_pos[i] / nrows;
            const size_t k = std::find(fpos.begin(), fpos.end(), fp) - fpos.begin();
            if (!feq<F>(vals[k][jj], cur_eval[i])) reject("FRI layer values are inconsistent with the previous layer");
        }
        // fold: value at row i of the next layer = interpolant through (x_i w_F^j, v_j) evaluated at alpha
        //       = (1/F) sum_k (alpha / x_i)^k sum_j v_j w_F^(-jk),  x_i = 7 w_dom^i
        const uint64_t wF_inv = gl::inv(gl::pow(omega, nrows));
        std::vector<T> nxt(fpos.size());
        for (size_t k = 0; k < fpos.size(); k++) {
            const uint64_t xinv = gl::mul(gen_inv, gl::inv(gl::pow(omega, fpos[k])));
            const T r = F::mulb(alphas[l], xinv);
            T rp = F::one(), acc = F::zero();
            for (size_t kk = 0; kk < Fd; kk++) {
                T ck = F::zero();
                for (size_t j = 0; j < Fd; j++) ck = F::add(ck, F::mulb(vals[k][j], gl::pow(wF_inv, j * kk)));
                acc = F::add(acc, F::mul(ck, rp));
                rp = F::mul(rp, r);
            }
            nxt[k] = F::mulb(acc, f_inv);
        }
        cur_pos = fpos; cur_eval = nxt; dom = nrows; omega = gl::pow(omega, Fd);
    }
    // remainder (channel.cairo:80-100, fri_verifier.cairo:261-265)
    if (pr.fri_remainder.size() != dom * EB) reject("bad remainder length");
    if (dom < Fd) reject("remainder smaller than the folding factor");
    std::vector<T> rem(dom);
    for (size_t i = 0; i < dom; i++) rem[i] = rd_elem<F>(pr.fri_remainder, i);
    for (size_t i = 0; i < cur_pos.size(); i++) if (!feq<F>(rem[cur_pos[i]], cur_eval[i])) reject("remainder values are inconsistent with the last layer");
    {
        const size_t nrows = dom / Fd;
        std::vector<Digest> leaves(nrows);
        for (size_t i = 0; i < nrows; i++) {
            std::vector<T> row(Fd);
            for (size_t j = 0; j < Fd; j++) row[j] = rem[i + j * nrows];
            leaves[i] = hash_elems<F>(row);
        }
        std::vector<Digest> lvl = leaves;
        while (lvl.size() > 1) {
            std::vector<Digest> up(lvl.size() / 2);
            for (size_t i = 0; i < up.size(); i++) up[i] = b2s::merge(lvl[2 * i], lvl[2 * i + 1]);
            lvl.swap(up);
        }
        if (!same(lvl[0], roots[ri + layers])) reject("remainder does not match its commitment");
    }
    {   // degree of the remainder polynomial: < max(1, n / F^layers)
        size_t bound = n;
        for (int l = 0; l < layers; l++) bound /= Fd;
        if (bound == 0) bound = 1;
        if (bound >= dom) reject("remainder degree bound is not below the remainder domain");
        for (int d = 0; d < F::DEG; d++) {
            std::vector<uint64_t> comp(dom);
            for (size_t i = 0; i < dom; i++) comp[i] = F::comp(rem[i], d);
            intt_host(comp);
            for (size_t i = bound; i < dom; i++) if (comp[i] != 0) reject("remainder polynomial degree is too high");
        }
    }
}

}  // namespace
}  // namespace aero

static int32_t verify_entry(const uint8_t* proof, size_t proof_len, const uint64_t* pub_elements, uint32_t n_pub, const aero_fib_air* air,
                            const aero::air::Program* prog, const aero_verify_policy* policy, char* err, size_t err_cap) {
    using namespace aero;
    auto put = [&](const std::string& s) { if (err && err_cap) { size_t k = std::min(err_cap - 1, s.size()); memcpy(err, s.data(), k); err[k] = 0; } };
    try {
        if (!proof || (!pub_elements && n_pub)) { put("verify: null argument"); return AERO_E_BAD_ARG; }
        aero_verify_policy pol{};
        if (policy) pol = *policy; else pol.min_query_security_bits = 96;
        if (!air && !prog && !pol.allow_unknown_air) {
            put("verify: the AIR descriptor is mandatory (without it the out-of-domain constraint check cannot run and any low-degree "
                "commitment would be accepted); set policy.allow_unknown_air to verify everything but that check");
            return AERO_E_BAD_ARG;
        }
        put("");
        const Parsed pr = parse(proof, proof_len);
        try { pr.opt.validate(); } catch (const Error& e) { reject(e.what()); }
        std::vector<uint64_t> pub(pub_elements, pub_elements + n_pub);
        for (uint64_t v : pub) if (v >= gl::P) reject("non-canonical public input");
        if (pr.opt.field_extension == EXT_NONE) verify_impl<gl::FB>(pr, pub, air, pol, prog);
        else verify_impl<gl::FQ>(pr, pub, air, pol, prog);
        return AERO_OK;
    } catch (const Error& e) {
        put(e.what());
        return e.code == AERO_E_VERIFY ? AERO_E_VERIFY : (e.code ? e.code : AERO_E_VERIFY);
    } catch (const std::exception& e) {
        put(std::string("verify: ") + e.what());
        return AERO_E_INTERNAL;
    }
}

namespace aero {
void self_verify_or_throw(const uint8_t* proof, size_t len, const std::vector<uint64_t>& pub, const aero_fib_air* air, const air::Program* prog,
                          uint32_t log_n, const aero_proof_options& opt) {
    aero_verify_policy pol{};
    pol.min_query_security_bits = 0;        // the options are the caller's choice; what is checked is that the proof is a proof under them
    pol.expected_log_n = log_n;
    pol.require_options = 1;
    pol.options = opt;
    char err[384];
    err[0] = 0;
    const int32_t rc = verify_entry(proof, len, pub.data(), (uint32_t)pub.size(), air, prog, &pol, err, sizeof err);
    if (rc != AERO_OK)
        throw Error(AERO_E_SELF_VERIFY, std::string("self-verify: the proof this call produced is rejected by the library's own verifier (") + err +
                                            "); no bytes are returned");
}
}  // namespace aero

extern "C" int32_t aero_verify_fib(const uint8_t* proof, size_t proof_len, const uint64_t* pub_elements, uint32_t n_pub, const aero_fib_air* air,
                                   const aero_verify_policy* policy, char* err, size_t err_cap) {
    return verify_entry(proof, proof_len, pub_elements, n_pub, air, nullptr, policy, err, err_cap);
}
extern "C" int32_t aero_verify_air(const uint8_t* proof, size_t proof_len, const uint64_t* pub, uint32_t n_pub, const aero_air* air,
                                   const aero_verify_policy* policy, char* err, size_t err_cap) {
    if (!air) { if (err && err_cap) snprintf(err, err_cap, "verify: null program"); return AERO_E_BAD_ARG; }
    return verify_entry(proof, proof_len, pub, n_pub, nullptr, &air->prog, policy, err, err_cap);
}

// num_queries * log2(blowup) + grinding, and the field-size term 64 * extension degree - log2(LDE domain) that caps any
// security claim over this field (Winterfell's conjectured-security formula takes the minimum of the two, minus one).
extern "C" int32_t aero_proof_security_bits(const uint8_t* proof, size_t proof_len, uint32_t* query_bits, uint32_t* field_bits) {
    using namespace aero;
    try {
        if (!proof) return AERO_E_BAD_ARG;
        const Parsed pr = parse(proof, proof_len);
        pr.opt.validate();
        int log_b = 0;
        while ((1u << log_b) < pr.opt.blowup_factor) log_b++;
        if (query_bits) *query_bits = (uint32_t)pr.opt.num_queries * (uint32_t)log_b + pr.opt.grinding_factor;
        if (field_bits) *field_bits = 64u * (uint32_t)pr.deg() - (uint32_t)(pr.log_n + log_b);
        return AERO_OK;
    } catch (const Error& e) {
        return e.code == AERO_E_VERIFY ? AERO_E_VERIFY : AERO_E_BAD_ARG;
    } catch (...) { return AERO_E_INTERNAL; }
}
