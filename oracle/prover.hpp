// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/gl.hpp header). Never linked into libaero_stark.so.
//
// CPU restatement of the prover side of the hot path for the built-in FibAir(W).
// Stage order and split points follow the reference's driver:
//   aero-sdk/miden-wasm/src/proving_worker.rs:238-278  stage 1: channel, domain, interpolate_columns, evaluate_columns_over
//   proving_worker.rs:280-310 + hashing_worker.rs:12-26  row hashing
//   proving_worker.rs:161-162                            MerkleTree::new
//   proving_worker.rs:323-332                            commit_to_trace_and_validate (reseed with root)
//   proving_worker.rs:355-439 + constraints_worker.rs:14-79  constraint evaluation (numerators per divisor)
//   proving_worker.rs:344-352                            prove_after_constraint_eval: composition poly -> commit ->
//                                                        OOD -> DEEP -> FRI -> grind -> queries -> StarkProof
// The bodies of those winter-prover 0.4 functions are not in the mount (empty submodule); the published
// algorithms are restated (see oracle/stark.hpp header for how they are pinned).
// Threading: OpenMP loops give the "all host cores" variant (per-column/per-coset NTTs, per-row hashing,
// per-row constraint evaluation — the same split as winter's `concurrent` feature); OMP_NUM_THREADS=1 or
// orc_set_threads(1) gives the single-threaded run the reference binary actually performs.
#pragma once
#include <chrono>
#include "stark.hpp"
#include "air.hpp"

namespace orc {

struct StageTimes {   // seconds, stage names after proving_worker.rs:125-172 console labels
    double interpolate = 0, lde = 0, trace_commit = 0, constraints = 0, composition = 0, comp_commit = 0,
           ood = 0, deep = 0, fri = 0, grind = 0, queries = 0, total = 0;
};
static inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Captured intermediates for per-stage parity tests against the HIP path.
template <class F> struct ProverArtifacts {
    std::vector<Col> trace_polys, trace_lde;          // W x n, W x N
    std::vector<Digest> trace_leaves;                 // N
    std::vector<std::vector<typename F::T>> ce_cols;  // 3 x (C*n) numerators
    std::vector<Col> comp_polys;                      // (C*DEG) x n  component columns
    std::vector<Col> comp_lde;                        // (C*DEG) x N
    std::vector<typename F::T> deep;                  // N
    std::vector<std::vector<typename F::T>> fri_layers;   // transposed evaluations per layer (incl. remainder layer)
    std::vector<typename F::T> ood_cur, ood_next, ood_h;
    Col cons_coeffs;                                  // constraint composition coefficients in draw order, DEG u64 each
    std::vector<Col> aux_cols, aux_lde;               // auxiliary segment: (A*DEG) x n trace columns, (A*DEG) x N
    Col aux_rands;                                    // the R drawn elements, DEG u64 each
};

// hash every row of a column-major matrix: leaf_j = hash_elements(row j)   [a5]
static std::vector<Digest> hash_rows(const std::vector<Col>& cols, size_t rows) {
    std::vector<Digest> out(rows);
    size_t w = cols.size();
#pragma omp parallel for schedule(static)
    for (size_t j = 0; j < rows; j++) {
        uint64_t buf[512];
        for (size_t c = 0; c < w; c++) buf[c] = cols[c][j];
        out[j] = hash_elements(buf, w);
    }
    return out;
}

// The prover, generic over the AIR model M (FibAir in oracle/stark.hpp, ProgramAir in oracle/air.hpp). `air` arrives complete:
// shape, trace length and the public inputs that seed the coin.
template <class F, class M>
static Bytes prove_model(const M& air, const std::vector<Col>& trace, int log_n, const Options& opt,
                         StageTimes* times = nullptr, ProverArtifacts<F>* art = nullptr) {
    typedef typename F::T T;
    const uint32_t W = (uint32_t)trace.size(), A = air.A, R = air.R;
    const size_t n = (size_t)1 << log_n, B = opt.blowup, N = n * B, Fd = opt.fri_fold;
    const size_t C = air.ce_blowup(), ceN = C * n, ce_step = B / C, NC = air.num_columns();
    if (W != air.W || log_n != air.log_n) throw Err("prove: trace shape does not match the AIR");
    if (B < C || (B & (B - 1))) throw Err("prove: blowup must be a power of two >= the constraint-evaluation blowup");
    if (Fd != 2 && Fd != 4 && Fd != 8 && Fd != 16) throw Err("prove: unsupported FRI folding factor");
    if (opt.hash_fn != HASH_BLAKE2S_256) throw Err("prove: only Blake2s_256 is supported");
    if (ilog2(N) > TWO_ADICITY) throw Err("prove: LDE domain exceeds field two-adicity");
    for (auto& c : trace) if (c.size() != n) throw Err("prove: ragged trace");
    if (A > 255 - W || (A && (R == 0 || R > 255))) throw Err("prove: bad auxiliary segment shape");
    StageTimes tm; double t0 = now_s(), t1;

    // 0. AIR + channel [proving_worker.rs:248-268]
    Coin coin = Coin::from_pub_elements(air.pub_elements().data(), air.pub_elements().size());
    const uint64_t g = gl_root_of_unity(log_n), gN = gl_root_of_unity(ilog2(N));
    Proof pr;
    pr.main_width = (uint8_t)W; pr.aux_width = (uint8_t)A; pr.aux_rands = (uint8_t)(A ? R : 0); pr.log_n = (uint8_t)log_n;
    pr.modulus.resize(8); for (int i = 0; i < 8; i++) pr.modulus[i] = (uint8_t)(P >> (8 * i));
    pr.opt = opt;

    // 1. interpolate_columns [a3, proving_worker.rs:273]
    std::vector<Col> polys = trace;
    for (uint32_t c = 0; c < W; c++) intt(polys[c].data(), n, true);
    t1 = now_s(); tm.interpolate = t1 - t0; t0 = t1;
    // 2. evaluate_columns_over [a4, proving_worker.rs:274]
    std::vector<Col> tlde(W);
    for (uint32_t c = 0; c < W; c++) tlde[c] = lde(polys[c].data(), n, B, GEN);
    t1 = now_s(); tm.lde = t1 - t0; t0 = t1;
    // 3. row hashes + Merkle tree, commit [a5, a6, a8]
    std::vector<Digest> tleaves = hash_rows(tlde, N);
    MerkleTree ttree(tleaves);
    wbytes(pr.commitments, Bytes(ttree.root().b, ttree.root().b + 32));
    coin.reseed(ttree.root());
    // 3b. auxiliary segment [a8; stark_verifier.cairo:266-294]: draw R elements, build the columns, commit like the main segment
    std::vector<T> rands;
    std::vector<Col> apolys(A * F::DEG), alde(A * F::DEG);    // component columns, index c*DEG + k
    std::vector<Digest> aleaves;
    MerkleTree atree;
    if (A) {
        for (uint32_t i = 0; i < R; i++) rands.push_back(coin.draw<F>());
        for (auto& c : apolys) c.resize(n);
        air.template build_aux<F>(trace, rands, apolys);
        if (art) { art->aux_cols = apolys; f_flatten<F>(rands.data(), rands.size(), art->aux_rands); }
        for (size_t c = 0; c < A * F::DEG; c++) { intt(apolys[c].data(), n, true); alde[c] = lde(apolys[c].data(), n, B, GEN); }
        if (art) art->aux_lde = alde;
        aleaves = hash_rows(alde, N);
        atree = MerkleTree(aleaves);
        wbytes(pr.commitments, Bytes(atree.root().b, atree.root().b + 32));
        coin.reseed(atree.root());
    }
    t1 = now_s(); tm.trace_commit = t1 - t0; t0 = t1;

    // 4. constraint evaluation over the ce domain [a9, a10]
    auto cc = draw_constraint_coeffs<F>(coin, air.num_transition(), air.num_assertions());
    std::vector<std::vector<T>> ce(NC, std::vector<T>(ceN));
    const uint64_t gce = gl_root_of_unity(ilog2(ceN));
#pragma omp parallel
    {
        std::vector<uint64_t> cur(W), nxt(W);
        std::vector<T> acur(A), anxt(A), o(NC);
#pragma omp for schedule(static)
        for (size_t s = 0; s < ceN; s++) {
            size_t r = s * ce_step, rn = (r + B) % N;     // frame = (row j, row j + blowup mod N)
            for (uint32_t c = 0; c < W; c++) { cur[c] = tlde[c][r]; nxt[c] = tlde[c][rn]; }
            for (uint32_t c = 0; c < A; c++) {
                uint64_t a0[2] = {alde[c * F::DEG][r], F::DEG > 1 ? alde[c * F::DEG + F::DEG - 1][r] : 0};
                uint64_t a1[2] = {alde[c * F::DEG][rn], F::DEG > 1 ? alde[c * F::DEG + F::DEG - 1][rn] : 0};
                acur[c] = F::make(a0); anxt[c] = F::make(a1);
            }
            uint64_t x = gl_mul(GEN, gl_pow(gce, s));
            air.template eval_row<F>(cc, cur.data(), nxt.data(), acur.data(), anxt.data(), rands.data(), x, o.data());
            for (size_t k = 0; k < NC; k++) ce[k][s] = o[k];
        }
    }
    t1 = now_s(); tm.constraints = t1 - t0; t0 = t1;

    // 5. composition polynomial [a11]: divide by divisors, sum, interpolate over the coset, split into C columns
    std::vector<Col> hcomp(F::DEG, Col(ceN));
    air.template divide<F>(ce, hcomp);
    for (int k = 0; k < F::DEG; k++) intt_coset(hcomp[k].data(), ceN, GEN, true);
    // split: coefficient i -> column i mod C (H(x) = sum_c x^c H_c(x^C); stark_verifier.cairo:166-176)
    std::vector<Col> cpolys(C * F::DEG, Col(n));   // index c*DEG + k
    for (size_t c = 0; c < C; c++) for (int k = 0; k < F::DEG; k++) {
        Col& dst = cpolys[c * F::DEG + k];
        const Col& src = hcomp[k];
#pragma omp parallel for schedule(static) if (n >= 65536)
        for (size_t i = 0; i < n; i++) dst[i] = src[i * C + c];
    }
    t1 = now_s(); tm.composition = t1 - t0; t0 = t1;
    // 6. composition commitment [a12]
    std::vector<Col> clde(C * F::DEG);
    for (size_t c = 0; c < C * F::DEG; c++) clde[c] = lde(cpolys[c].data(), n, B, GEN);
    std::vector<Digest> cleaves = hash_rows(clde, N);
    MerkleTree ctree(cleaves);
    wbytes(pr.commitments, Bytes(ctree.root().b, ctree.root().b + 32));
    coin.reseed(ctree.root());
    t1 = now_s(); tm.comp_commit = t1 - t0; t0 = t1;

    // 7. OOD frame [a13]
    T z = coin.draw<F>();
    T z_next = F::mulb(z, g), zC = f_pow<F>(z, C);
    const size_t TW = W + A;
    std::vector<T> ood_cur(TW), ood_next(TW), ood_h(C);
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t c = 0; c < W; c++) { ood_cur[c] = horner<F>(polys[c].data(), n, z); ood_next[c] = horner<F>(polys[c].data(), n, z_next); }
    for (uint32_t c = 0; c < A; c++) {   // aux column polynomials have E-valued coefficients: recombine the components
        T a0 = F::zero(), a1 = F::zero();
        for (int k = 0; k < F::DEG; k++) {
            uint64_t basis[2] = {k == 0 ? 1ULL : 0ULL, k == 1 ? 1ULL : 0ULL};
            a0 = F::add(a0, F::mul(horner<F>(apolys[c * F::DEG + k].data(), n, z), F::make(basis)));
            a1 = F::add(a1, F::mul(horner<F>(apolys[c * F::DEG + k].data(), n, z_next), F::make(basis)));
        }
        ood_cur[W + c] = a0; ood_next[W + c] = a1;
    }
    for (size_t c = 0; c < C; c++) {
        // H_c has E-valued coefficients: evaluate component polynomials and recombine (a0 + a1*phi)
        T acc = F::zero();
        for (int k = 0; k < F::DEG; k++) {
            T e = horner<F>(cpolys[c * F::DEG + k].data(), n, zC);
            uint64_t basis[2] = {k == 0 ? 1ULL : 0ULL, k == 1 ? 1ULL : 0ULL};
            acc = F::add(acc, F::mul(e, F::make(basis)));
        }
        ood_h[c] = acc;
    }
    {
        Col flat; f_flatten<F>(ood_cur.data(), TW, flat); f_flatten<F>(ood_next.data(), TW, flat);
        for (uint64_t v : flat) w64(pr.ood_trace_states, v);
        Col fh; f_flatten<F>(ood_h.data(), C, fh);
        for (uint64_t v : fh) w64(pr.ood_evaluations, v);
    }
    coin.reseed(f_hash<F>(ood_cur.data(), TW));
    coin.reseed(f_hash<F>(ood_next.data(), TW));
    coin.reseed(f_hash<F>(ood_h.data(), C));
    t1 = now_s(); tm.ood = t1 - t0; t0 = t1;

    // 8. DEEP composition over the LDE domain [a14]
    std::vector<T> da(TW), db(TW), dg(TW), dc(C);
    for (uint32_t i = 0; i < TW; i++) { da[i] = coin.draw<F>(); db[i] = coin.draw<F>(); dg[i] = coin.draw<F>(); }
    for (size_t i = 0; i < C; i++) dc[i] = coin.draw<F>();
    T lambda = coin.draw<F>(), mu = coin.draw<F>();
    T z_conj = F::conj(z);
    std::vector<T> deep(N);
    {
        const size_t CH = 1024;
        const int ND = F::DEG > 1 ? 4 : 3;
#pragma omp parallel for schedule(static)
        for (size_t c0 = 0; c0 < N; c0 += CH) {
            size_t m = std::min(CH, N - c0);
            std::vector<T> d(ND * m), pre(ND * m);
            Col xs(m);
            uint64_t x = gl_mul(GEN, gl_pow(gN, c0));
            for (size_t i = 0; i < m; i++) {
                xs[i] = x; T xe = F::from(x);
                d[ND * i] = F::sub(xe, z); d[ND * i + 1] = F::sub(xe, z_next); d[ND * i + 2] = F::sub(xe, zC);
                if (ND == 4) d[ND * i + 3] = F::sub(xe, z_conj);
                x = gl_mul(x, gN);
            }
            T acc = F::one();
            for (size_t i = 0; i < ND * m; i++) { pre[i] = acc; acc = F::mul(acc, d[i]); }
            T ia = F::inv(acc);
            for (size_t i = ND * m; i-- > 0;) { T inv = F::mul(ia, pre[i]); ia = F::mul(ia, d[i]); d[i] = inv; }
            for (size_t i = 0; i < m; i++) {
                size_t r = c0 + i;
                T s1 = F::zero(), s2 = F::zero(), s3 = F::zero();
                for (uint32_t c = 0; c < W; c++) {
                    T v = F::from(tlde[c][r]);
                    s1 = F::add(s1, F::mul(F::sub(v, ood_cur[c]), da[c]));
                    s2 = F::add(s2, F::mul(F::sub(v, ood_next[c]), db[c]));
                    if (F::DEG > 1) s3 = F::add(s3, F::mul(F::sub(v, F::conj(ood_cur[c])), dg[c]));
                }
                for (uint32_t c = 0; c < A; c++) {   // aux columns live in E: no conjugate term (composer.cairo:196-316)
                    uint64_t comp[2] = {alde[c * F::DEG][r], F::DEG > 1 ? alde[c * F::DEG + F::DEG - 1][r] : 0};
                    T v = F::make(comp);
                    s1 = F::add(s1, F::mul(F::sub(v, ood_cur[W + c]), da[W + c]));
                    s2 = F::add(s2, F::mul(F::sub(v, ood_next[W + c]), db[W + c]));
                }
                T t = F::add(F::mul(s1, d[ND * i]), F::mul(s2, d[ND * i + 1]));
                if (F::DEG > 1) t = F::add(t, F::mul(s3, d[ND * i + 3]));
                T sc = F::zero();
                for (size_t c = 0; c < C; c++) {
                    uint64_t comp[2] = {clde[c * F::DEG][r], F::DEG > 1 ? clde[c * F::DEG + F::DEG - 1][r] : 0};
                    sc = F::add(sc, F::mul(F::sub(F::make(comp), ood_h[c]), dc[c]));
                }
                t = F::add(t, F::mul(sc, d[ND * i + 2]));
                deep[r] = F::mul(t, F::add(lambda, F::mulb(mu, xs[i])));
            }
        }
    }
    t1 = now_s(); tm.deep = t1 - t0; t0 = t1;

    // 9. FRI commit phase [a15]: num_fri_layers + 1 rounds (the last one commits the remainder)
    const int layers = num_fri_layers(N, Fd, (uint64_t)1 << opt.log_max_remainder);
    { uint64_t rem = N; for (int l = 0; l < layers; l++) rem /= Fd; if (rem < Fd) throw Err("prove: FRI remainder smaller than the folding factor"); }
    struct Layer { MerkleTree tree; std::vector<T> rows; size_t nrows; bool has_tree; Digest root; };
    std::vector<Layer> fl(layers + 1);
    {
        std::vector<T> ev = deep;
        uint64_t dom = N, omega = gN;
        // inverse DFT matrix of size Fd: winv[j*k] = w_F^(-jk)
        for (int l = 0; l <= layers; l++) {
            size_t rows = dom / Fd;
            Layer& L = fl[l];
            L.nrows = rows;
            L.rows.resize(dom);
            std::vector<Digest> leaves(rows);
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < rows; i++) {
                T row[16];
                for (size_t j = 0; j < Fd; j++) { row[j] = ev[i + j * rows]; L.rows[i * Fd + j] = row[j]; }
                leaves[i] = f_hash<F>(row, Fd);
            }
            if (rows >= 2) { L.tree = MerkleTree(leaves); L.root = L.tree.root(); L.has_tree = true; }
            else { L.root = leaves[0]; L.has_tree = false; }
            wbytes(pr.commitments, Bytes(L.root.b, L.root.b + 32));
            coin.reseed(L.root);
            T alpha = coin.draw<F>();
            if (l == layers) break;   // the alpha drawn after the remainder commitment is unused
            // fold: p_i = interpolant through (7 w^i w_F^j, row_i[j]); next[i] = p_i(alpha); offset stays 7
            const uint64_t wF = gl_pow(omega, rows), wFi = gl_inv(wF), Finv = gl_inv(Fd), oinv = gl_inv(omega);
            Col tw(Fd * Fd);
            for (size_t j = 0; j < Fd; j++) for (size_t k = 0; k < Fd; k++) tw[j * Fd + k] = gl_pow(wFi, j * k);
            std::vector<T> nxt(rows);
            const size_t CH = 1024;
#pragma omp parallel for schedule(static)
            for (size_t c0 = 0; c0 < rows; c0 += CH) {
                size_t m = std::min(CH, rows - c0);
                uint64_t xinv = gl_mul(gl_inv(GEN), gl_pow(oinv, c0));
                for (size_t i = 0; i < m; i++) {
                    const T* row = &L.rows[(c0 + i) * Fd];
                    T r = F::mulb(alpha, xinv), rp = F::one(), acc = F::zero();
                    for (size_t k = 0; k < Fd; k++) {
                        T ck = F::zero();
                        for (size_t j = 0; j < Fd; j++) ck = F::add(ck, F::mulb(row[j], tw[j * Fd + k]));
                        acc = F::add(acc, F::mul(ck, rp));
                        rp = F::mul(rp, r);
                    }
                    nxt[c0 + i] = F::mulb(acc, Finv);
                    xinv = gl_mul(xinv, oinv);
                }
            }
            ev.swap(nxt); dom = rows; omega = gl_pow(omega, Fd);
        }
    }
    t1 = now_s(); tm.fri = t1 - t0; t0 = t1;

    // 10. grinding [a16]: smallest nonce >= 1 with enough leading zeros (winter-prover 0.4 scans 1..u64::MAX)
    {
        uint64_t nonce = 0;
        const uint64_t CH = 1 << 14;
        for (uint64_t base = 1; nonce == 0; base += CH) {
            uint64_t best = UINT64_MAX;
#pragma omp parallel for schedule(static) reduction(min : best)
            for (uint64_t v = base; v < base + CH; v++)
                if (coin.check_leading_zeros(v) >= opt.grinding && v < best) best = v;
            if (best != UINT64_MAX) nonce = best;
        }
        pr.pow_nonce = nonce;
        coin.reseed_int(nonce);
    }
    t1 = now_s(); tm.grind = t1 - t0; t0 = t1;

    // 11. queries [a17]
    std::vector<uint64_t> pos = coin.draw_integers(opt.num_queries, N);
    {
        Proof::Q q;
        for (uint64_t p : pos) for (uint32_t c = 0; c < W; c++) w64(q.values, tlde[c][p]);
        q.paths = batch_serialize(batch_prove(ttree, pos));
        pr.trace_queries.push_back(q);
    }
    if (A) {
        Proof::Q q;
        for (uint64_t p : pos) for (size_t c = 0; c < A * F::DEG; c++) w64(q.values, alde[c][p]);
        q.paths = batch_serialize(batch_prove(atree, pos));
        pr.trace_queries.push_back(q);
    }
    {
        Proof::Q& q = pr.constraint_queries;
        for (uint64_t p : pos) for (size_t c = 0; c < C * F::DEG; c++) w64(q.values, clde[c][p]);
        q.paths = batch_serialize(batch_prove(ctree, pos));
    }
    {
        std::vector<uint64_t> fp = pos;
        uint64_t dom = N;
        for (int l = 0; l < layers; l++) {
            fp = fold_positions(fp, dom, Fd);
            Proof::Q q;
            for (uint64_t p : fp) { Col flat; f_flatten<F>(&fl[l].rows[p * Fd], Fd, flat); for (uint64_t v : flat) w64(q.values, v); }
            q.paths = batch_serialize(batch_prove(fl[l].tree, fp));
            pr.fri_layers.push_back(q);
            dom /= Fd;
        }
        // remainder: un-transpose the last layer (winter-fri 0.4 build_proof)
        const Layer& L = fl[layers];
        size_t rows = L.nrows;
        std::vector<T> rem(rows * Fd);
        for (size_t i = 0; i < rows; i++) for (size_t j = 0; j < Fd; j++) rem[i + rows * j] = L.rows[i * Fd + j];
        Col flat; f_flatten<F>(rem.data(), rem.size(), flat);
        for (uint64_t v : flat) w64(pr.fri_remainder, v);
        pr.fri_log_partitions = 0;
    }
    t1 = now_s(); tm.queries = t1 - t0;
    tm.total = tm.interpolate + tm.lde + tm.trace_commit + tm.constraints + tm.composition + tm.comp_commit + tm.ood +
               tm.deep + tm.fri + tm.grind + tm.queries;
    if (times) *times = tm;
    if (art) {
        art->trace_polys = polys; art->trace_lde = tlde; art->trace_leaves = tleaves; art->ce_cols = ce;
        art->comp_polys = cpolys; art->comp_lde = clde; art->deep = deep;
        for (auto& L : fl) art->fri_layers.push_back(L.rows);
        art->ood_cur = ood_cur; art->ood_next = ood_next; art->ood_h = ood_h;
        for (size_t i = 0; i < cc.ta.size(); i++) { f_flatten<F>(&cc.ta[i], 1, art->cons_coeffs); f_flatten<F>(&cc.tb[i], 1, art->cons_coeffs); }
        for (size_t i = 0; i < cc.ba.size(); i++) { f_flatten<F>(&cc.ba[i], 1, art->cons_coeffs); f_flatten<F>(&cc.bb[i], 1, art->cons_coeffs); }
    }
    return pr.to_bytes();
}

// FibAir(W) [+ auxiliary segment (A, R, D)] on a given trace: the public inputs are read off the trace's last row.
template <class F>
static Bytes prove_fib(const std::vector<Col>& trace, int log_n, const Options& opt, Col* pub_out,
                       StageTimes* times = nullptr, ProverArtifacts<F>* art = nullptr, uint32_t A = 0, uint32_t R = 0, uint32_t D = 2) {
    const uint32_t W = (uint32_t)trace.size();
    const size_t n = (size_t)1 << log_n;
    if (D < 2 || D > 8) throw Err("prove: aux constraint degree must be in [2, 8]");
    if (W < 2 || (W & 1) || W > 254) throw Err("prove: FibAir needs an even column count in [2, 254]");
    for (auto& c : trace) if (c.size() != n) throw Err("prove: ragged trace");
    FibAir air; air.W = W; air.log_n = log_n; air.A = A; air.R = A ? R : 0; air.D = D;
    for (uint32_t k = 0; k < W / 2; k++) air.results.push_back(trace[2 * k + 1][n - 1]);
    if (pub_out) *pub_out = air.results;
    return prove_model<F, FibAir>(air, trace, log_n, opt, times, art);
}

static Bytes prove_fib_any(const std::vector<Col>& trace, int log_n, const Options& opt, Col* pub_out, StageTimes* times = nullptr,
                           uint32_t A = 0, uint32_t R = 0, uint32_t D = 2) {
    if (opt.field_ext == EXT_NONE) return prove_fib<FB>(trace, log_n, opt, pub_out, times, nullptr, A, R, D);
    if (opt.field_ext == EXT_QUADRATIC) return prove_fib<FQ>(trace, log_n, opt, pub_out, times, nullptr, A, R, D);
    throw Err("prove: unsupported field extension");
}

}  // namespace orc
