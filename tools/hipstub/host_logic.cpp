// Host-logic exercise of libaero_stark against the stand-in HIP runtime (hipstub.cpp), built with -fsanitize=thread or =address:
// handle lifetimes from several threads, the pool's worker threads, the local group's rendezvous + event protocol with real byte
// movement (copies run, kernels do not), error paths. TEST INFRASTRUCTURE (tools/hipstub/run.sh); exits non-zero on a wrong byte.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/aero_air.h"
#include "../../include/aero_air_builder.hpp"
#include "../../include/aero_stark.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s (line %d)\n", #cond, __LINE__); exit(1); } } while (0)
static const uint64_t P = 0xFFFFFFFF00000001ull;

static void fill(std::vector<uint64_t>& v, uint64_t seed) { for (size_t i = 0; i < v.size(); i++) v[i] = (seed * 0x9E3779B97F4A7C15ull + i * 1315423911ull) % P; }

static void lifetimes(int tid) {
    for (int it = 0; it < 20; it++) {
        aero_ctx* ctx = nullptr;
        CHECK(aero_ctx_create(0, &ctx) == AERO_OK);
        const uint32_t w = 1 + (it % 4), log_n = 3 + (it % 6);
        std::vector<uint64_t> host((size_t)w << log_n), back(host.size());
        fill(host, tid * 1000 + it);
        aero_matrix *m = nullptr, *m2 = nullptr;
        CHECK(aero_trace_upload(ctx, host.data(), w, log_n, &m) == AERO_OK);
        CHECK(aero_trace_upload(ctx, host.data(), w, log_n, &m2) == AERO_OK);
        CHECK(aero_matrix_download(ctx, m, back.data()) == AERO_OK);
        CHECK(back == host);
        aero_matrix_free(ctx, m);
        if (it & 1) { aero_ctx_destroy(ctx); aero_matrix_free(nullptr, m2); }      // the context handle goes first: the matrix keeps the Context alive
        else { aero_matrix_free(ctx, m2); aero_ctx_destroy(ctx); }
        // bad arguments end in status codes, not in exceptions or crashes
        aero_ctx* c2 = nullptr;
        CHECK(aero_ctx_create(0, &c2) == AERO_OK);
        CHECK(aero_trace_upload(c2, nullptr, 2, 4, &m) != AERO_OK);
        CHECK(aero_trace_upload(c2, host.data(), 0, 4, &m) != AERO_OK);
        CHECK(aero_trace_upload(nullptr, host.data(), 2, 4, &m) != AERO_OK);
        CHECK(aero_last_error(c2)[0] != 0);
        aero_ctx_destroy(c2);
    }
}

// the stage entry points: parameter blocks through the pinned staging ring (which wraps), table caches, handle creation on every
// path - the kernels are skipped, so only the host side (sizes, copies, lifetimes) is what runs; results are not inspected
static void stages(int tid) {
    aero_ctx* ctx = nullptr;
    CHECK(aero_ctx_create(0, &ctx) == AERO_OK);
    for (int it = 0; it < 120; it++) {
        const uint32_t w = 2 * (1 + it % 3), log_n = 4 + (it % 7);
        const size_t n = (size_t)1 << log_n;
        std::vector<uint64_t> host(w * n);
        fill(host, tid * 31 + it);
        aero_matrix *m = nullptr, *polys = nullptr, *lde = nullptr;
        CHECK(aero_trace_upload(ctx, host.data(), w, log_n, &m) == AERO_OK);
        CHECK(aero_interpolate_columns(ctx, m, &polys) == AERO_OK);
        CHECK(aero_evaluate_columns_over(ctx, polys, 3, &lde) == AERO_OK);
        std::vector<uint64_t> ev(w);
        CHECK(aero_poly_eval(ctx, polys, 12345, ev.data()) == AERO_OK);
        std::vector<uint8_t> digests(8 * n * 32), root(32);
        CHECK(aero_hash_matrix_rows(ctx, lde, digests.data()) == AERO_OK);
        aero_tree* tree = nullptr;
        CHECK(aero_merkle_commit_rows(ctx, lde, &tree, root.data()) == AERO_OK);
        const uint64_t pos[3] = {1, 5, (8 * n) - 1};
        std::vector<uint8_t> open(1 + 3 * (1 + 32 * 40));
        size_t open_len = 0;
        CHECK(aero_merkle_open_batch(ctx, tree, pos, 3, open.data(), open.size(), &open_len) == AERO_OK);
        CHECK(open_len <= open.size());
        aero_tree_free(ctx, tree);
        std::vector<uint64_t> rows(n * 3), hd(n * 32 / 8);
        CHECK(aero_hash_rows(ctx, rows.data(), 3, n, reinterpret_cast<uint8_t*>(hd.data())) == AERO_OK);
        std::vector<uint64_t> folded(8 * n / 4);
        std::vector<uint64_t> vals(8 * n);
        CHECK(aero_fri_fold(ctx, vals.data(), 8 * n, 4, 7, folded.data()) == AERO_OK);
        aero_matrix_free(ctx, lde); aero_matrix_free(ctx, polys); aero_matrix_free(ctx, m);
        CHECK(aero_merkle_from_leaves(ctx, digests.data(), 3, &tree, nullptr) != AERO_OK);      // not a power of two
        CHECK(aero_evaluate_columns_over(ctx, nullptr, 3, &lde) != AERO_OK);
    }
    aero_ctx_destroy(ctx);
}

// whole proofs: with kernels that do nothing the device transcript cannot agree with the host's, so a proof ends in the prover's
// consistency check - after the whole host pipeline up to the FRI layers has run (staging, parameter packs, handle and scratch
// lifetimes, the run-time compiled kernel's module path) and through the error path that drains the streams and resets the scratch list
static void proofs(int tid) {
    aero_ctx* ctx = nullptr;
    CHECK(aero_ctx_create(0, &ctx) == AERO_OK);
    for (int it = 0; it < 6; it++) {
        const uint32_t w = 2 + 2 * (it % 2), log_n = 4 + 2 * (it % 3);
        std::vector<uint64_t> t((size_t)w << log_n), pub(w);
        CHECK(aero_fib_trace(w, log_n, t.data()) == AERO_OK);
        const aero_proof_options opt = {8, 8, 0, 4, (uint8_t)(1 + it % 2), 4, 4};       // no grinding: its search would never end without a kernel
        uint8_t* proof = nullptr;
        size_t len = 0;
        const aero_fib_air aux = {2, 3, 2};
        const int32_t rc = (it & 1) ? aero_prove_fib_air_host(ctx, t.data(), w, log_n, &aux, &opt, &proof, &len, pub.data())
                                    : aero_prove_fib_host(ctx, t.data(), w, log_n, &opt, &proof, &len, pub.data());
        CHECK(rc == AERO_OK || aero_last_error(ctx)[0] != 0);
        aero_free(proof);
        // the same through a constraint program (the AIR-as-data path: pool of scalars, code generation + module load against the stand-in)
        uint8_t* program = nullptr;
        size_t plen = 0;
        CHECK(aero_air_fib_program(w, (it & 1) ? &aux : nullptr, &program, &plen) == AERO_OK);
        aero_air* air = nullptr;
        char err[256];
        CHECK(aero_air_load(program, plen, &air, err, sizeof err) == AERO_OK);
        aero_free(program);
        proof = nullptr;
        const int32_t rc2 = aero_prove_air_host(ctx, air, t.data(), log_n, pub.data(), w / 2, &opt, &proof, &len);
        CHECK(rc2 == AERO_OK || aero_last_error(ctx)[0] != 0);
        aero_free(proof);
        aero_air_free(air);
        (void)tid;
    }
    aero_ctx_destroy(ctx);
}

static void pools() {
    for (int it = 0; it < 6; it++) {
        aero_pool* pool = nullptr;
        CHECK(aero_pool_create(0, 1 + it % 4, &pool) == AERO_OK);
        CHECK(aero_pool_slots(pool) == (uint32_t)(1 + it % 4));
        std::vector<uint64_t> host(2 << 6);
        fill(host, it);
        aero_matrix* m = nullptr;
        CHECK(aero_trace_upload(aero_pool_ctx(pool, 0), host.data(), 2, 6, &m) == AERO_OK);
        if (it & 1) { aero_pool_destroy(pool); aero_matrix_free(nullptr, m); }      // a matrix may outlive its pool (copy gate)
        else { aero_matrix_free(aero_pool_ctx(pool, 0), m); aero_pool_destroy(pool); }
    }
}

struct RankData { aero_ctx* ctx = nullptr; aero_comm comm{}; };

static void local_group(uint32_t world, int rounds, bool abort_midway) {
    aero_local_group* g = nullptr;
    CHECK(aero_local_group_create(world, &g) == AERO_OK);
    std::vector<RankData> rd(world);
    std::atomic<int> failures{0}, aborted_seen{0};
    std::vector<std::thread> th;
    for (uint32_t r = 0; r < world; r++)
        th.emplace_back([&, r] {
            CHECK(aero_ctx_create(0, &rd[r].ctx) == AERO_OK);
            CHECK(aero_local_group_comm(g, rd[r].ctx, (int32_t)r, 0, &rd[r].comm) == AERO_OK);
            aero_comm& c = rd[r].comm;
            for (int it = 0; it < rounds; it++) {
                const uint32_t log_n = 4 + (it % 5);
                const size_t n = (size_t)1 << log_n, per = n / world ? n / world : 1;      // elements per peer chunk
                std::vector<uint64_t> send(n), zero(n * world, 0), got(n * world);
                fill(send, r * 7919 + it);
                aero_matrix *ms = nullptr, *mr = nullptr;
                CHECK(aero_trace_upload(rd[r].ctx, send.data(), 1, log_n, &ms) == AERO_OK);
                CHECK(aero_trace_upload(rd[r].ctx, zero.data(), world, log_n, &mr) == AERO_OK);
                uint64_t *ds = nullptr, *dr = nullptr;
                aero_matrix_device_ptr(ms, &ds); aero_matrix_device_ptr(mr, &dr);
                if (abort_midway && it == rounds / 2 && r == 1) { aero_local_group_abort(g); aero_matrix_free(rd[r].ctx, ms); aero_matrix_free(rd[r].ctx, mr); break; }
                // all_gather: every rank's whole vector
                int rc = c.all_gather(c.user, ds, dr, n * 8);
                if (rc != 0) { aborted_seen++; aero_matrix_free(rd[r].ctx, ms); aero_matrix_free(rd[r].ctx, mr); break; }
                CHECK(aero_ctx_synchronize(rd[r].ctx) == AERO_OK);
                CHECK(aero_matrix_download(rd[r].ctx, mr, got.data()) == AERO_OK);
                for (uint32_t p = 0; p < world; p++) {
                    std::vector<uint64_t> want(n);
                    fill(want, p * 7919 + it);
                    if (memcmp(got.data() + (size_t)p * n, want.data(), n * 8) != 0) failures++;
                }
                // all_to_all: chunk r of every peer
                rc = c.all_to_all(c.user, ds, dr, per * 8);
                if (rc != 0) { aborted_seen++; aero_matrix_free(rd[r].ctx, ms); aero_matrix_free(rd[r].ctx, mr); break; }
                CHECK(aero_matrix_download(rd[r].ctx, mr, got.data()) == AERO_OK);       // download waits on the rank's own stream
                for (uint32_t p = 0; p < world && per * world <= n; p++) {
                    std::vector<uint64_t> want(n);
                    fill(want, p * 7919 + it);
                    if (memcmp(got.data() + (size_t)p * per, want.data() + (size_t)r * per, per * 8) != 0) failures++;
                }
                // pairwise shift, then the all-reduce (its sum is a kernel: only the protocol runs here)
                if (c.send_recv) {
                    rc = c.send_recv(c.user, ds, (int32_t)((r + 1) % world), dr, (int32_t)((r + world - 1) % world), n * 8);
                    if (rc != 0) { aborted_seen++; aero_matrix_free(rd[r].ctx, ms); aero_matrix_free(rd[r].ctx, mr); break; }
                    CHECK(aero_matrix_download(rd[r].ctx, mr, got.data()) == AERO_OK);
                    std::vector<uint64_t> want(n);
                    fill(want, ((r + world - 1) % world) * 7919 + it);
                    if (memcmp(got.data(), want.data(), n * 8) != 0) failures++;
                }
                rc = c.all_reduce_sum_u64(c.user, dr, n);
                if (rc != 0) { aborted_seen++; aero_matrix_free(rd[r].ctx, ms); aero_matrix_free(rd[r].ctx, mr); break; }
                CHECK(aero_ctx_synchronize(rd[r].ctx) == AERO_OK);
                aero_matrix_free(rd[r].ctx, ms); aero_matrix_free(rd[r].ctx, mr);
            }
        });
    for (auto& t : th) t.join();
    CHECK(failures.load() == 0);
    if (abort_midway) CHECK(aborted_seen.load() >= 1);
    uint64_t st[4];
    CHECK(aero_local_group_stats(g, 0, st) == AERO_OK);
    aero_local_group_destroy(g);
    for (auto& d : rd) aero_ctx_destroy(d.ctx);
}

// one proof over thread ranks inside the library (aero_prove_fib_sharded_local): contexts, rank threads, host hand-over, every exchange
// of the sharded pipeline up to the point where the kernel-less transcript is found out - all ranks must come back, none may hang
static void sharded_proofs() {
    for (uint32_t world : {2u, 4u}) {
        const uint32_t w = 4, log_n = 7;
        std::vector<uint64_t> t((size_t)w << log_n), pub(w);
        CHECK(aero_fib_trace(w, log_n, t.data()) == AERO_OK);
        const aero_proof_options opt = {8, 8, 0, 4, 1, 4, 4};
        std::vector<int32_t> devs(world, 0);
        std::vector<uint8_t*> proofs(world, nullptr);
        std::vector<size_t> lens(world, 0);
        std::vector<double> ms(world, 0);
        std::vector<uint64_t> sent(world, 0);
        char err[1024] = {0};
        const int32_t rc = aero_prove_fib_sharded_local(devs.data(), world, t.data(), w, log_n, nullptr, &opt, 0, proofs.data(), lens.data(), pub.data(), ms.data(),
                                                        sent.data(), err, sizeof err);
        CHECK(rc == AERO_OK || err[0] != 0);
        for (uint8_t* p : proofs) aero_free(p);
    }
}

// a constraint program proven over thread ranks from host memory (aero_prove_air_sharded_host): the program's kernel is prepared once,
// every rank copies its share of the columns, the auxiliary-segment exchanges and the constraint-domain gather run as far as the
// kernel-less transcript allows; a rank that fails releases its peers (aero_local_group_abort)
static void sharded_program_proofs() {
    for (uint32_t world : {2u, 4u}) {
        const uint32_t w = 8, log_n = 6;
        std::vector<uint64_t> t((size_t)w << log_n);
        CHECK(aero_fib_trace(w, log_n, t.data()) == AERO_OK);
        const aero_fib_air aux = {3, 4, 2};
        uint8_t* program = nullptr;
        size_t plen = 0;
        CHECK(aero_air_fib_program(w, &aux, &program, &plen) == AERO_OK);
        aero_air* air = nullptr;
        char err[256];
        CHECK(aero_air_load(program, plen, &air, err, sizeof err) == AERO_OK);
        aero_free(program);
        std::vector<uint64_t> pub(w / 2);
        for (uint32_t k = 0; k < w / 2; k++) pub[k] = t[(size_t)(2 * k + 1) * ((size_t)1 << log_n) + ((size_t)1 << log_n) - 1];
        const aero_proof_options opt = {8, 8, 0, 4, 1, 4, 4};
        aero_local_group* g = nullptr;
        CHECK(aero_local_group_create(world, &g) == AERO_OK);
        std::vector<aero_ctx*> ctxs(world, nullptr);
        std::vector<std::thread> th;
        for (uint32_t r = 0; r < world; r++)
            th.emplace_back([&, r] {
                CHECK(aero_ctx_create(0, &ctxs[r]) == AERO_OK);
                aero_comm comm{};
                CHECK(aero_local_group_comm(g, ctxs[r], (int32_t)r, 0, &comm) == AERO_OK);
                uint8_t* proof = nullptr;
                size_t len = 0;
                const int32_t rc = aero_prove_air_sharded_host(ctxs[r], &comm, air, t.data(), log_n, pub.data(), w / 2, &opt, &proof, &len);
                if (rc != AERO_OK) aero_local_group_abort(g);
                aero_free(proof);
            });
        for (auto& x : th) x.join();
        aero_local_group_destroy(g);
        for (aero_ctx* c : ctxs) aero_ctx_destroy(c);
        aero_air_free(air);
    }
}

// an AEROAIR version-2 program (recorded with include/aero_air_builder.hpp): sequence assertions (value tables: host interpolation, sparse
// scatter list, transform), an affine builder, and a GENERAL auxiliary recurrence - the one part of a proof that is real host arithmetic
// (air_host.hip: the row-after-row evaluation), so its buffers and indices are what the sanitizers look at here
static void version2_program_proofs() {
    using aero_air_builder::Builder;
    using aero_air_builder::Expr;
    for (int log_n : {5, 8}) {
        const size_t n = (size_t)1 << log_n;
        std::vector<uint64_t> trace(3 * n);
        uint64_t x = 1, y = 2;
        for (size_t i = 0; i < n; i++) {
            trace[i] = x; trace[n + i] = y; trace[2 * n + i] = (5 + 3 * (uint64_t)i) % P;
            const uint64_t nx = (uint64_t)(((unsigned __int128)x + y) % P), ny = (uint64_t)(((unsigned __int128)y + nx) % P);
            x = nx; y = ny;
        }
        Builder b(3, 2, 2, 1);
        Expr a = b.main(0), bb = b.main(1), na = b.main_next(0), nb = b.main_next(1);
        b.transition(na - (a + bb), 1);
        b.transition(nb - (bb + na), 1);
        b.transition(b.main_next(2) - b.main(2) - 3, 1);
        Expr den = b.rand(0) + b.main(2);
        b.aux_transition((b.aux_next(0) - b.aux(0)) * den - b.main(0), 2);
        Expr sq = b.aux(1) * b.aux(1);
        b.aux_transition(b.aux_next(1) - (sq + b.rand(1) * b.aux(0) + b.main(1)), 2);
        b.assert_single(0, 0, (uint64_t)1);
        b.assert_single(1, -1, b.pub(0));
        std::vector<uint64_t> seq;
        for (size_t i = 0; i < n / 8; i++) seq.push_back(trace[2 * n + 3 + 8 * i]);
        b.assert_sequence(2, 3, 8, seq);
        b.aux_assert_single(0, 0, (uint64_t)0);
        b.aux_assert_sequence(1, 0, (uint32_t)(n / 2), {5, 5});
        b.aux_builder(0, b.constant(0), b.constant(1), Expr(), b.main(0), den);
        b.aux_builder_general(1, b.constant(5), sq + b.rand(1) * b.aux(0) + b.main(1));
        const std::vector<uint8_t> program = b.to_bytes();
        char err[256] = {0};
        aero_air* air = nullptr;
        CHECK(aero_air_load(program.data(), program.size(), &air, err, sizeof err) == AERO_OK);
        aero_ctx* ctx = nullptr;
        CHECK(aero_ctx_create(0, &ctx) == AERO_OK);
        const uint64_t pub[1] = {trace[n + n - 1]};
        for (uint8_t ext : {(uint8_t)1, (uint8_t)2}) {
            const aero_proof_options opt = {8, 8, 0, 4, ext, 4, 4};
            uint8_t* proof = nullptr;
            size_t len = 0;
            const int32_t rc = aero_prove_air_host(ctx, air, trace.data(), (uint32_t)log_n, pub, 1, &opt, &proof, &len);
            CHECK(rc == AERO_OK || aero_last_error(ctx)[0] != 0);
            aero_free(proof);
            // the stage entry point that builds the auxiliary columns: the general column is computed for real, check its recurrence on the host
            aero_matrix *m = nullptr, *auxm = nullptr;
            CHECK(aero_trace_upload(ctx, trace.data(), 3, (uint32_t)log_n, &m) == AERO_OK);
            const uint64_t rands[4] = {11, 12, 13, 14};
            CHECK(aero_aux_columns_program(ctx, air, m, pub, 1, rands, ext, &auxm) == AERO_OK);
            uint32_t cols = 0; uint64_t rows = 0;
            aero_matrix_shape(auxm, &cols, &rows);
            CHECK(cols == 2u * ext && rows == n);
            aero_matrix_free(ctx, auxm); aero_matrix_free(ctx, m);
        }
        aero_ctx_destroy(ctx);
        aero_air_free(air);
    }
}

// Three general recurrences that read each other (the shape of tests/air_examples.py: general_chain_air) over more rows than one chunk of
// the host step: one thread per column, each a chunk behind the column it reads (air_host.hip: `done` counters), the finished columns
// uploaded while the later ones are still computed. Base field: every value is recomputed here with plain 128-bit arithmetic.
static uint64_t mulp(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % P); }
static uint64_t addp(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % P); }
static void general_chain_columns() {
    using aero_air_builder::Builder;
    using aero_air_builder::Expr;
    for (int log_n : {6, 14}) {
        const size_t n = (size_t)1 << log_n;
        std::vector<uint64_t> trace(2 * n);
        uint64_t x = 1, y = 2;
        for (size_t i = 0; i < n; i++) {
            trace[i] = x; trace[n + i] = y;
            const uint64_t nx = addp(x, y), ny = addp(y, nx);
            x = nx; y = ny;
        }
        Builder b(2, 3, 2, 2);
        Expr k = b.periodic({1, 2, 3, 4});
        b.transition(b.main_next(0) - (b.main(0) + b.main(1)), 1);
        b.transition(b.main_next(1) - (b.main(1) + b.main_next(0)), 1);
        Expr e0 = b.aux(0) * b.aux(0) + b.main(0);
        Expr e1 = b.aux(1) * b.aux(0) + k * b.main(1) + b.pub(0) + 5;
        Expr e2 = (b.aux(2) + b.rand(0)) * (b.aux(1) + b.main_next(0));
        b.aux_transition(b.aux_next(0) - e0, 2);
        b.aux_transition(b.aux_next(1) - e1, 2);
        b.aux_transition(b.aux_next(2) - e2, 2);
        b.assert_single(0, 0, (uint64_t)1);
        b.assert_single(1, -1, b.pub(1));
        b.aux_assert_single(0, 0, (uint64_t)3);
        b.aux_assert_single(1, 0, (uint64_t)4);
        b.aux_assert_single(2, 0, (uint64_t)6);
        b.aux_builder_general(0, b.constant(3), e0);
        b.aux_builder_general(1, b.constant(4), e1);
        b.aux_builder_general(2, b.constant(6), e2);
        const std::vector<uint8_t> program = b.to_bytes();
        char err[256] = {0};
        aero_air* air = nullptr;
        CHECK(aero_air_load(program.data(), program.size(), &air, err, sizeof err) == AERO_OK);
        aero_ctx* ctx = nullptr;
        CHECK(aero_ctx_create(0, &ctx) == AERO_OK);
        const uint64_t pub[2] = {77, trace[n + n - 1]};
        const uint64_t rands[4] = {11, 0, 13, 0};
        for (uint8_t ext : {(uint8_t)1, (uint8_t)2}) {
            for (int rep = 0; rep < 2; rep++) {          // twice: the second run reuses the context's pinned block behind the first run's uploads
                aero_matrix *m = nullptr, *auxm = nullptr;
                CHECK(aero_trace_upload(ctx, trace.data(), 2, (uint32_t)log_n, &m) == AERO_OK);
                CHECK(aero_aux_columns_program(ctx, air, m, pub, 2, rands, ext, &auxm) == AERO_OK);
                uint32_t cols = 0; uint64_t rows = 0;
                aero_matrix_shape(auxm, &cols, &rows);
                CHECK(cols == 3u * ext && rows == n);
                std::vector<uint64_t> a((size_t)cols * n);
                CHECK(aero_matrix_download(ctx, auxm, a.data()) == AERO_OK);
                if (ext == 1) {
                    uint64_t g0 = 3, g1 = 4, g2 = 6;
                    for (size_t i = 0; i < n; i++) {
                        CHECK(a[i] == g0 && a[n + i] == g1 && a[2 * n + i] == g2);
                        if (i + 1 == n) break;
                        const uint64_t kk = 1 + (i & 3);
                        const uint64_t n0 = addp(mulp(g0, g0), trace[i]);
                        const uint64_t n1 = addp(addp(addp(mulp(g1, g0), mulp(kk, trace[n + i])), pub[0]), 5);
                        const uint64_t n2 = mulp(addp(g2, rands[0]), addp(g1, trace[i + 1]));
                        g0 = n0; g1 = n1; g2 = n2;
                    }
                }
                aero_matrix_free(ctx, auxm); aero_matrix_free(ctx, m);
            }
        }
        aero_ctx_destroy(ctx);
        aero_air_free(air);
    }
}

// a batch through the pool: every slot's worker thread runs the host pipeline of a proof from host memory (copy gate between the slots),
// the batch comes back with the workers' statuses
static void pool_batches() {
    aero_pool* pool = nullptr;
    CHECK(aero_pool_create(0, 3, &pool) == AERO_OK);
    const uint32_t w = 2, log_n = 6;
    std::vector<uint64_t> t((size_t)w << log_n);
    CHECK(aero_fib_trace(w, log_n, t.data()) == AERO_OK);
    const aero_proof_options opt = {8, 8, 0, 4, 1, 4, 4};
    for (int it = 0; it < 4; it++) {
        const uint64_t* hosts[3] = {t.data(), t.data(), t.data()};
        uint8_t* proofs[3] = {nullptr, nullptr, nullptr};
        size_t lens[3] = {0, 0, 0};
        std::vector<uint64_t> pubs(3 * w);
        // rounds >= 2 with AERO_POOL_PREFETCH_MIN_MB below the trace size (main): the copy of round r + 1 on the copy stream, gated between the slots,
        // while round r is proven from the other landing buffer
        const int32_t rc = aero_pool_prove_fib_host(pool, hosts, w, log_n, 3, nullptr, &opt, 2 + it, proofs, lens, pubs.data());
        CHECK(rc == AERO_OK || aero_last_error(aero_pool_ctx(pool, 0))[0] != 0);
        for (uint8_t* p : proofs) aero_free(p);
    }
    {   // a queue of 7 traces over 3 slots: every proof comes back, every slot prefetches its next trace
        const uint64_t* q[7];
        uint8_t* qp[7];
        size_t ql[7];
        std::vector<uint64_t> qpub(7 * (w / 2));
        for (auto& x : q) x = t.data();
        const int32_t rc = aero_pool_prove_fib_queue(pool, q, 7, w, log_n, nullptr, &opt, qp, ql, qpub.data());
        CHECK(rc == AERO_OK || aero_last_error(aero_pool_ctx(pool, 0))[0] != 0);
        for (uint8_t* p : qp) aero_free(p);
    }
    aero_pool_destroy(pool);
}

extern "C" uint64_t hipstub_launches();
extern "C" uint64_t hipstub_copies();

int main() {
    setenv("AERO_POOL_PREFETCH_MIN_MB", "0.0005", 1);      // the pool prefetches its 1 KiB traces too
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 4; t++) th.emplace_back(lifetimes, t);
        for (int t = 0; t < 3; t++) th.emplace_back(stages, t);
        for (int t = 0; t < 2; t++) th.emplace_back(proofs, t);
        th.emplace_back(pools);
        for (auto& t : th) t.join();
    }
    for (uint32_t world : {2u, 4u, 8u}) local_group(world, 24, false);
    local_group(4, 12, true);                 // one rank leaves: its peers must come back with an error, not hang
    sharded_proofs();
    setenv("AERO_EXCHANGE_CHUNKS", "4", 1);        // the same proofs with every commitment's exchange in pieces: the second stream and its events
    sharded_proofs();
    unsetenv("AERO_EXCHANGE_CHUNKS");
    sharded_program_proofs();
    version2_program_proofs();
    general_chain_columns();
    pool_batches();
    printf("host logic ok: %llu copies moved, %llu kernel launches skipped\n", (unsigned long long)hipstub_copies(), (unsigned long long)hipstub_launches());
    return 0;
}
