// Host side of the MI355X STARK prover: the C++ mirror of the Winterfell `Prover` surface that Aero drives.
//
// Names, argument meaning and stage boundaries follow the reference's call sites (the trait bodies are in the
// absent winterfell submodule):
//   ProofOptions .................. miden-proof-generator/src/main.rs:23 (`with_96_bit_security`),
//                                   aero-sdk/miden-wasm/src/convert/convert_inputs.rs:54-66 (7-field ctor)
//   ProverChannel ................. aero-sdk/miden-wasm/src/proving_worker.rs:264-268
//   Matrix / interpolate_columns / evaluate_columns_over ... proving_worker.rs:271-274, utils.rs:235-236
//   MerkleTree::new ............... proving_worker.rs:161-162
//   commit_to_trace_and_validate .. proving_worker.rs:323-332   (fork-only split of Prover::prove)
//   evaluate_constraints .......... proving_worker.rs:355-439, constraints_worker.rs:14-79
//   prove_after_constraint_eval ... proving_worker.rs:344-352   (fork-only)
//   Prover::prove ................. proving_worker.rs:465-467, miden-proof-generator/src/main.rs:31
//   StarkProof::to_bytes .......... miden-proof-generator/src/main.rs:38 (layout: SURVEY a18)
// The transcript (random coin) lives on the host and is the only serialisation point between device stages.
#pragma once
#include <cstdlib>
#include <memory>

#include "aero_internal.hpp"

namespace aero {

namespace air { struct Program; }

typedef std::vector<uint8_t> Bytes;

enum HashFn : uint8_t { HASH_BLAKE2S_256 = 4 };
enum FieldExtension : uint8_t { EXT_NONE = 1, EXT_QUADRATIC = 2 };

struct ProofOptions {
    uint8_t num_queries, blowup_factor, grinding_factor, hash_fn, field_extension, fri_folding_factor, fri_log_max_remainder;
    static ProofOptions with_96_bit_security() { return ProofOptions{27, 8, 16, HASH_BLAKE2S_256, EXT_NONE, 8, 8}; }
    static ProofOptions from_bytes(const uint8_t b[7]) { return ProofOptions{b[0], b[1], b[2], b[3], b[4], b[5], b[6]}; }
    void validate() const;
};

// Random coin + proof accumulation (host). Mirror: src/stark_verifier/crypto/random.cairo.
struct HostCoin {
    Digest seed;
    uint64_t counter = 0;
    static HostCoin from_elements(const uint64_t* e, uint32_t n);
    void reseed(const Digest& d) { seed = b2s::merge(seed, d); counter = 0; }
    void reseed_with_int(uint64_t v) { seed = b2s::merge_with_int(seed, v); counter = 0; }
    Digest next() { counter += 1; return b2s::merge_with_int(seed, counter); }
    uint64_t draw_base();
    gl::E2 draw_quad();
    template <class F> typename F::T draw();
    std::vector<uint64_t> draw_integers(size_t k, uint64_t domain);
};
template <> inline uint64_t HostCoin::draw<gl::FB>() { return draw_base(); }
template <> inline gl::E2 HostCoin::draw<gl::FQ>() { return draw_quad(); }

struct QueriesBytes { Bytes values, paths; };

// StarkProof container (byte layout: SURVEY a18).
struct StarkProof {
    uint8_t main_width = 0, aux_width = 0, aux_rands = 0, log_n = 0;
    ProofOptions options{};
    Bytes commitments;
    std::vector<QueriesBytes> trace_queries;
    QueriesBytes constraint_queries;
    Bytes ood_trace_states, ood_evaluations;
    std::vector<QueriesBytes> fri_layers;
    Bytes fri_remainder;
    uint64_t pow_nonce = 0;
    Bytes to_bytes() const;
};

// Column-major device matrix (winter `Matrix<Felt>`: num_rows/num_cols/columns, utils.rs:235-236).
struct Matrix {
    DevBuf<uint64_t> data;
    size_t rows = 0;
    int cols = 0;
    Matrix() {}
    Matrix(Context* ctx, int ncols, size_t nrows) : data(ctx, (size_t)ncols * nrows), rows(nrows), cols(ncols) {}
    uint64_t* col(int c) const { return data.get() + (size_t)c * rows; }
    size_t num_rows() const { return rows; }
    int num_cols() const { return cols; }
};

// Device Merkle tree: nodes[1] = root, leaves at nodes[n + j].
// `skip` = number of lowest levels that are NOT stored (0 or 3): nodes then holds only heap indices < 2n >> skip and the
// openings recompute the few low nodes they need from the leaf source (which must outlive the tree).
struct MerkleTree {
    DevBuf<Digest> nodes;
    size_t n = 0;
    int skip = 0;
    int src_kind = 0;          // 0 none (everything stored), 1 RowSrc, 2 FriSrc
    RowSrc row_src{};
    FriSrc fri_src{};
    Digest root_host{};
    MerkleTree() {}
    MerkleTree(Context* ctx, size_t leaves, int skip_levels = 0) : nodes(ctx, (2 * leaves) >> skip_levels), n(leaves), skip(skip_levels) {}
    size_t stored_limit() const { return (2 * n) >> skip; }
    Digest* leaves() const { return nodes.get() + n; }   // only valid when skip == 0
    const Digest& root() const { return root_host; }
    int depth() const { int d = 0; while (((size_t)1 << d) < n) d++; return d; }
};

// Built-in AIR: FibAir(W). Pair k = columns (2k, 2k+1) = (a, b); a' = a + b, b' = b + a';
// seeds (1+2k, 2+2k); assertions a(0), b(0), b(n-1) = results[k]; public inputs = results.
// Optional auxiliary segment (synthetic stand-in for Miden's multiset-check columns): aux_width columns
// over E built after the main commitment from aux_rands coin elements, p_c(0) = 1, p_c(i+1) = p_c(i) * (r_(c mod R) +
// main_(c mod W)(i))^(D-1); one degree-D transition constraint and the assertion p_c(0) = 1 per column. D = 2 is the plain
// multiset-check shape; larger D raises the constraint-evaluation blowup and the number of composition columns (8 for D in
// [5, 8], the shape of the golden Miden proof).
struct FibAir {
    uint32_t width = 0;
    int log_n = 0;
    std::vector<uint64_t> results;
    uint32_t aux_width = 0, aux_rands = 0;
    uint32_t aux_degree = 2;   // aux transition constraint p' = p * (r + main)^(aux_degree - 1), degree in [2, 8]
    size_t trace_length() const { return (size_t)1 << log_n; }
    size_t num_transition_constraints() const { return width + aux_width; }
    size_t num_assertions() const { return width + width / 2 + aux_width; }
    // constraint-evaluation blowup = number of composition columns = max(next_pow2(max constraint degree), 2)
    size_t ce_blowup_factor() const { size_t d = aux_width ? aux_degree : 1, e = 2; while (e < d) e <<= 1; return e; }
    static size_t plain_ce_blowup_factor() { return 2; }   // FibAir without an auxiliary segment
};

struct StageMs {   // per-stage wall time (ms) of the last prove(), names after proving_worker.rs console labels
    double interpolate = 0, lde = 0, trace_commit = 0, constraints = 0, composition = 0, comp_commit = 0, ood = 0, deep = 0,
           fri = 0, grind = 0, queries = 0, total = 0;
};

// Exchange steps of a proof sharded over `world` GPUs (include/aero_stark.h: aero_comm). Rank k owns the LDE rows
// j = k (mod world) = the coset 7 w_N^k <w_(N/world)>; world == 1 means the whole domain and no exchange.
struct ShardComm {
    int rank = 0, world = 1;
    void* user = nullptr;
    int32_t (*all_to_all)(void*, const void*, void*, uint64_t) = nullptr;
    int32_t (*all_gather)(void*, const void*, void*, uint64_t) = nullptr;
    int32_t (*all_reduce_sum_u64)(void*, void*, uint64_t) = nullptr;
    int32_t (*send_recv)(void*, const void*, int32_t, void*, int32_t, uint64_t) = nullptr;   // optional
    uint32_t min_peer_digests = 2048;
    bool stream_ordered = false;   // the callbacks enqueue on the context's stream (aero_comm flag AERO_COMM_STREAM_ORDERED)
};
// A commitment as the opening phase sees it: either a whole tree on this GPU, or this rank's contiguous subtree of
// n_global / world leaves plus the top log2(world) levels (host copy, heap order: top[1] = root, top[world + r] = subtree r).
struct Commitment {
    MerkleTree tree;
    bool sharded = false;
    size_t n_global = 0;
    int leaf_parts_log = 0;   // > 0: the subtree's leaf level is stored as 2^k pieces in arrival order (Context::merkle_build_parts)
    std::vector<Digest> top;
    Digest root{};
};

// Inputs of the DEEP composition [a14]: the OOD point, the OOD frame (main then aux columns) and composition values, and
// the coefficients in draw order (3 per trace column, one per composition column, then lambda, mu).
template <class F> struct DeepInputs {
    typedef typename F::T T;
    T z;
    std::vector<T> ood_cur, ood_next, ood_h, da, db, dg, dc;
    T lambda, mu;
};
// Result of the FRI commit phase [a15]: evaluations per layer ([DEG][dom] component arrays, natural order; the last entry
// is the remainder layer) and one commitment per layer incl. the remainder.
struct FriLayers {
    std::vector<DevBuf<uint64_t>> vals;
    std::vector<Commitment> coms;
    uint64_t lde_size = 0;
    int deg = 1, fold = 0, layers = 0;
};

class Prover {
public:
    Prover(Context* ctx, const ProofOptions& opt) : ctx_(ctx), opt_(opt) { opt_.validate(); }
    void set_comm(const ShardComm& c) { comm_ = c; }
    void set_aux_segment(uint32_t aux_width, uint32_t aux_rands, uint32_t aux_degree = 2) {
        aux_width_ = aux_width; aux_rands_ = aux_width ? aux_rands : 0; aux_degree_ = aux_degree;
    }
    // The next prove() takes its trace from HOST memory (column-major width x 2^log_n; `trace_dev` is then ignored): it is copied
    // straight into the interpolation buffer (a device copy is kept only while an auxiliary segment still has to be built from it).
    // *verdict (pinned) receives 0 when every element was canonical.
    void set_host_trace(const uint64_t* trace_host, unsigned int* verdict) { host_trace_ = trace_host; host_verdict_ = verdict; }
    // The host trace has already been sent on its way: a copy into the buffer prove() is given as `trace_dev` is in flight on another stream and
    // `ready` fires when it has landed. prove() waits for the event on its stream, reads the public inputs' row from `trace_host`, and lets the
    // first inverse pass check the canonical form (as set_host_trace does); the landing buffer is not modified. One GPU only.
    void set_landed_trace(const uint64_t* trace_host, hipEvent_t ready, unsigned int* verdict) { host_trace_ = trace_host; host_verdict_ = verdict; landed_ready_ = ready; landed_ = true; }
    // Prove against a program AIR (air_program.hpp; include/aero_air.h) instead of the built-in FibAir: the constraint set, the
    // assertions, the auxiliary segment's shape and construction all come from the program; `pub` = its public inputs (they seed
    // the coin). The program must outlive the proof.
    void set_program(const air::Program* prog, const std::vector<uint64_t>& pub) { program_ = prog; program_pub_ = pub; }
    const ProofOptions& options() const { return opt_; }
    // trace: device, column-major W x 2^log_n (not modified). Returns StarkProof::to_bytes().
    Bytes prove(const uint64_t* trace_dev, uint32_t width, int log_n, std::vector<uint64_t>* pub_inputs_out);
    StageMs last_stage_ms;
    bool collect_stage_times = false;   // adds a stream sync per stage
    bool fri_tail = getenv("AERO_FRI_TAIL") ? getenv("AERO_FRI_TAIL")[0] != '0' : true;   // small FRI layers in one launch (Context::fri_tail)
    bool compact_rows = getenv("AERO_COMPACT_ROWS") ? getenv("AERO_COMPACT_ROWS")[0] != '0' : true;   // compact every-k-th-row LDE copies for constraints / DEEP
    bool low_level_skip = true;         // large trees: the 3 lowest Merkle levels are not stored but recomputed by the openings
    bool exchange_rows = getenv("AERO_EXCHANGE_ROWS") ? getenv("AERO_EXCHANGE_ROWS")[0] != '0' : true;   // sharded: rows instead of digests when shorter
    int exchange_chunks = getenv("AERO_EXCHANGE_CHUNKS") ? atoi(getenv("AERO_EXCHANGE_CHUNKS")) : 1;   // sharded: pieces per peer of a commitment's exchange, overlapped with hashing (1 = one exchange)
    // stop behind the main segment's commitment (the first half of the reference's fork-only split `commit_to_trace_and_validate`,
    // proving_worker.rs:323-332): prove() then returns root || subtree roots (world digests; the root itself on one GPU) instead of a proof
    bool trace_commit_only = false;
    bool h2d_pipeline = getenv("AERO_H2D_PIPELINE") ? getenv("AERO_H2D_PIPELINE")[0] != '0' : true;   // wide host traces travel in column groups behind the transforms

    // ---- stage-level entry points (the reference's split API; also what the C ABI exposes) ----
    // interpolate_columns: evaluations on <w_n> -> polys (bit-reversed coefficients pre-scaled by 7^i)
    Matrix interpolate_columns(const uint64_t* trace_dev, uint32_t width, int log_n);
    // evaluate_columns_over: polys -> LDE over 7<w_N>, natural row order
    Matrix evaluate_columns_over(const Matrix& polys, int log_blowup);
    // row hashes + MerkleTree::new. keep_low_levels = false drops the 3 lowest levels (recomputed on demand from `lde`,
    // which must then outlive the tree).
    MerkleTree commit_to_rows(const Matrix& lde, bool keep_low_levels = true);
    MerkleTree commit_fri_layer(const FriSrc& src, bool keep_low_levels = false);
    // same without waiting for the root: root_host is NOT set; the root sits in nodes[1] on the device
    // `coin` (optional): the tree build's last launch also reseeds the device coin with the root and draws the folding challenge
    MerkleTree commit_fri_layer_async(const FriSrc& src, const CoinStep* coin = nullptr);

    // H on the constraint domain (components [DEG][ce_n], evaluations on h<w_ce>) -> coefficients of the C column
    // polynomials, in place: chunk c of component d = column c (internal form, pre-scaled by h^i)   [a11]
    void composition_from_evaluations(uint64_t* hbuf, int deg, int log_ce, int log_c, uint64_t h);
    // DEEP composition over the coset h<w_M>, M = n << log_bl: evaluates on every (M/n)-th row, interpolates, extends.
    // tlde: W x M, clde: (C*DEG) x M (column c*DEG + d), alde: (A*DEG) x M or nullptr. Returns [DEG][M].   [a14]
    // Optional compact copies (every 2^log_step-th row, written by the LDE's last pass): nullptr = walk the full matrix.
    struct DeepCompact { const uint64_t* t = nullptr; int t_log = 0; size_t t_stride = 0; const uint64_t* c = nullptr; int c_log = 0;
                         const uint64_t* a = nullptr; int a_log = 0; size_t a_stride = 0; };
    template <class F>
    DevBuf<uint64_t> deep_compose(const uint64_t* tlde, const uint64_t* clde, const uint64_t* alde, uint32_t W, uint32_t A, uint32_t C, int log_n,
                                  int log_bl, uint64_t h, const DeepInputs<F>& in, const DeepCompact* compact = nullptr);
    // FRI commit phase on one GPU: per layer transpose-hash-commit, reseed, draw alpha, fold; `roots` receives every
    // commitment (layers + remainder) in order.   [a15]
    template <class F> FriLayers fri_build_layers(DevBuf<uint64_t>&& evals, uint64_t N, HostCoin& coin, Bytes* roots);
    // serialised FriProof for LDE-domain query positions: u8 #layers, per layer Queries, u16 remainder, u8 0   [a17]
    template <class F> Bytes fri_open(const FriLayers& fl, const std::vector<uint64_t>& positions);

private:
    template <class F> Bytes prove_impl(const uint64_t* trace_dev, uint32_t width, int log_n, std::vector<uint64_t>* pub_out);
    // sharded commitment: `local` = this rank's coset leaves (count L); returns the subtree over global leaves [rank*L, (rank+1)*L)
    Commitment commit_exchange(DevBuf<Digest>& local, size_t L);
    Commitment commit_exchange_rows(const Matrix& lde);    // rows shorter than a digest: exchange the rows, hash on arrival
    // the same two commitments with the exchange cut into `chunks` pieces per peer, every piece's all-to-all on the proving stream while the row
    // hashing of the neighbouring piece runs on the context's second stream (AERO_EXCHANGE_CHUNKS)
    Commitment commit_exchange_chunked(const Matrix& lde, int chunks, bool rows_path);
    void finish_exchange(Commitment& c);
    void comm_all_to_all(const void* send, void* recv, size_t bytes);
    void comm_all_gather(const void* send, void* recv, size_t bytes);
    void comm_all_reduce(uint64_t* buf, size_t count);
    void comm_send_recv(const void* send, int to, void* recv, int from, size_t bytes);
    void gather_h_cosets(const uint64_t* mine, uint64_t* all, size_t bytes, int q, int step);
    Context* ctx_;
    ProofOptions opt_;
    ShardComm comm_;
    uint32_t aux_width_ = 0, aux_rands_ = 0, aux_degree_ = 2;
    const uint64_t* host_trace_ = nullptr;
    unsigned int* host_verdict_ = nullptr;
    hipEvent_t landed_ready_ = nullptr;
    bool landed_ = false;
    const air::Program* program_ = nullptr;
    std::vector<uint64_t> program_pub_;
};

// BatchMerkleProof node selection (winter-crypto 0.4 MerkleTree::prove_batch restated):
// returns, per vector, the node indices (into the 2n-slot node array) whose digests are serialised.
std::vector<std::vector<uint64_t>> batch_proof_indices(size_t n_leaves, const std::vector<uint64_t>& positions);
// The same plan in flat form (two allocations instead of one per path: the prover builds 6-8 of them on the critical path of every
// proof): path p holds count[p] node indices at idx[p * cap ...].
struct BatchPlan {
    size_t cap = 0;
    std::vector<uint8_t> count;
    std::vector<uint64_t> idx;
    size_t paths() const { return count.size(); }
    size_t total() const { size_t t = 0; for (uint8_t c : count) t += c; return t; }
    template <class Fn> void for_each(Fn&& fn) const { for (size_t p = 0; p < count.size(); p++) for (uint8_t k = 0; k < count[p]; k++) fn(idx[p * cap + k]); }
};
BatchPlan batch_proof_plan(size_t n_leaves, const std::vector<uint64_t>& positions);
// serialised BatchMerkleProof nodes for `positions` (u8 #vectors, per vector u8 len + digests)
Bytes open_batch(Context* ctx, const MerkleTree& tree, const std::vector<uint64_t>& positions);
std::vector<uint64_t> fold_positions(const std::vector<uint64_t>& positions, uint64_t source_domain, uint64_t folding_factor);
int num_fri_layers(uint64_t domain, uint64_t fold, uint64_t max_remainder);

}  // namespace aero
