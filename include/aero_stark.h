/*
 * aero_stark.h — C ABI of libaero_stark.so, the MI355X-native backend for the Winterfell proving hot path that
 * Aero's miden-proof-generator drives (Goldilocks NTT/LDE, constraint evaluation, BLAKE2s Merkle commitment,
 * DEEP, FRI, proof bytes).
 *
 * This is the drop-in boundary: what a Rust `impl winter_prover::Prover` (or the browser worker protocol) would bind
 * through FFI. Every entry point cites the reference interface it replaces (paths relative to the reference repo
 * starkoracles/Aero @ v1). Conventions:
 *   - plain C types only; opaque handles for device-resident objects; caller owns every host buffer it passes in;
 *   - every function returns an int32 status: 0 = ok, negative = error (AERO_E_*); aero_last_error() gives the text;
 *   - no exceptions cross the boundary; a context is NOT thread-safe (one host thread drives it);
 *   - all integers little-endian, field elements canonical u64 < p = 2^64 - 2^32 + 1, digests 32 bytes;
 *   - matrices are COLUMN-MAJOR (column c occupies elements [c*rows, (c+1)*rows)), like winter's Matrix<Felt>;
 *   - there is no CPU fallback: without a HIP device aero_ctx_create fails with AERO_E_HIP.
 */
#ifndef AERO_STARK_H
#define AERO_STARK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AERO_OK 0
#define AERO_E_BAD_ARG (-1)
#define AERO_E_OOM (-2)
#define AERO_E_HIP (-3)
#define AERO_E_COMM (-4)
#define AERO_E_UNSUPPORTED (-5)
#define AERO_E_INTERNAL (-6)
#define AERO_E_VERIFY (-7) /* aero_verify_fib: the proof was parsed and rejected */
#define AERO_E_SELF_VERIFY (-8) /* a prove call's own proof was rejected by the library's verifier before it left: no bytes returned (aero_ctx_set_self_verify) */

typedef struct aero_ctx aero_ctx;
typedef struct aero_matrix aero_matrix; /* device, column-major u64 matrix */
typedef struct aero_tree aero_tree;     /* device Merkle tree: node 1 = root, children 2i / 2i+1, leaves at n + j */

/* The 7 option bytes of the proof context, in serialisation order.
 * Replaces winter_air::ProofOptions::new(num_queries, blowup_factor, grinding_factor, hash_fn, field_extension,
 * fri_folding_factor, fri_max_remainder_size): aero-sdk/miden-wasm/src/convert/convert_inputs.rs:54-66,
 * aero-sdk/proto/context.proto:24-33; `ProofOptions::with_96_bit_security()` (miden-proof-generator/src/main.rs:23)
 * = {27, 8, 16, 4, 1, 8, 8}. hash_fn: 4 = Blake2s_256 (only one implemented). field_extension: 1 = None,
 * 2 = Quadratic. fri_log_max_remainder = log2 of the max remainder size (256 -> 8). */
typedef struct aero_proof_options {
    uint8_t num_queries;
    uint8_t blowup_factor;
    uint8_t grinding_factor;
    uint8_t hash_fn;
    uint8_t field_extension;
    uint8_t fri_folding_factor;
    uint8_t fri_log_max_remainder;
} aero_proof_options;

/* Parameters of the built-in AIR's optional auxiliary segment (described at aero_prove_fib_air below). */
typedef struct aero_fib_air {
    uint32_t aux_width;   /* 0 = no auxiliary segment */
    uint32_t aux_rands;
    uint32_t aux_degree;  /* ignored when aux_width = 0 */
} aero_fib_air;

/* ---- context --------------------------------------------------------------------------------------------------- */
/* One context = one GPU + one HIP stream + a device memory pool. (Reference has no analogue: its workers are
 * web workers, aero-sdk/miden-wasm/src/pool.rs:28-45.)
 * A context is SINGLE-device on purpose. SURVEY 8(b) sketched `ctx_create(device list)`; what was built instead keeps the
 * one-process-per-GPU model of the launch contract intact and composes from the outside: several GPUs work on ONE proof through an
 * aero_comm handed to the *_sharded entry points - RCCL between processes (aero_rccl_*), or the in-process group for a host that
 * drives several devices from one process with a context per device (aero_local_group_*); several proofs in flight on one GPU
 * are an aero_pool. A multi-device context would have had to own that choice (threads or processes, which exchange library) itself. */
int32_t aero_device_count(void);
int32_t aero_ctx_create(int32_t device_id, aero_ctx** out);
void aero_ctx_destroy(aero_ctx* ctx);
/* Prove-then-verify, as both reference callers do before a proof leaves the process (miden-proof-generator/src/main.rs:47
 * `miden::verify(...)` right after `prove`; aero-sdk/miden-wasm/src/proving_worker.rs:196-203 verifies inside `prove` and turns a
 * rejection into an error instead of a result). With the check on, every aero_prove_* call of this context runs the library's own
 * host verifier (aero_verify_fib / aero_verify_air: every check of src/stark_verifier plus the out-of-domain constraint check, the
 * proof's options and trace length pinned to the call's) on the bytes it is about to return; a rejection yields AERO_E_SELF_VERIFY,
 * the reason in aero_last_error, and NO bytes (*proof = NULL). Modes:
 *   AERO_SELF_VERIFY_AUTO (default)  on for every proof made by more than one rank (aero_comm world > 1: the *_sharded* entry points,
 *                                    the local group, the RCCL communicator - each rank checks the bytes it returns), off on one GPU
 *   AERO_SELF_VERIFY_OFF / _ON       never / always (0.7 ms of host time for a 2^20 x 2 proof; it grows with the proof's size, not the trace's)
 * The environment variable AERO_SELF_VERIFY=0|1 sets the initial mode of contexts created afterwards (pools included).
 * What the check can and cannot see: a proof the verifier accepts is a valid proof of the statement; a corrupted exchange whose damage
 * none of the queries touches (one wrong leaf among 2^23) yields bytes that differ from the single-GPU proof and still verify. */
#define AERO_SELF_VERIFY_AUTO (-1)
#define AERO_SELF_VERIFY_OFF 0
#define AERO_SELF_VERIFY_ON 1
int32_t aero_ctx_set_self_verify(aero_ctx* ctx, int32_t mode);
/* Field-arithmetic self test on the context's GPU: the device formulations of Goldilocks add / sub / mul / inverse, the
 * power-of-two multiplications of the NTT butterflies and the F_p^2 product, run inside one kernel on `samples` random operand
 * pairs plus the carry-boundary edge cases, against host results computed with 128-bit integers. AERO_OK, or AERO_E_INTERNAL with
 * the failing operation and operands in aero_last_error - a deployment can run it once per driver / compiler update. (The
 * reference relies on Rust's u128 arithmetic in winter-math and needs no such check.) */
int32_t aero_selftest(aero_ctx* ctx, uint32_t samples, uint64_t seed);
/* Wait until everything enqueued on the context's stream has completed (entry points that return host data do this
 * themselves; needed after stream-ordered exchanges issued through an aero_comm outside a proof). */
int32_t aero_ctx_synchronize(aero_ctx* ctx);
/* Text of the last error on this context; ctx may be NULL for errors of aero_ctx_create itself. */
const char* aero_last_error(const aero_ctx* ctx);
void aero_free(void* p); /* releases buffers returned through uint8_t** out-parameters */
/* Crash diagnostics for the hosting process (opt-in; also installed at load time when AERO_CRASH_TRACE=1 is in the environment).
 * Installs a std::terminate handler and SIGABRT / SIGSEGV / SIGBUS handlers that write the escaping exception's what(), the
 * thread and a native backtrace with write(2) to the stderr that existed at install time, to the current fd 2 and, when
 * AERO_CRASH_LOG names a file, to that file - then chain to the previous handlers (core dumps, Python's faulthandler). A host that
 * drives the prover repeatedly from one long-lived worker (proving_worker.rs:124-223) gets the cause of a native abort even
 * when a harness has redirected its descriptors. Idempotent. */
int32_t aero_install_crash_diagnostics(void);

/* ---- matrices -------------------------------------------------------------------------------------------------- */
/* Trace hand-over. Replaces `trace.main_segment()` handed to the prover: proving_worker.rs:272. */
int32_t aero_trace_upload(aero_ctx* ctx, const uint64_t* col_major, uint32_t width, uint32_t log_n, aero_matrix** out);
int32_t aero_matrix_shape(const aero_matrix* m, uint32_t* cols, uint64_t* rows);
int32_t aero_matrix_download(aero_ctx* ctx, const aero_matrix* m, uint64_t* col_major_out);
/* Device address of the matrix's column-major storage (for a caller's own kernels or collectives on the context's GPU; work the
 * caller enqueues elsewhere must be ordered against the context's stream, e.g. with aero_ctx_synchronize). */
int32_t aero_matrix_device_ptr(const aero_matrix* m, uint64_t** dev_ptr_out);
void aero_matrix_free(aero_ctx* ctx, aero_matrix* m);
/* Synthetic Fibonacci trace (host buffer, column-major width x 2^log_n): pair k = columns (2k, 2k+1) = (a, b),
 * a' = a + b, b' = b + a', seeds (1 + 2k, 2 + 2k). Pure function of (width, log_n). */
int32_t aero_fib_trace(uint32_t width, uint32_t log_n, uint64_t* col_major_out);

/* ---- trace files: how a trace produced elsewhere reaches this library ------------------------------------------------------------- */
/* The reference obtains its trace from the VM in the same process (`processor::execute`: miden-proof-generator/src/main.rs:20-31,
 * proving_worker.rs:239-259 `build_execution_trace`). For a trace dumped by another program the hand-over is a file:
 *   bytes 0..7   "AEROTRC" followed by the format version byte 1
 *   u32 x 6      width, log_n, air_id, aux_width, aux_rands, aux_degree   (little-endian)
 *   u64 x width * 2^log_n   the main segment, COLUMN-MAJOR (column c = elements [c * 2^log_n, (c + 1) * 2^log_n)), canonical (< p)
 * air_id names the constraint set the trace is to be proven against: AERO_AIR_FIB = 0 is the built-in FibAir (aux_* = its
 * optional auxiliary segment, see aero_fib_air); 1 is reserved for Miden's ProcessorAir, whose constraints are not part of the
 * reference mount - loading such a file yields AERO_E_UNSUPPORTED, not a proof against the wrong AIR.
 * aero_trace_file_load streams the file to the device through pinned double buffers (disk reads overlap the copies). */
#define AERO_AIR_FIB 0u
#define AERO_AIR_MIDEN_PROCESSOR 1u
int32_t aero_trace_file_write(const char* path, const uint64_t* col_major, uint32_t width, uint32_t log_n, uint32_t air_id,
                              const aero_fib_air* air);
int32_t aero_trace_file_info(const char* path, uint32_t* width, uint32_t* log_n, uint32_t* air_id, aero_fib_air* air);
int32_t aero_trace_file_load(aero_ctx* ctx, const char* path, aero_matrix** trace_out, uint32_t* air_id, aero_fib_air* air);

/* ---- stage 1: interpolation and low-degree extension ----------------------------------------------------------------- */
/* Replaces `main_trace.interpolate_columns()` (proving_worker.rs:273). Output polynomials are kept in the backend's
 * internal form (bit-reversed coefficient order, coefficient i pre-multiplied by 7^i); they are only meaningful as
 * input to aero_evaluate_columns_over / aero_poly_eval. */
int32_t aero_interpolate_columns(aero_ctx* ctx, const aero_matrix* trace, aero_matrix** polys);
/* Replaces `trace_polys.evaluate_columns_over(&domain)` (proving_worker.rs:274): evaluations over 7 * <w_N>,
 * N = rows * 2^log_blowup, natural order (row j <-> x_j = 7 * w_N^j). */
int32_t aero_evaluate_columns_over(aero_ctx* ctx, const aero_matrix* polys, uint32_t log_blowup, aero_matrix** lde);
/* Evaluate every column polynomial at a base-field point z (out: cols values). Replaces `trace_polys.get_ood_frame(z)`
 * inside prove_after_constraint_eval (proving_worker.rs:344-352). */
int32_t aero_poly_eval(aero_ctx* ctx, const aero_matrix* polys, uint64_t z, uint64_t* out);

/* ---- row hashing and Merkle commitment ---------------------------------------------------------------------------- */
/* The reference's hashing seam: HashingWorkItem{data: Vec<Vec<u64>> rows, batch_idx} -> HashingResult{hashes:
 * Vec<[u8;32]>} (aero-sdk/miden-wasm/src/utils.rs:358-362,411-415; hashing_worker.rs:12-26). rows_row_major holds
 * n_rows rows of `width` elements each; digests_out receives n_rows * 32 bytes. digest = BLAKE2s-256 over the elements,
 * each serialised as 32 little-endian bytes (src/stark_verifier/crypto/random.cairo:93-104). */
int32_t aero_hash_rows(aero_ctx* ctx, const uint64_t* rows_row_major, uint32_t width, uint64_t n_rows, uint8_t* digests_out);
/* Same, for the rows of a device matrix (the `read_row_into` gather of proving_worker.rs:293-296 is the coalesced
 * column-major read on the device). digests_out may be NULL. */
int32_t aero_hash_matrix_rows(aero_ctx* ctx, const aero_matrix* m, uint8_t* digests_out);
/* Replaces `MerkleTree::new(trace_row_hashes)` (proving_worker.rs:161-162). n_leaves must be a power of two >= 2. */
int32_t aero_merkle_from_leaves(aero_ctx* ctx, const uint8_t* leaves, uint64_t n_leaves, aero_tree** out, uint8_t root_out[32]);
/* Row hashing + tree in one call (winter `commit_to_rows`). */
int32_t aero_merkle_commit_rows(aero_ctx* ctx, const aero_matrix* m, aero_tree** out, uint8_t root_out[32]);
/* Replaces `MerkleTree::prove_batch(positions)` + `BatchMerkleProof::serialize_nodes()`: u8 #vectors, then per vector
 * u8 len + len * 32 bytes (layout: SURVEY.md App. A.2; consumer: miden-to-cairo-parser/src/lib.rs:363-388). */
int32_t aero_merkle_open_batch(aero_ctx* ctx, const aero_tree* tree, const uint64_t* positions, uint32_t k, uint8_t* out,
                               size_t cap, size_t* out_len);
/* All 2n node slots (slot 0 zeroed, slot 1 = root, slots n.. = leaves); for tests. */
int32_t aero_merkle_nodes(aero_ctx* ctx, const aero_tree* tree, uint8_t* out);
void aero_tree_free(aero_ctx* ctx, aero_tree* tree);

/* ---- constraint evaluation ------------------------------------------------------------------------------------------ */
/* The reference's constraint seam for the built-in FibAir: ConstraintComputeWorkItem{.., constraint_coeffs,
 * trace_lde, ComputationFragment{fragment_offset, num_fragments}} -> ConstraintComputeResult{frag_index, frag_num,
 * constraint_evaluations: Vec<Vec<u64>> column-major} (utils.rs:302-347,417-422; constraints_worker.rs:14-79:
 * `evaluation_table.fragments(n)[k]`, `evaluator.evaluate_fragment`). Columns = one per divisor: [transition,
 * boundary(step 0), boundary(step n-1)], numerators only. coeffs = (alpha, beta) pairs, first one per transition
 * constraint (width of them) then one per assertion (width + width/2), each element `deg` u64 (deg = 1, or 2 for the
 * quadratic extension). out_cols receives 3*deg columns of frag_rows = (2 * trace_len / num_fragments) values each;
 * *frag_index_out = first constraint-domain row of the fragment (`frag.offset()`). */
int32_t aero_eval_constraints_fib(aero_ctx* ctx, const aero_matrix* trace_lde, uint32_t log_blowup, const uint64_t* results,
                                  const uint64_t* coeffs, uint8_t field_extension, uint32_t fragment_offset,
                                  uint32_t num_fragments, uint64_t* out_cols, uint64_t* frag_index_out);

/* The same seam for the built-in AIR WITH its auxiliary segment (aero_fib_air; air == NULL or aux_width == 0 is exactly
 * aero_eval_constraints_fib): `aux_lde` = the auxiliary segment's LDE as (aux_width * deg) component columns over the same rows
 * (column c * deg + d), `rands` = the aux_rands drawn elements (deg u64 each) - what the reference ships as
 * `ConstraintComputeWorkItem.aux_rand_elements` (utils.rs:302-347). The constraint domain has C * trace_len points, C = 2 / 4 / 8 for
 * aux_degree 2 / 3-4 / 5-8; coeffs = (alpha, beta) per transition constraint (width main, then aux_width aux), then per assertion
 * (width + width/2 main, then aux_width aux). Columns as above: [transition, boundary(step 0), boundary(step n-1)] x deg. */
int32_t aero_eval_constraints_air(aero_ctx* ctx, const aero_matrix* trace_lde, const aero_matrix* aux_lde, const aero_fib_air* air,
                                  uint32_t log_blowup, const uint64_t* results, const uint64_t* rands, const uint64_t* coeffs,
                                  uint8_t field_extension, uint32_t fragment_offset, uint32_t num_fragments, uint64_t* out_cols,
                                  uint64_t* frag_index_out);
/* The AIR-specific half of the fork's `commit_to_trace_and_validate` (proving_worker.rs:323-332): the auxiliary columns of the built-in
 * stand-in AIR from the main trace and the drawn elements (p_c(0) = 1, p_c(i+1) = p_c(i) * (r_(c mod R) + main_(c mod W)(i))^(D-1)),
 * as an (aux_width * deg) x trace_len matrix of component columns - input to aero_interpolate_columns / aero_evaluate_columns_over
 * / aero_merkle_commit_rows like the main segment. */
int32_t aero_aux_columns_fib(aero_ctx* ctx, const aero_matrix* trace, const aero_fib_air* air, const uint64_t* rands, uint8_t field_extension,
                             aero_matrix** aux_out);

/* ---- composition polynomial, DEEP composition (the stages inside the fork's `prove_after_constraint_eval`,
 *      proving_worker.rs:344-352; bodies in winter-prover 0.4) --------------------------------------------------------------- */
/* `ConstraintEvaluationTable::into_poly` -> `CompositionPoly` for FibAir: numer_cols = the 3*deg numerator columns of
 * aero_eval_constraints_fib over the WHOLE constraint domain (ce_n = 2 * 2^log_n values each, host); every column is divided
 * by its divisor ((x^n - 1)/(x - w^(n-1)), x - 1, x - w^(n-1)), the columns are summed, interpolated over the coset and
 * split into the 2 column polynomials H(x) = H_0(x^2) + x H_1(x^2). Output: matrix of 2*deg columns x 2^log_n rows in the
 * backend's internal polynomial form (input to aero_evaluate_columns_over / aero_poly_eval); column order [component][c]. */
int32_t aero_composition_poly_fib(aero_ctx* ctx, const uint64_t* numer_cols, uint32_t log_n, uint8_t field_extension,
                                  aero_matrix** comp_polys);
/* Same for C = 2, 4 or 8 composition columns (the AIR with an auxiliary constraint of degree up to 8): numer_cols over the
 * C * 2^log_n-point constraint domain. Column order of the result: column d * C + q = component d of composition column
 * bitrev(q) (the identity for C = 2) - the order aero_evaluate_columns_over / aero_poly_eval hand back as well. */
int32_t aero_composition_poly_air(aero_ctx* ctx, const uint64_t* numer_cols, uint32_t log_n, uint32_t num_columns, uint8_t field_extension,
                                  aero_matrix** comp_polys);
/* DEEP composition over the LDE domain (winter-prover `DeepCompositionPoly`; verifier-side mirror
 * src/stark_verifier/composer.cairo:48-316). trace_lde: W columns; comp_lde: C*deg columns, column c*deg + d = component d of
 * composition column c (the order in which rows are hashed). All field elements are `deg` u64 each (deg = 1, or 2 for the
 * quadratic extension): z; ood_frame = current row (W) then next row (W) of trace evaluations at z and z*g; ood_evals = C
 * values H_c(z^C); coeffs in draw order = 3 per trace column (alpha, beta, gamma), one per composition column, lambda, mu.
 * Output: deg columns x N rows (component columns of the DEEP evaluations, natural order) = FRI layer 0. */
int32_t aero_deep_compose(aero_ctx* ctx, const aero_matrix* trace_lde, const aero_matrix* comp_lde, uint32_t log_blowup,
                          uint8_t field_extension, const uint64_t* z, const uint64_t* ood_frame, const uint64_t* ood_evals,
                          const uint64_t* coeffs, aero_matrix** deep_evals);

/* ---- FRI and grinding ------------------------------------------------------------------------------------------------- */
/* `FriProver::build_layers(channel, evaluations)` (winter-fri 0.4; verifier mirror fri_verifier.cairo:56-82,243-340): per layer
 * transpose to rows of `fold` values, hash, commit, reseed the coin with the root, draw alpha, fold; the remainder is
 * committed the same way. evals = deg columns x N rows (aero_deep_compose output). seed_in = coin seed before the first FRI
 * commitment (draws never change the seed; reseeding resets the counter). roots_out receives (layers + 1) * 32 bytes;
 * seed_out the coin seed after the remainder commitment. The handle keeps every layer for aero_fri_open. */
typedef struct aero_fri aero_fri;
int32_t aero_fri_build_layers(aero_ctx* ctx, const aero_matrix* evals, const aero_proof_options* options, const uint8_t seed_in[32],
                              uint8_t* roots_out, size_t roots_cap, uint32_t* num_roots, uint8_t seed_out[32], aero_fri** out);
/* `FriProver::build_proof(positions)`: the serialised FriProof exactly as it sits in StarkProof::to_bytes — u8 #layers, per
 * layer u32 len + values, u32 len + BatchMerkleProof nodes, then u16 len + remainder, u8 0. positions are
 * LDE-domain query positions (folded per layer, duplicates dropped keeping the first: miden-to-cairo-parser/src/lib.rs:421-436). */
int32_t aero_fri_open(aero_ctx* ctx, const aero_fri* fri, const uint64_t* positions, uint32_t k, uint8_t** out, size_t* out_len);
void aero_fri_free(aero_ctx* ctx, aero_fri* fri);

/* One FRI layer fold over the base field (winter-fri `apply_drp`, mirrored by src/stark_verifier/fri/
 * fri_verifier.cairo:305-315): values = dom evaluations in natural order over 7 * <w_dom>; out = dom/fold values. */
int32_t aero_fri_fold(aero_ctx* ctx, const uint64_t* values, uint64_t dom, uint32_t fold, uint64_t alpha, uint64_t* out);
/* Replaces `channel.grind_query_seed()`: smallest nonce >= 1 such that BLAKE2s(seed || LE64(nonce)) has at least `bits`
 * leading zero bits counted MSB-first from byte 0 (src/stark_verifier/crypto/random.cairo:282-316). */
int32_t aero_grind(aero_ctx* ctx, const uint8_t seed[32], uint32_t bits, uint64_t* nonce_out);

/* ---- whole proof ---------------------------------------------------------------------------------------------------------- */
/* Replaces `Prover::prove(trace)` for the built-in FibAir (proving_worker.rs:465-467; miden-proof-generator/src/
 * main.rs:31) followed by `proof.to_bytes()` (main.rs:38). `trace` is already resident on the device. *proof is
 * malloc'd (aero_free). pub_out receives the width/2 public inputs (the asserted results). */
int32_t aero_prove_fib(aero_ctx* ctx, const aero_matrix* trace, const aero_proof_options* options, uint8_t** proof,
                       size_t* proof_len, uint64_t* pub_out);
/* Same with the trace in HOST memory — the reference's actual hand-over: `Prover::prove(trace)` receives the ExecutionTrace
 * the VM left in host memory (proving_worker.rs:140,465-467; miden-proof-generator/src/main.rs:31). The host-to-device copy is
 * enqueued on the context's stream in front of the proof; the canonical-form check of the elements runs behind it without a
 * stream synchronisation of its own (a trace with an element >= p yields AERO_E_BAD_ARG and no proof). With the buffer pinned
 * (aero_host_register) the copy is an asynchronous DMA that overlaps other contexts' kernels. */
int32_t aero_prove_fib_host(aero_ctx* ctx, const uint64_t* trace_col_major, uint32_t width, uint32_t log_n,
                            const aero_proof_options* options, uint8_t** proof, size_t* proof_len, uint64_t* pub_out);
/* Pin / unpin a caller-owned host buffer (e.g. the Vec<Felt> behind winter's `Matrix<Felt>` columns, utils.rs:235-236) so that
 * trace hand-overs from it are asynchronous DMA transfers. On failure the text is available from aero_last_error(NULL). */
int32_t aero_host_register(void* p, size_t bytes);
int32_t aero_host_unregister(void* p);
/* Pinned host memory allocated by the library's HIP runtime (`hipHostMalloc`) - the alternative to registering memory the host
 * already owns: a host that can choose where its trace lives (the Vec<Felt> behind `Matrix<Felt>`, utils.rs:235-236) fills such a
 * buffer and hands it to the *_host entry points. aero_host_free(NULL) is a no-op. Text on failure: aero_last_error(NULL). */
int32_t aero_host_alloc(size_t bytes, void** out);
int32_t aero_host_free(void* p);
/* ---- host placement on a multi-socket node -------------------------------------------------------------------------------- */
/* The reference sizes its worker pool by `navigator.hardwareConcurrency` and has no notion of where a worker runs
 * (aero-sdk/miden-wasm/src/pool.rs:28-45,105-124). A rank process that drives ONE GPU of a two-socket 8-GPU node wants its worker
 * threads and its pinned trace buffers on the GPU's own NUMA node; a process that has touched the GPU must not re-exec under
 * numactl, so the library does it from inside:
 *   aero_numa_device_node ... NUMA node of the device (sysfs: /sys/bus/pci/devices/<bdf>/numa_node), -1 = unknown / AERO_NUMA=0
 *   aero_numa_bind_thread ... binds the CALLING thread to that node's CPUs (those the process is allowed to use); *n_cpus_out = how
 *                             many it now runs on, 0 = left alone. aero_pool_create does this for its worker threads by itself.
 *   aero_host_alloc_near .... aero_host_alloc with the pages placed on the device's node (falls back to aero_host_alloc)
 *   aero_numa_query ......... the sysfs parsing alone, against a sysfs tree rooted at `sysfs_root_dir` ("/sys"): node of the PCI
 *                             device `pci_bdf` and the CPUs of that node (cpus_out may be NULL; *n_cpus_out = their number)
 *   aero_numa_parse_cpulist . "0-15,32-47" -> CPU numbers (the kernel's cpulist format); AERO_E_BAD_ARG on malformed text */
int32_t aero_numa_device_node(int32_t device_id, int32_t* node_out);
int32_t aero_numa_bind_thread(int32_t device_id, uint32_t* n_cpus_out);
int32_t aero_host_alloc_near(size_t bytes, int32_t device_id, void** out);
int32_t aero_numa_query(const char* sysfs_root_dir, const char* pci_bdf, int32_t* node_out, int32_t* cpus_out, uint32_t cpus_cap, uint32_t* n_cpus_out);
int32_t aero_numa_parse_cpulist(const char* text, int32_t* cpus_out, uint32_t cpus_cap, uint32_t* n_cpus_out);
/* ---- one proof sharded over the GPUs of a node ------------------------------------------------------------------------------ */
/* The exchange steps of a sharded proof, supplied by the host (one process per GPU; torch.distributed over RCCL in this
 * repo's harness, `ncclSend/Recv`-style bindings from Rust). The reference has no multi-device prover; its parallel
 * decomposition is the worker pool of aero-sdk/miden-wasm/src/proving_worker.rs:276-321 (row-hash batches) and :374-437
 * (constraint fragments), whose gather step these callbacks replace. All buffers are DEVICE pointers on the context's GPU.
 * Default contract: each callback must return only when `recv` (or `buf`) is complete and safe to read from any stream of the
 * device, and the library synchronises its own stream before calling. With AERO_COMM_STREAM_ORDERED in `flags` the callbacks
 * instead ENQUEUE the exchange on the context's own stream (the native RCCL communicator below does): the library then neither
 * synchronises before the call nor assumes completion after it - ordering is the stream's. Return 0 on success.
 *   all_to_all ........ send = world chunks of `bytes` (chunk r goes to rank r); recv = world chunks (chunk r came from rank r)
 *   all_gather ........ send = `bytes`; recv = world chunks of `bytes` in rank order
 *   all_reduce_sum_u64  in-place wrapping sum of `count` u64 over all ranks
 *   send_recv ......... (optional) one chunk to one peer, one chunk from another - see the struct
 * min_peer_digests: a FRI layer stays sharded while every rank still sends at least this many leaf digests to every peer;
 * smaller layers are all-gathered once and finished redundantly on every rank (0 = default 2048 = 64 KiB messages). */
typedef struct aero_comm {
    int32_t rank, world; /* world = power of two, <= blowup factor */
    void* user;
    int32_t (*all_to_all)(void* user, const void* send, void* recv, uint64_t bytes);
    int32_t (*all_gather)(void* user, const void* send, void* recv, uint64_t bytes);
    int32_t (*all_reduce_sum_u64)(void* user, void* buf, uint64_t count);
    uint32_t min_peer_digests;
    uint32_t flags;      /* 0, or AERO_COMM_STREAM_ORDERED */
    /* optional (NULL: the library falls back to all_gather): `bytes` from `send` go to rank send_to while `bytes` from rank recv_from
     * arrive in `recv` - every rank calls it with the same shift (send_to - rank = rank - recv_from mod world). Used where a rank
     * needs the data of one or two peers only (the constraint evaluations of the cosets that make up its constraint domain). */
    int32_t (*send_recv)(void* user, const void* send, int32_t send_to, void* recv, int32_t recv_from, uint64_t bytes);
} aero_comm;
#define AERO_COMM_STREAM_ORDERED 1u

/* Native communicator: RCCL over xGMI, one process per GPU, created next to the context (SURVEY 8b asked for the communicator
 * at context creation; with one context = one GPU = one process it is its own object so that single-GPU users never touch
 * RCCL). Usage on every rank: rank 0 calls aero_rccl_unique_id and distributes the 128 bytes out of band (a file, a socket, the
 * launcher's store); all ranks call aero_rccl_create (collective: `ncclCommInitRank`), then aero_rccl_comm to obtain the
 * aero_comm to pass to aero_prove_fib_sharded / aero_prove_fib_air. The exchanges are `ncclSend`/`ncclRecv` groups (digest
 * all-to-all), `ncclAllGather` (subtree roots) and one `ncclAllReduce` (openings), all enqueued on the context's stream.
 * librccl is bound with dlopen at first use: AERO_E_COMM when it is not installed. The context must outlive the communicator.
 * (The reference has no multi-device prover; its gather step is the worker-pool fan-in of proving_worker.rs:302-310,428-437.) */
#define AERO_RCCL_ID_BYTES 128
typedef struct aero_rccl aero_rccl;
/* AERO_OK when librccl can be bound in this process (nothing is created), AERO_E_COMM + aero_rccl_last_error(NULL) otherwise: the
 * pre-flight every rank runs before anybody enters the collective `ncclCommInitRank` (a rank without RCCL must make all ranks give
 * up instead of leaving its peers in the bootstrap). (No counterpart in the reference: its prover is single-device, the fan-in of
 * its worker pool is proving_worker.rs:302-310,428-437.) */
int32_t aero_rccl_available(void);
int32_t aero_rccl_unique_id(uint8_t id_out[AERO_RCCL_ID_BYTES]);
int32_t aero_rccl_create(aero_ctx* ctx, int32_t rank, int32_t world, const uint8_t id[AERO_RCCL_ID_BYTES], aero_rccl** out);
int32_t aero_rccl_comm(aero_rccl* r, uint32_t min_peer_digests, aero_comm* out);
/* out = {all_to_all calls, all_gather calls, all_reduce calls, bytes sent by this rank} since creation */
int32_t aero_rccl_stats(const aero_rccl* r, uint64_t out[4]);
/* What the communicator itself reports: out = {ranks RCCL counts in it (ncclCommCount), this rank as RCCL numbers it (ncclCommUserRank),
 * the device RCCL bound (ncclCommCuDevice), the world the caller asked for}; -1 where the bound librccl lacks the query. The first thing a
 * first run on a multi-GPU node prints (tools/first_contact.sh). (No counterpart in the reference: single-device prover.) */
int32_t aero_rccl_info(const aero_rccl* r, int32_t out[4]);
const char* aero_rccl_last_error(const aero_rccl* r);   /* r may be NULL for errors of unique_id / create */
void aero_rccl_destroy(aero_rccl* r);
/* `Prover::prove` + `to_bytes` for ONE trace proven cooperatively by comm->world GPUs (BASELINE config 4). Rank k owns
 * the LDE rows j = k (mod world), i.e. the coset 7 w_N^k <w_(N/world)>: LDEs, row hashing, constraint evaluation, DEEP
 * and FRI folds are local; per commitment the ranks exchange leaf digests (all_to_all) so that each builds one
 * contiguous Merkle subtree, and all-gather the subtree roots — the only collective on the transcript's critical path.
 * Every rank passes the whole trace (device) and receives the identical proof bytes, which are byte-identical to
 * aero_prove_fib's. */
int32_t aero_prove_fib_sharded(aero_ctx* ctx, const aero_comm* comm, const aero_matrix* trace, const aero_proof_options* options,
                               uint8_t** proof, size_t* proof_len, uint64_t* pub_out);
/* The main segment's commitment alone: interpolate, extend this rank's coset, exchange rows or digests, build the subtree, all-gather
 * the subtree roots - everything aero_prove_fib_sharded does up to the first reseed of the coin, then stop. The first half of the
 * reference's fork-only split `commit_to_trace_and_validate` (aero-sdk/miden-wasm/src/proving_worker.rs:323-332; the tree it is handed
 * is built at :283-321). root_out = the trace commitment (bytes 24..55 of the proof); subtree_roots_out (may be NULL) = comm->world x 32
 * bytes, subtree r = the leaves [r N / world, (r + 1) N / world). comm == NULL or world == 1: one GPU (subtree_roots_out = the root).
 * Used by the exchange stress loop (tests/shard_stress_worker.py): thousands of exchanges per minute against a known root. */
int32_t aero_commit_trace_sharded(aero_ctx* ctx, const aero_comm* comm, const aero_matrix* trace, const aero_proof_options* options,
                                  uint8_t root_out[32], uint8_t* subtree_roots_out);

/* The same with the trace in HOST memory - the hand-over the metric is defined on (`Prover::prove(trace)` receives a host
 * ExecutionTrace, proving_worker.rs:465-467): every rank is given the same column-major host buffer (one process per GPU on one
 * node: e.g. a shared mapping) and copies only ITS width / world columns to its GPU (1 / world of the PCIe traffic per rank),
 * interpolates them, and the coefficients are all-gathered over the GPU links; with width < world every rank copies the whole
 * trace. air may be NULL (plain FibAir); with an auxiliary segment the evaluations are all-gathered as well (its builders read the
 * main segment). comm == NULL or world == 1 is aero_prove_fib_air_host. Same bytes on every rank as the single-GPU proof. */
int32_t aero_prove_fib_sharded_host(aero_ctx* ctx, const aero_comm* comm, const uint64_t* trace_col_major, uint32_t width, uint32_t log_n,
                                    const aero_fib_air* air, const aero_proof_options* options, uint8_t** proof, size_t* proof_len,
                                    uint64_t* pub_out);

/* In-process communicator: `world` ranks = `world` contexts of THIS process (one host thread each), on one GPU or on several
 * (peer access) - no RCCL, no second process. Exchanges are device-to-device copies enqueued on the ranks' own streams and ordered
 * by events (AERO_COMM_STREAM_ORDERED); the host threads only rendezvous. For hosts that drive several GPUs from one process, and
 * for exercising the sharded prover's stream-ordered contract on a one-GPU box.
 *   aero_local_group_create ... world = power of two
 *   aero_local_group_comm ..... binds rank `rank` to its context and fills the aero_comm to pass to the sharded entry points
 *   aero_local_group_abort .... releases ranks waiting at a rendezvous (call when a rank failed outside an exchange)
 *   aero_local_group_stats .... {all_to_all calls, all_gather calls, all_reduce calls, bytes sent} of a rank
 * aero_prove_fib_sharded_local: the whole thing in one call - a context and a thread per rank on device_ids[r] (ids may repeat),
 * trace in host memory (aero_prove_fib_sharded_host per rank); proofs[r] / proof_lens[r] per rank (aero_free each), rank_ms and
 * bytes_sent (may be NULL) per rank. */
typedef struct aero_local_group aero_local_group;
int32_t aero_local_group_create(uint32_t world, aero_local_group** out);
int32_t aero_local_group_comm(aero_local_group* group, aero_ctx* ctx, int32_t rank, uint32_t min_peer_digests, aero_comm* out);
int32_t aero_local_group_stats(const aero_local_group* group, int32_t rank, uint64_t out[4]);
const char* aero_local_group_last_error(const aero_local_group* group, int32_t rank);
void aero_local_group_abort(aero_local_group* group);
void aero_local_group_destroy(aero_local_group* group);
int32_t aero_prove_fib_sharded_local(const int32_t* device_ids, uint32_t world, const uint64_t* trace_col_major, uint32_t width, uint32_t log_n,
                                     const aero_fib_air* air, const aero_proof_options* options, uint32_t min_peer_digests, uint8_t** proofs,
                                     size_t* proof_lens, uint64_t* pub_out, double* rank_ms, uint64_t* bytes_sent, char* err, size_t err_cap);

/* ---- auxiliary trace segment ---------------------------------------------------------------------------------------------------- */
/* `Prover::prove` for FibAir extended by ONE auxiliary segment, the step the fork's `commit_to_trace_and_validate`
 * (proving_worker.rs:323-332) performs per aux segment: after the main commitment `aux_rands` elements are drawn from the
 * coin, the AIR builds `aux_width` columns over E from them, the columns are interpolated / extended / committed like the
 * main segment (second trace root, second trace-query block, aux columns in the OOD frame and the DEEP composition:
 * src/stark_verifier/stark_verifier.cairo:117-130,266-294, composer.cairo:196-316). The Miden AIR that defines the real
 * columns is absent from the reference mount; the built-in stand-in is the multiset-check shape: p_c(0) = 1,
 * p_c(i+1) = p_c(i) * (r_(c mod aux_rands) + main_(c mod width)(i)), one degree-2 transition constraint and the
 * assertion p_c(0) = 1 per column. comm may be NULL (single GPU) or describe a sharded run (see above).
 * aux_width = 0 is exactly aero_prove_fib / aero_prove_fib_sharded. */
int32_t aero_prove_fib_aux(aero_ctx* ctx, const aero_comm* comm, const aero_matrix* trace, uint32_t aux_width, uint32_t aux_rands,
                           const aero_proof_options* options, uint8_t** proof, size_t* proof_len, uint64_t* pub_out);

/* Same with the degree of the auxiliary transition constraint as a parameter: p_c(i+1) = p_c(i) * (r + main)^(aux_degree - 1),
 * aux_degree in [2, 8]. The constraint-evaluation blowup and the number of composition columns follow Winterfell's rule
 * max(next_pow2(max constraint degree), 2): 2 / 4 / 8 for degree 2 / 3-4 / 5-8 — 8 is the shape of the reference's golden
 * Miden proof (proofs/fib.bin carries 8 composition columns; stark_verifier.cairo:166-176). blowup_factor must be >= that. */
int32_t aero_prove_fib_air(aero_ctx* ctx, const aero_comm* comm, const aero_matrix* trace, const aero_fib_air* air,
                           const aero_proof_options* options, uint8_t** proof, size_t* proof_len, uint64_t* pub_out);
/* aero_prove_fib_host with an AIR descriptor (air may be NULL = plain FibAir). */
int32_t aero_prove_fib_air_host(aero_ctx* ctx, const uint64_t* trace_col_major, uint32_t width, uint32_t log_n, const aero_fib_air* air,
                                const aero_proof_options* options, uint8_t** proof, size_t* proof_len, uint64_t* pub_out);

/* ---- several proofs in flight on one GPU --------------------------------------------------------------------------------------- */
/* A pool = `slots` contexts on one device, each driven by its own host thread inside the library. One proof alone cannot fill
 * an MI355X (its tree tops and transcript round trips are latency-bound); with ~8 independent proofs in flight the GPU stays
 * busy. The reference's analogue is its worker pool (aero-sdk/miden-wasm/src/pool.rs:28-45), which
 * spreads ONE proof's row-hash / constraint batches over web workers; here whole proofs are the unit.
 *   aero_pool_ctx ........ the context of slot i (for aero_trace_upload etc.; do not destroy it, and do not use it while a
 *                          aero_pool_prove_fib call is running)
 *   aero_pool_prove_fib .. `count` <= slots proofs at once: traces[i] must be resident on slot i's context; proofs[i] is
 *                          malloc'd (aero_free), pubs receives width/2 values per proof back to back. air may be NULL (plain
 *                          FibAir). `rounds` > 1 repeats the batch that many times back to back inside the library (every
 *                          slot proves its trace `rounds` times; the last proof of each slot is returned) - for throughput
 *                          measurements without host re-entry. Returns the first non-zero status of any slot. */
typedef struct aero_pool aero_pool;
int32_t aero_pool_create(int32_t device_id, uint32_t slots, aero_pool** out);
void aero_pool_destroy(aero_pool* pool);
uint32_t aero_pool_slots(const aero_pool* pool);
/* aero_ctx_set_self_verify for every slot's context (call between batches). */
int32_t aero_pool_set_self_verify(aero_pool* pool, int32_t mode);
aero_ctx* aero_pool_ctx(aero_pool* pool, uint32_t slot);
/* A QUEUE of different host traces of one shape (column-major width x 2^log_n each, pinned or pageable): trace t is dealt to slot
 * t mod slots - the reference's pool deals its batches the same way (`batch_idx % concurrency`, aero-sdk/miden-wasm/src/pool.rs:105-124) -,
 * every slot proves its share in order, copying its next trace while it proves the current one (traces of >= 32 MiB), and EVERY proof
 * comes back: proofs[t] (malloc'd: aero_free), proof_lens[t], pubs + t * width / 2 (pubs may be NULL). All or nothing: on a non-zero
 * status every proofs[t] is NULL. n_traces may be smaller or (much) larger than the number of slots. */
int32_t aero_pool_prove_fib_queue(aero_pool* pool, const uint64_t* const* host_traces, uint32_t n_traces, uint32_t width, uint32_t log_n,
                                  const aero_fib_air* air, const aero_proof_options* options, uint8_t** proofs, size_t* proof_lens, uint64_t* pubs);
/* where the pool's worker threads ended up: node_out = NUMA node of the device (-1 unknown), pinned_out = workers bound to its CPUs */
int32_t aero_pool_placement(const aero_pool* pool, int32_t* node_out, uint32_t* pinned_out);
int32_t aero_pool_prove_fib(aero_pool* pool, const aero_matrix* const* traces, uint32_t count, const aero_fib_air* air,
                            const aero_proof_options* options, uint32_t rounds, uint8_t** proofs, size_t* proof_lens, uint64_t* pubs);
/* Same with the traces in HOST memory (host_traces[i] = column-major width x 2^log_n, ideally pinned): every proof of every
 * round starts with the host-to-device copy of its trace on the slot's own stream, so one slot's transfer overlaps the other
 * slots' kernels. This is the "trace in host memory -> proof bytes" job the reference's `prove` performs per call. */
int32_t aero_pool_prove_fib_host(aero_pool* pool, const uint64_t* const* host_traces, uint32_t width, uint32_t log_n, uint32_t count,
                                 const aero_fib_air* air, const aero_proof_options* options, uint32_t rounds, uint8_t** proofs,
                                 size_t* proof_lens, uint64_t* pubs);

/* ---- verification (host only, no GPU) ------------------------------------------------------------------------------------------ */
/* The counterpart of `winter_verifier::verify` for this backend's proofs, written against the reference's in-tree verifier
 * (src/stark_verifier/stark_verifier.cairo:105-304 and the files it calls): transcript, proof of work, every Merkle opening, DEEP
 * composition at every query, FRI layer consistency, remainder commitment and degree, and - which the Cairo code leaves commented
 * out (:151-159,183-187) - the out-of-domain constraint consistency check for the built-in FibAir.
 *
 * A proof declares its own options and trace length; what the CALLER accepts is the policy (NULL = {96, 0, 0, 0, 0}):
 *   min_query_security_bits  reject when num_queries * log2(blowup) + grinding is below this (with_96_bit_security() = 97)
 *   expected_log_n           the trace length the statement is about (FibAir: "the 2^k-th term is pub"); 0 = take the proof's
 *   allow_unknown_air        non-zero: `air` may be NULL - everything except the OOD constraint check, exactly what the Cairo
 *                            verifier does (e.g. the golden Miden proof proofs/fib.bin; pub_elements = the coin-seed elements,
 *                            for Miden proofs program hash || stack inputs || outputs, crypto/random.cairo:254-280). Never use
 *                            this to accept FibAir proofs: any low-degree commitment passes without that check.
 *   cairo_compat             non-zero: additionally require the shape src/stark_verifier hard-codes (72 + 9 columns, 8
 *                            composition columns, 27 queries, blowup 8, FRI folding factor 8, no extension field)
 *   require_options/options  non-zero: the proof's 7 option bytes must equal `options`
 *   min_conjectured_security_bits  non-zero: reject when min(query bits, 64 * extension degree - log2(LDE domain size)) is below
 *                            this - Winterfell's conjectured-security estimate (what aero_proof_security_bits reports); a base-field
 *                            proof of a 2^20-row trace has 41 field bits however many queries it carries
 * air: the built-in FibAir descriptor (aux fields must match the proof); pub_elements = the width/2 results.
 * Returns AERO_OK, AERO_E_VERIFY (rejected; reason in err) or AERO_E_BAD_ARG (incl. air == NULL without allow_unknown_air). */
typedef struct aero_verify_policy {
    uint32_t min_query_security_bits;
    uint32_t expected_log_n;
    uint32_t allow_unknown_air;
    uint32_t cairo_compat;
    uint32_t require_options;
    aero_proof_options options;
    uint32_t min_conjectured_security_bits;
} aero_verify_policy;
int32_t aero_verify_fib(const uint8_t* proof, size_t proof_len, const uint64_t* pub_elements, uint32_t n_pub, const aero_fib_air* air,
                        const aero_verify_policy* policy, char* err, size_t err_cap);
/* The two terms of Winterfell's conjectured-security estimate for a proof's self-declared parameters: query_bits =
 * num_queries * log2(blowup) + grinding; field_bits = 64 * extension degree - log2(LDE domain size). */
int32_t aero_proof_security_bits(const uint8_t* proof, size_t proof_len, uint32_t* query_bits, uint32_t* field_bits);

/* bincode ProofData{input_bytes, proof_bytes} = u64 len || inputs || u64 len || proof
 * (miden-proof-generator/src/lib.rs:1-6, main.rs:49-51). */
int32_t aero_proof_container(const uint8_t* inputs, size_t inputs_len, const uint8_t* proof, size_t proof_len, uint8_t** out,
                             size_t* out_len);

/* ---- re-encoding a proof for the reference's downstream consumers (host only, no GPU) ------------------------------------------ */
/* Cairo-memory image: the JSON array `stark_parser <file> <sub-command>` prints (miden-to-cairo-parser/src/main.rs:42-113) and
 * src/stark_verifier/utils.py:10-23 loads into the Cairo VM - values as hex strings (`0x...`), pointers as decimal offsets into the
 * same array (segment model and formats: memory.rs:31-150; field order: lib.rs:41-260,395-470).
 *   AERO_CAIRO_PROOF               the parsed StarkProof (context, commitments, OOD frame, pow nonce, queried states, remainder)
 *   AERO_CAIRO_PUBLIC_INPUTS       Miden `PublicInputs` from the container's input bytes (proof may be NULL)
 *   AERO_CAIRO_TRACE_QUERIES       per trace segment, per index: the Merkle authentication path (`indexes` = the query positions
 *   AERO_CAIRO_CONSTRAINT_QUERIES  the proof was opened at, in draw order; the paths are checked against the commitments)
 *   AERO_CAIRO_FRI_QUERIES         per FRI layer, per folded index: path + the row's folding_factor values
 * Like the reference's encoder this needs a proof WITH an auxiliary trace segment over the base field (lib.rs:138,147 unwrap the
 * auxiliary frame / states): otherwise AERO_E_UNSUPPORTED. *json_out is malloc'd (aero_free), NUL-terminated, without the
 * trailing newline `println!` adds. Errors: AERO_E_VERIFY (malformed / inconsistent proof), text in err. */
#define AERO_CAIRO_PROOF 0u
#define AERO_CAIRO_PUBLIC_INPUTS 1u
#define AERO_CAIRO_TRACE_QUERIES 2u
#define AERO_CAIRO_CONSTRAINT_QUERIES 3u
#define AERO_CAIRO_FRI_QUERIES 4u
int32_t aero_cairo_memory(uint32_t what, const uint8_t* proof, size_t proof_len, const uint8_t* input_bytes, size_t input_len,
                          const uint64_t* indexes, uint32_t n_indexes, char** json_out, size_t* json_len, char* err, size_t err_cap);
/* protobuf `sdk.StarkProof` (aero-sdk/proto/stark_proof.proto:13-30 and the files it imports; contents after
 * `StarkProof::into_sdk`, aero-sdk/miden-wasm/src/convert/convert_proof.rs:13-307: parsed tables, per-segment and per-layer
 * BatchMerkleProofs with explicit leaves and depth) and `sdk.MidenPublicInputs` (proto/miden_vm.proto:14-18) - the bytes the
 * reference's browser prover hands to the SDK (proving_worker.rs:210-222). proto3 wire format as prost 0.11 writes it. Base
 * field + Blake2s only (the reference has no enum values for anything else): otherwise AERO_E_UNSUPPORTED. */
int32_t aero_proof_to_protobuf(const uint8_t* proof, size_t proof_len, uint8_t** out, size_t* out_len, char* err, size_t err_cap);
int32_t aero_miden_public_inputs_to_protobuf(const uint8_t* input_bytes, size_t input_len, uint8_t** out, size_t* out_len, char* err,
                                             size_t err_cap);

/* ---- the reference's worker seam at the message level --------------------------------------------------------------------------- */
/* The browser SDK's proving worker off-loads row hashing and constraint evaluation to web workers with bincode 1.3 messages
 * (aero-sdk/miden-wasm/src/utils.rs:442-450 `to_uint8array` / `from_uint8array`; pool.rs:84-125 posts them,
 * proving_worker.rs:154-159,428-437 merges the answers). These entry points take the very bytes the pool posts and return the bytes
 * the proving worker expects back (layouts restated in aero_amd/csrc/worker_messages.hpp); *result is malloc'd (aero_free).
 *
 * aero_worker_hash_rows: HashingWorkItem { data: Vec<Vec<Felt>>, batch_idx } (utils.rs:358-362) -> HashingResult { batch_idx,
 *   hashes: Vec<[u8; 32]> } (utils.rs:411-415) = `blake2_hash_elements` (hashing_worker.rs:12-26): Blake2s_256::hash_elements of
 *   every row, in row order. The rows are hashed where they lie in the message (one copy of the bytes to the device, one lane
 *   per row); they may differ in length, a row without elements hashes to BLAKE2s of the empty string like hash_elements(&[]),
 *   a value >= p is reduced like Felt::new does.
 * aero_worker_eval_constraints (declared in aero_air.h: it takes the AIR as a program): ConstraintComputeWorkItem { trace_info,
 *   public_inputs, proof_options, aux_rand_elements, constraint_coeffs, trace_lde_wrapper, computation_fragment } (utils.rs:302-347)
 *   -> ConstraintComputeResult { frag_index, frag_num, constraint_evaluations } (utils.rs:417-422) = `constraint_compute`
 *   (constraints_worker.rs:14-79): one numerator column per divisor, for the fragment.
 * aero_prover_output: what the proving worker hands back to the SDK (proving_worker.rs:205-222, utils.rs:424-430): bincode
 *   ProverOutput { proof, program_outputs, public_inputs } - the protobuf encodings of sdk.StarkProof, sdk.MidenProgramOutputs
 *   and sdk.MidenPublicInputs - from proof bytes and the container's input bytes. */
/* Host-only look at a message (no GPU): validates the layout (AERO_E_BAD_ARG with the reason in err otherwise) and reports
 *   AERO_MSG_HASHING_WORK_ITEM:    out = {rows, batch_idx, shortest row, longest row, total elements}
 *   AERO_MSG_CONSTRAINT_WORK_ITEM: out = {main width, aux width, aux rands, trace length, blowup, fragment_offset, num_fragments,
 *                                         number of (alpha, beta) pairs} */
#define AERO_MSG_HASHING_WORK_ITEM 0u
#define AERO_MSG_CONSTRAINT_WORK_ITEM 1u
int32_t aero_worker_message_info(uint32_t kind, const uint8_t* msg, size_t len, uint64_t out[8], char* err, size_t err_cap);
int32_t aero_worker_hash_rows(aero_ctx* ctx, const uint8_t* work_item, size_t work_item_len, uint8_t** result, size_t* result_len);
int32_t aero_prover_output(const uint8_t* proof, size_t proof_len, const uint8_t* input_bytes, size_t input_len, uint8_t** out,
                           size_t* out_len, char* err, size_t err_cap);
/* ---- instrumentation --------------------------------------------------------------------------------------------------------- */
/* Per-stage wall-clock of the last aero_prove_* (ms; adds one stream sync per stage when enabled). Order:
 * interpolate, lde, trace_commit, constraints, composition, comp_commit, ood, deep, fri, grind, queries, total
 * (stage names after the console labels of proving_worker.rs:125-172). */
int32_t aero_set_stage_timing(aero_ctx* ctx, int32_t enable);
int32_t aero_last_stage_ms(const aero_ctx* ctx, double out[12]);
/* Per-kernel HIP-event timing on the context's own stream. When enabled, kernel launches (all of them, or only the
 * kernel named `only_kernel` when that is non-NULL) are bracketed by events; aero_kernel_timing_report writes
 * "name calls total_ms algorithmic_bytes\n" lines (sorted by total time) and resets the counters. algorithmic_bytes =
 * every input element read once + every output element written once, summed over the bracketed launches. */
int32_t aero_set_kernel_timing(aero_ctx* ctx, int32_t enable, const char* only_kernel);
int32_t aero_kernel_timing_report(aero_ctx* ctx, char* buf, size_t cap);
/* Device memory currently held / peak (bytes). */
int32_t aero_memory_stats(const aero_ctx* ctx, uint64_t* in_use, uint64_t* peak);

#ifdef __cplusplus
}
#endif
#endif /* AERO_STARK_H */
