/*
 * aero_air.h — AIR-as-data: a constraint system handed to libaero_stark.so as BYTES, and the entry points that prove, evaluate
 * and verify against it.
 *
 * What this replaces in the reference: the `Air` trait object the constraint seam is generic over —
 *   `ProcessorAir::new(trace_info, public_inputs, proof_options)`          aero-sdk/miden-wasm/src/constraints_worker.rs:32-36,
 *                                                                           proving_worker.rs:255-259
 *   `ConstraintEvaluator::new(&air, aux_rand_elements, &constraint_coeffs)` constraints_worker.rs:38-43, proving_worker.rs:374-381
 *   `evaluator.evaluate_fragment(&trace_lde, &domain, &mut fragment)`       constraints_worker.rs:56-59
 *   `Trace::build_aux_segment` inside `commit_to_trace_and_validate`        proving_worker.rs:323-332
 * A Rust `Air` is code (`evaluate_transition`, `get_assertions`, `get_periodic_column_values`, the aux variants); code cannot
 * cross a C ABI onto a GPU, so the constraint set crosses it as a program: a flat SSA expression list over the frame
 * (current / next row of the main and auxiliary segments), periodic columns, constants, public inputs and the auxiliary random
 * elements, with one root per transition constraint and its declared degree, plus the assertions. A host pass inside the
 * library groups the constraints by evaluation degree, derives divisors and degree adjustments by Winterfell 0.4's rules
 * (mirrored for the verifier in src/stark_verifier/air/transitions/evaluator.cairo:79-86,131-150,216-218), allocates registers,
 * and a device interpreter runs the result once per row of the constraint-evaluation domain.
 *
 * ---- AEROAIR versions 1 and 2 (all integers little-endian; field elements canonical u64 < p) ---------------------------------
 *   bytes 0..7   "AEROAIR" followed by the version byte: 1, or 2 for programs that use what is marked [v2] below (a version-1
 *                program is also a valid version-2 program except for the size of its builder records)
 *   u32 x 16     main_width W (1..255), aux_width A (0..255-W), aux_rands R (0 iff A = 0, <= 255), num_pub (<= 4096),
 *                num_exemptions e (>= 1: the transition divisor is (x^n - 1) / prod_{i=1..e} (x - w^(n-i)); Winterfell's default 1),
 *                num_consts, num_periodic, num_nodes,
 *                num_main_transition, num_aux_transition, num_main_assertions, num_aux_assertions,
 *                num_aux_builders (0, or A), num_sequences [v2; 0 in version 1], reserved x 2 (0)
 *   consts       num_consts x u64
 *   periodic     per column: u32 cycle_len (a power of two >= 2; must not exceed the trace length at proving time), cycle_len x u64.
 *                The column's value at trace step i is values[i mod cycle_len] (`Air::get_periodic_column_values`).
 *   sequences    [v2] per sequence: u32 count (a power of two >= 2), count x u64: the values of one `Assertion::sequence`
 *   nodes        num_nodes x { u32 op; u32 a; u32 b }   op: 1 = a + b, 2 = a - b, 3 = a * b. a, b = operand references; a node may
 *                reference only nodes before it.
 *   transition constraints, main then aux: { u32 root; u32 degree_base; u32 n_cycles; u32 cycle_len x n_cycles }
 *                root = operand reference whose value must vanish on every step but the last e; the declared degree is
 *                `TransitionConstraintDegree::with_cycles(degree_base, cycles)`: evaluation degree
 *                degree_base * (n - 1) + sum (n / cycle) * (cycle - 1). Main constraints may reference MAIN_*, PERIODIC, CONST, PUB
 *                and nodes built from those; aux constraints additionally AUX_* and RAND.
 *   assertions, main then aux: { u32 column; i32 first_step; u32 stride; u32 value }
 *                first_step < 0 means n + first_step (-1 = the last row). stride 0 = `Assertion::single(column, step, value)`;
 *                stride > 0 (a power of two < n, first_step < stride) = `Assertion::periodic(column, first_step, stride, value)`:
 *                the column equals `value` at first_step, first_step + stride, ... . value = operand reference: CONST or PUB for
 *                main assertions; for aux assertions also RAND or a node built from CONST / PUB / RAND only.
 *                [v2] value = 9 << 24 | sequence index with stride > 0 = `Assertion::sequence(column, first_step, stride, values)`: the
 *                column equals values[i] at step first_step + i * stride; at proving time stride * count must equal the trace length.
 *                winter-air 0.4 turns the values into the interpolant P over the subgroup of their own size and checks
 *                column(x) - P(x * w_n^-first_step) against the periodic divisor of (stride, first_step) ("poly_offset"); one
 *                (alpha, beta) pair per assertion, as for the other kinds. Values are program constants (base field), on main and
 *                auxiliary columns alike.
 *   aux builders (how the prover constructs the auxiliary columns, the `build_aux_segment` of this AIR), one per aux column:
 *                version 1: { u32 init; u32 num; u32 den }   column(0) = init, column(i + 1) = column(i) * num(i) / den(i), where
 *                num / den are operand references evaluated on the frame (row i, row i + 1 mod n) of the MAIN segment (MAIN_*,
 *                PERIODIC, CONST, PUB, RAND and nodes of those); den = 0xFFFFFFFF means 1. init: CONST, PUB, RAND or a
 *                row-independent node. This is the running-product shape of every multiset / permutation argument.
 *                [v2] { u32 init; u32 num; u32 den; u32 add_num; u32 add_den }: the affine recurrence
 *                column(i + 1) = column(i) * num(i) / den(i) + add_num(i) / add_den(i) (add_num = 0xFFFFFFFF: no additive term;
 *                add_den = 0xFFFFFFFF: 1) - running sums (log-derivative arguments: num = 1), and mixed forms; row maps
 *                x -> m x + t compose associatively, so the device builds the column with a prefix scan like the products.
 *                [v2] den = 0xFFFFFFFE marks a GENERAL recurrence: column(i + 1) = num, where num may read MAIN_*, PERIODIC, CONST, PUB,
 *                RAND and the CURRENT row of the auxiliary columns up to and including its own (AUX_CUR | j, j <= column): any
 *                per-row recurrence over the frame, e.g. one that squares its own previous value. Such a column cannot be scanned:
 *                the library builds it row after row on the HOST after the device-built columns (about 0.1 us per row and node; a serial
 *                chain has no parallel form, and winter-prover's build_aux_segment is host code too; products, sums and affine forms are
 *                scanned on the device and never take this step).
 *                Not representable: more than ONE auxiliary segment - winter-air 0.4's TraceLayout has NUM_AUX_SEGMENTS = 1
 *                and its proof bytes hold exactly one (width, random elements) pair (proof_format.hpp; pinned on proofs/fib.bin).
 *                A program without builders can be verified and its constraints evaluated, but proving needs the columns.
 *   operand reference = kind << 24 | index:
 *                0 NODE, 1 MAIN_CUR (column of the current row), 2 MAIN_NXT, 3 AUX_CUR, 4 AUX_NXT, 5 PERIODIC, 6 CONST, 7 PUB, 8 RAND,
 *                9 SEQ [v2; only as the value of an assertion]
 *
 * Derived quantities (Winterfell 0.4, restated; `n` = trace length, known at proving time):
 *   constraint-evaluation blowup C = number of composition columns = max over constraints of max(next_pow2(degree_base + n_cycles), 2);
 *   composition coefficients are drawn as (alpha, beta) pairs per transition constraint (main, then aux), then per assertion in
 *   SORTED order — by (stride, first_step, column), main assertions first, then aux (air_instance.cairo:115-142 draws them);
 *   a transition constraint of evaluation degree d is multiplied by alpha + beta * x^((C n - 1 + n - e) - d);
 *   assertions sharing (stride, first_step) share a divisor: x - w^first_step, or x^(n/stride) - w^(first_step * n / stride);
 *   an assertion is multiplied by alpha + beta * x^((C n - 1 + deg(divisor)) - (n - 1));
 *   the numerator columns of the reference's ConstraintEvaluationTable are [transition, one per distinct boundary divisor in
 *   (stride, first_step) order of the main groups, then aux-only divisors].
 */
#ifndef AERO_AIR_H
#define AERO_AIR_H

#include "aero_stark.h"

#ifdef __cplusplus
extern "C" {
#endif

#define AERO_AIR_OP_ADD 1u
#define AERO_AIR_OP_SUB 2u
#define AERO_AIR_OP_MUL 3u
#define AERO_AIR_NODE 0u
#define AERO_AIR_MAIN_CUR 1u
#define AERO_AIR_MAIN_NXT 2u
#define AERO_AIR_AUX_CUR 3u
#define AERO_AIR_AUX_NXT 4u
#define AERO_AIR_PERIODIC 5u
#define AERO_AIR_CONST 6u
#define AERO_AIR_PUB 7u
#define AERO_AIR_RAND 8u
#define AERO_AIR_SEQ 9u
#define AERO_AIR_NONE 0xFFFFFFFFu
#define AERO_AIR_GENERAL 0xFFFFFFFEu /* aux builder `den`: general recurrence (see the format description) */

/* AIR id of a trace file (aero_trace_file_*) whose constraint set travels as an AEROAIR program next to it. */
#define AERO_AIR_PROGRAM 2u

typedef struct aero_air aero_air; /* a parsed, validated and compiled program (host object, immutable, shareable between contexts) */

/* Parse + validate + compile. Replaces `Air::new` / `ProcessorAir::new` (constraints_worker.rs:32-36): everything that does not
 * depend on the trace length is derived here, the rest (degree adjustments, divisors) per call from log_n. AERO_E_BAD_ARG with
 * the reason in err for a malformed program; AERO_E_UNSUPPORTED for a well-formed one this build cannot run. */
int32_t aero_air_load(const uint8_t* program, size_t len, aero_air** out, char* err, size_t err_cap);
void aero_air_free(aero_air* air);
/* The built-in AIRs as programs (so that a host without the Python helpers can obtain them): FibAir(width) with its optional
 * auxiliary segment (desc may be NULL) - byte-for-byte the constraint set aero_prove_fib_air hard-wires. *program is malloc'd. */
int32_t aero_air_fib_program(uint32_t width, const aero_fib_air* desc, uint8_t** program, size_t* len);
/* A synthetic AIR in the SHAPE of a VM's, as a program, and a trace that satisfies it (the role aero_fib_trace plays for FibAir;
 * Miden's own AIR and traces are absent from the reference mount): 20 + 2 * pairs main columns (clock, binary counter, state
 * under power maps of degree 2..7 gated by a periodic selector, degree 5..8 accumulators, permutation columns, Fibonacci pairs),
 * `aux` auxiliary running products over `rands` random elements (two with denominators), 2 transition exemptions, first / last /
 * interior / periodic assertions, 8 composition columns. pairs = 26, aux = 9, rands = 16 is Miden's 72 + 9 shape (BASELINE
 * configs[4]). trace_out: column-major (20 + 2 pairs) x 2^log_n; pub_out: the pairs + 1 public inputs. */
int32_t aero_air_synth_vm_program(uint32_t log_n, uint32_t pairs, uint32_t aux, uint32_t rands, uint8_t** program, size_t* len);
int32_t aero_air_synth_vm_trace(uint32_t log_n, uint32_t pairs, uint64_t* trace_out, uint64_t* pub_out);
/* out = { main_width, aux_width, aux_rands, num_pub, num_exemptions, num_main_transition, num_aux_transition,
 *         num_main_assertions, num_aux_assertions, ce_blowup (= composition columns), num_periodic, num_nodes,
 *         device instructions, base-field registers, extension-field registers, has_aux_builders }
 * (`air.context().num_transition_constraints()`, `num_assertions()`, `ce_blowup_factor()`, `trace_layout()`:
 * miden-to-cairo-parser/src/lib.rs:264-302). */
int32_t aero_air_info(const aero_air* air, uint32_t out[16]);
/* Number of numerator columns (= distinct divisors) for a trace of 2^log_n rows: 1 + boundary divisor groups. */
int32_t aero_air_num_divisors(const aero_air* air, uint32_t log_n, uint32_t* out);

/* Several program-AIR proofs in flight on one GPU (aero_pool_* of aero_stark.h: one context, stream and worker thread per slot):
 * slot i proves traces[i] / host_traces[i] (column-major main_width x 2^log_n) `rounds` times, all with the same program and
 * public inputs; the last proof of each slot is returned (malloc'ed, aero_free). Returns the first non-zero status of any slot. */
struct aero_pool;
int32_t aero_pool_prove_air(struct aero_pool* pool, const aero_air* air, const aero_matrix* const* traces, uint32_t count, const uint64_t* pub,
                            uint32_t n_pub, const aero_proof_options* options, uint32_t rounds, uint8_t** proofs, size_t* proof_lens);
int32_t aero_pool_prove_air_host(struct aero_pool* pool, const aero_air* air, const uint64_t* const* host_traces, uint32_t log_n, uint32_t count,
                                 const uint64_t* pub, uint32_t n_pub, const aero_proof_options* options, uint32_t rounds, uint8_t** proofs,
                                 size_t* proof_lens);
/* The same for a QUEUE of different host traces (aero_pool_prove_fib_queue in aero_stark.h): trace t to slot t mod slots, one statement per
 * trace (pubs_per_trace holds n_traces x n_pub elements), every proof comes back in proofs[t] / proof_lens[t]; all or nothing. */
int32_t aero_pool_prove_air_queue(aero_pool* pool, const aero_air* air, const uint64_t* const* host_traces, uint32_t n_traces, uint32_t log_n,
                                  const uint64_t* pubs_per_trace, uint32_t n_pub, const aero_proof_options* options, uint8_t** proofs, size_t* proof_lens);

/* The evaluation kernel the library generates from the program and compiles at run time (hiprtc, gfx950) when the first proof of a
 * (program, trace length, field) arrives - straight-line HIP over the expression DAG, same arithmetic as the interpreter, same
 * bytes; AERO_AIR_JIT=0 in the environment of aero_create keeps the interpreter. aero_air_jit_compile builds (and caches in the
 * handle) the kernel ahead of the first proof; it needs no GPU. field_extension 1 | 2; fused 1 = the proving path (numerators
 * divided and summed), 0 = the numerator-column form of aero_eval_constraints_program. source: malloc'ed text (aero_free).
 * AERO_AIR_JIT_CACHE=<directory> in the environment keeps the compiled code objects on disk (keyed by the generated source and the
 * hiprtc version), so that a restarted prover does not compile again. */
int32_t aero_air_jit_compile(const aero_air* air, uint32_t log_n, uint32_t field_extension, int32_t fused);
/* Build, ahead of the first proof, exactly the evaluation kernel a proof of 2^log_n rows under `options` will ask for (`world` =
 * ranks the proof is sharded over, 1 = one GPU). The pool entry points above and aero_prove_air_sharded_host call it themselves
 * before their workers / ranks start; a host that wants the cold start off its first proof calls it when it loads the program
 * (the reference pays the equivalent once, when rustc monomorphises `ConstraintEvaluator<ProcessorAir, _>`:
 * constraints_worker.rs:32-43). AERO_OK also when the interpreter is
 * selected (AERO_AIR_JIT=0); AERO_E_UNSUPPORTED + aero_last_error(NULL) when hiprtc refuses the kernel (the proof still works, interpreted). */
int32_t aero_air_prepare(const aero_air* air, uint32_t log_n, const aero_proof_options* options, uint32_t world);
int32_t aero_air_jit_source(const aero_air* air, uint32_t log_n, uint32_t field_extension, int32_t fused, uint8_t** source, size_t* len);

/* `Prover::prove(trace)` + `to_bytes()` for a program AIR (proving_worker.rs:465-467 with the generic `Air`): trace = device
 * matrix of main_width columns; pub = the num_pub public-input elements - they seed the coin (`hash_elements` of the elements,
 * crypto/random.cairo:254-280) and are what PUB operands read. comm may be NULL (one GPU). */
int32_t aero_prove_air(aero_ctx* ctx, const aero_comm* comm, const aero_air* air, const aero_matrix* trace, const uint64_t* pub,
                       uint32_t n_pub, const aero_proof_options* options, uint8_t** proof, size_t* proof_len);
/* Same with the trace in HOST memory (column-major main_width x 2^log_n), the copy on the context's stream in front of the proof. */
int32_t aero_prove_air_host(aero_ctx* ctx, const aero_air* air, const uint64_t* trace_col_major, uint32_t log_n, const uint64_t* pub,
                            uint32_t n_pub, const aero_proof_options* options, uint8_t** proof, size_t* proof_len);

/* ONE proof over the ranks of `comm` from host memory (aero_prove_fib_sharded_host for a program AIR): every rank passes the same
 * column-major trace and copies only its share of the main columns over PCIe; every rank returns the identical proof bytes. */
int32_t aero_prove_air_sharded_host(aero_ctx* ctx, const aero_comm* comm, const aero_air* air, const uint64_t* trace_col_major, uint32_t log_n,
                                    const uint64_t* pub, uint32_t n_pub, const aero_proof_options* options, uint8_t** proof, size_t* proof_len);

/* The constraint seam (ConstraintComputeWorkItem -> ConstraintComputeResult, utils.rs:302-347,417-422; constraints_worker.rs:14-79)
 * for a program AIR: numerator columns of fragment `fragment_offset` of `num_fragments` over the C * n-point constraint domain.
 * aux_lde = (A * deg) component columns (NULL when A = 0), rands = R elements, coeffs = (alpha, beta) pairs in draw order (above),
 * every element `deg` u64. out_cols receives num_divisors * deg columns of C n / num_fragments values (column j * deg + d). */
int32_t aero_eval_constraints_program(aero_ctx* ctx, const aero_air* air, const aero_matrix* trace_lde, const aero_matrix* aux_lde,
                                      uint32_t log_blowup, const uint64_t* pub, uint32_t n_pub, const uint64_t* rands,
                                      const uint64_t* coeffs, uint8_t field_extension, uint32_t fragment_offset, uint32_t num_fragments,
                                      uint64_t* out_cols, uint64_t* frag_index_out);
/* `Trace::build_aux_segment(rand_elements)` from the program's builders: (A * deg) x n component columns. */
int32_t aero_aux_columns_program(aero_ctx* ctx, const aero_air* air, const aero_matrix* trace, const uint64_t* pub, uint32_t n_pub,
                                 const uint64_t* rands, uint8_t field_extension, aero_matrix** aux_out);
/* `Trace::validate(&air)` - the check a DEBUG build of the reference's prover runs inside commit_to_trace_and_validate
 * (proving_worker.rs:323-332): every transition constraint on every row but the exempted last ones, every assertion on the steps it
 * names, evaluated on the trace itself by a kernel compiled from the program (needs hiprtc). aux = the (aux_width * degree) x n
 * component columns of aero_aux_columns_program with the same `rands`, or NULL: then only the main segment is checked.
 * *first_failure = UINT64_MAX when the trace satisfies the program, else row << 24 | id of the first failing check (id = index of the
 * transition constraint, main first, or 0x800000 | index of the assertion, main first). The prover itself never validates a trace. */
int32_t aero_air_validate_trace(aero_ctx* ctx, const aero_air* air, const aero_matrix* trace, const aero_matrix* aux, const uint64_t* pub,
                                uint32_t n_pub, const uint64_t* rands, uint8_t field_extension, uint64_t* first_failure);
/* `ConstraintEvaluationTable::into_poly` -> `CompositionPoly` for a program AIR: numer_cols = the num_divisors * deg numerator
 * columns over the whole constraint domain (host); result as aero_composition_poly_air. */
int32_t aero_composition_poly_program(aero_ctx* ctx, const aero_air* air, const uint64_t* numer_cols, uint32_t log_n,
                                      uint8_t field_extension, aero_matrix** comp_polys);

/* `winter_verifier::verify` for a program AIR (host only): aero_verify_fib with the out-of-domain constraint check evaluated by a
 * host interpreter of the same program (periodic columns at z^(n/cycle) through their interpolants). */
int32_t aero_verify_air(const uint8_t* proof, size_t proof_len, const uint64_t* pub, uint32_t n_pub, const aero_air* air,
                        const aero_verify_policy* policy, char* err, size_t err_cap);

/* The constraint worker at the message level: the bincode ConstraintComputeWorkItem the SDK's pool posts (utils.rs:302-347) ->
 * bincode ConstraintComputeResult. The message carries no AIR identity (the reference's worker hard-wires ProcessorAir), so the
 * program is the second argument; PUB operands read pub (NULL / 0: the elements of the Miden PublicInputs in the message,
 * program hash || stack inputs || outputs.stack || overflow addresses, crypto/random.cairo:254-280). Base field only. */
int32_t aero_worker_eval_constraints(aero_ctx* ctx, const uint8_t* work_item, size_t work_item_len, const aero_air* air,
                                     const uint64_t* pub, uint32_t n_pub, uint8_t** result, size_t* result_len);

#ifdef __cplusplus
}
#endif
#endif /* AERO_AIR_H */
