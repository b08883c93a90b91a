/*
 * tsim_hip.h - C ABI of the MI355X (gfx950) stabilizer-rank sampling engine.
 *
 * This is the drop-in boundary for the one hot path of QuEraComputing/tsim:
 *
 *     tsim.sampler.sample_program(program, f_params, key) -> bool[B, num_outputs]
 *                                       (reference: src/tsim/sampler.py:117-167)
 *     tsim.compile.evaluate.evaluate(circuit, param_vals) -> complex64[B]
 *                                       (reference: src/tsim/compile/evaluate.py:15-59)
 *
 * The reference is pure Python/JAX and has no FFI of its own; these entry
 * points are what a binding for that seam binds (the ctypes stub is shown in
 * INTEGRATION.md, the shipped one is tsim_amd/_lib.py).  Plain pointers and
 * sizes only - no torch / numpy / jax types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative TSIM_E* code and never
 *     throws or prints; tsim_last_error() returns a thread-local message;
 *   - "program description" arrays use EXACTLY the reference layout: one byte
 *     per bit, row-major, padded to the per-family maximum term count
 *     (src/tsim/compile/compile.py:21-37, src/tsim/compile/terms.py:42-207);
 *     the library bit-packs them and uploads them once per device;
 *   - a handle is bound to one HIP device; calls on one handle must not race;
 *   - packed shot rows are little-endian bit strings: bit i of a row lives in
 *     64-bit word i/64 at position i%64 (== numpy.packbits(bitorder="little")).
 */
#ifndef TSIM_HIP_H
#define TSIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSIM_OK 0
#define TSIM_EINVAL (-22)      /* bad argument / malformed program           */
#define TSIM_ENOMEM (-12)      /* host or device allocation failed            */
#define TSIM_EHIP (-5)         /* a HIP runtime call failed                   */
#define TSIM_ENOTSUP (-95)     /* program exceeds a compiled-in limit         */
#define TSIM_ESTATE (-1)       /* call not valid in the handle's state        */

#define TSIM_MAX_PARAMS 2048   /* max n_params (f bits + outputs) per level   */

typedef struct tsim_program tsim_program;

/*
 * One autoregressive level == one reference CompiledScalarGraphs
 * (src/tsim/compile/compile.py:21-37).  G = num_graphs, P = n_params.
 */
typedef struct tsim_level_desc {
  int32_t num_graphs;            /* G */
  int32_t n_params;              /* P */
  int32_t ta, tb, tc, td;        /* padded term counts of the four families   */
  /* NodePhases   (terms.py:42-73)   */
  const uint8_t *a_phases;       /* [G, ta]     values 0..7                   */
  const uint8_t *a_params;       /* [G, ta, P]  0/1                           */
  const int32_t *a_counts;       /* [G]         real terms per graph          */
  /* HalfPiPhases (terms.py:76-107)  */
  const uint8_t *b_coeffs;       /* [G, tb]     0,2,4,6 (0 = padding)         */
  const uint8_t *b_params;       /* [G, tb, P]                                */
  /* PiProducts   (terms.py:110-144) */
  const uint8_t *c_psi_const;    /* [G, tc]                                   */
  const uint8_t *c_psi_params;   /* [G, tc, P]                                */
  const uint8_t *c_phi_const;    /* [G, tc]                                   */
  const uint8_t *c_phi_params;   /* [G, tc, P]                                */
  /* PhasePairs   (terms.py:147-187) */
  const uint8_t *d_alpha;        /* [G, td]     values 0..7                   */
  const uint8_t *d_alpha_params; /* [G, td, P]                                */
  const uint8_t *d_beta;         /* [G, td]                                   */
  const uint8_t *d_beta_params;  /* [G, td, P]                                */
  const int32_t *d_counts;       /* [G]                                       */
  /* ScalarPrefactor (terms.py:190-207) */
  const uint8_t *phase_indices;  /* [G]   0..7                                */
  const int32_t *floatfactor;    /* [G,4] (a,b,c,d) on basis (1,w,i,conj w)   */
  const int32_t *power2;         /* [G]                                       */
  const float *approx;           /* [G,2] complex64 (re,im); may be NULL      */
  int32_t has_approx;            /* static flag has_approximate_floatfactors  */
} tsim_level_desc;

/* ---- program construction (replaces the device upload implied by
 *      jnp.asarray of a CompiledProgram, src/tsim/core/types.py:80-107) ---- */

/* direct_f_indices/direct_flips: [n_direct]; output_order: [num_outputs]
 * (direct entries first, then the compiled components in processing order). */
int tsim_program_create(int32_t num_outputs, int32_t num_detectors, int32_t n_direct,
                        const int32_t *direct_f_indices, const uint8_t *direct_flips,
                        const int32_t *output_order, tsim_program **out);

/* Adds a CompiledComponent (types.py:55-77); components must be added in the
 * reference's processing order.  n_levels is n_out+1 (sequential mode) or 2
 * (joint mode, evaluate-only).  Returns the component index (>= 0). */
int tsim_program_add_component(tsim_program *p, int32_t n_out, const int32_t *output_indices,
                               int32_t F, const int32_t *f_selection, int32_t n_levels);

/* Adds the next level of component `component` (arrays are copied/packed). */
int tsim_program_add_level(tsim_program *p, int32_t component, const tsim_level_desc *level);

/*
 * Evaluation formulation, to be set before finalize (default TSIM_MODE_AUTO; the environment
 * variable TSIM_AMD_MODE=faithful forces TSIM_MODE_FAITHFUL):
 *   TSIM_MODE_FAITHFUL  operation-by-operation mirror of the reference's int32 arithmetic
 *                       (src/tsim/core/exact_scalar.py:19-137), identical even where it wraps;
 *   TSIM_MODE_AUTO      use the faster exact-value formulation (NodePhases by class counting,
 *                       phase exponent as a Dickson-reduced GF(2) quadratic form, pack-time term
 *                       tables; LDS chunk-table kernel when every component has <= 64 parameters - since
 *                       round 5 up to 80 with <= 64 selected f bits, and up to 128 for components of many graphs)
 *                       whenever every graph qualifies (<= 30 NodePhases terms,
 *                       even HalfPi coefficients, small floatfactors); it yields the same canonical
 *                       (a,b,c,d,power) and float32 amplitude as the reference whenever the
 *                       reference's own int32 arithmetic does not wrap.
 */
#define TSIM_MODE_AUTO 0
#define TSIM_MODE_FAITHFUL 1
#define TSIM_MODE_ROW_KERNEL 2   /* exact-value formulation, but the row-by-row kernel instead of the
                                    LDS chunk-table kernel (k_sample4); for tests and A/B timing */
int tsim_program_set_mode(tsim_program *p, int32_t mode);
/* after finalize: *fast = 1 if the exact-value formulation was selected */
int tsim_program_get_mode(const tsim_program *p, int32_t *fast);

/*
 * Low-weight error-pattern tables (before finalize).  The Bernoulli thresholds p1/prev of
 * _sample_component (sampler.py:54-79) depend on a shot only through the component's selected
 * f bits and the outcome prefix; for f_sel patterns of weight <= max_weight (0..7, as many as fit a
 * 1 GiB table per component; TSIM_AMD_PATTERN_TABLE_MB overrides) they are tabulated by the sampling kernels' own
 * arithmetic, and shots carrying such patterns in every component are finished by a light first
 * pass (one Threefry draw + one table read per output); only the remaining rows run the full
 * kernel.  Bit-identical results.  Requires <= 12 outputs and <= 64 parameters per component (round 5: <= 128, see
 * TSIM_MODE_AUTO above) - or, for the programs of the sparse-column kernels (any number of components of up to 511
 * selected f bits in ascending order, <= 8 outputs each, f indices below 2048): weight <= 4, below 4 GiB per component
 * (C(200, <= 4) patterns of 8 thresholds are 2.1 GB), built on the device by unranking the pattern index.
 *   enable: 1 on, 0 off, -1 default (on in TSIM_MODE_AUTO);  max_weight: 0..7 pins the depth and builds it at finalize;
 *   -1 (round 5) = the deepest tables that cost about half a millisecond at finalize (weight 3-4), the default depth -
 *   5, or 4 for the sparse-column programs - built in the background and put in place at a later launch
 *   (tsim_program_tables_pending), one more weight (up to 7) after ~10^10 rows that leave too many rows to the
 *   full kernels (TSIM_AMD_DEEP_TABLES=1: at once; -1: never).
 * The environment variable TSIM_AMD_PATTERN_TABLES=0/1 overrides `enable`.
 * Launch plan: the hard-row kernel reports the number of hard rows of each launch through mapped
 * host memory; when most rows of recent launches were hard (dense error patterns) the following
 * launches skip the first pass (re-probing every 16th launch), and when the hard-row lists are
 * short the overflow launch of the full kernel is dropped.  TSIM_AMD_ADAPTIVE=0 pins the default
 * plan.  Results never depend on the plan.
 */
int tsim_program_set_pattern_tables(tsim_program *p, int32_t enable, int32_t max_weight);
/* after finalize: *enabled, table bytes, and max tabulated weight per component ([n_components],
 * may be NULL) */
int tsim_program_pattern_table_info(const tsim_program *p, int32_t *enabled, int64_t *table_bytes,
                                    int32_t *max_weight);
/* *pending = 1 while pattern tables of another depth are being built in the background (the shallow start's default
 * depth, or a deeper one the launch plan asked for); they are put in place at a later launch.  Rates measured while
 * this is 1 are those of the transient.  (No reference counterpart: tsim_amd's own table machinery.) */
int tsim_program_tables_pending(const tsim_program *p, int32_t *pending);

/* Packs all levels into the device image and uploads it to HIP device `device`. */
int tsim_program_finalize(tsim_program *p, int32_t device);

void tsim_program_destroy(tsim_program *p);

/* ---- the hot path ---------------------------------------------------- */

/*
 * sample_program for one batch, host buffers in the reference layout
 * (replaces src/tsim/sampler.py:117-167 incl. the H2D at :398 and D2H at :415).
 *   f            uint8 [B, num_f], nonzero == 1
 *   key_hi/lo    the post-split Threefry-2x32 subkey handed to sample_program
 *                (src/tsim/sampler.py:399)
 *   shot_offset  in-batch index of row 0 (Threefry counter of shot s is
 *                shot_offset+s): a batch sharded over devices reproduces the
 *                unsharded result bit for bit
 *   out          out_packed == 0: uint8 [B, num_outputs] (0/1)
 *                out_packed != 0: uint8 [B, 8*ceil(num_outputs/64)] little-endian bits
 *   max_norm_dev float [n_components] (may be NULL): max |norm-1| of the
 *                normalisation check of src/tsim/sampler.py:71-72, valid only
 *                when the call contains in-batch shot 0 (else left untouched)
 */
int tsim_sample_batch(tsim_program *p, const uint8_t *f, int64_t B, int32_t num_f,
                      uint32_t key_hi, uint32_t key_lo, int64_t shot_offset,
                      uint8_t *out, int32_t out_packed, float *max_norm_dev);

/*
 * Same, on device-resident packed buffers (no host transfer, asynchronous on
 * `stream`; pass NULL for the handle's own stream):
 *   d_f    uint64 [B, ceil(num_f/64)]   packed error-mechanism rows
 *   d_out  uint64 [B, ceil(num_outputs/64)]
 *   d_max_norm_dev  device float [n_components] or NULL
 */
int tsim_sample_batch_device(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f,
                             uint32_t key_hi, uint32_t key_lo, int64_t shot_offset,
                             uint64_t *d_out, float *d_max_norm_dev, void *stream);

/*
 * Pipelined form of tsim_sample_batch_device.  With the pattern tables active a launch is two
 * passes; the second one (a few thousand "hard" rows, latency-bound) does not need the GPU to
 * itself.  Each `slot` (0 .. TSIM_PIPELINE_SLOTS-1) owns a stream - a lane (slot 0's lane is the handle's
 * own stream, tsim_get_stream: it sits on a hardware queue of its own); _begin enqueues the whole
 * launch on the slot's lane, so launches of one slot are ordered among themselves and launches of
 * different slots overlap (the second pass of one under the first pass of the next).  The lane first
 * waits for the work already queued on `stream` (the producer of d_f) unless `flags` has
 * TSIM_PIPE_INPUTS_READY (inputs complete, and nothing queued on `stream` still uses d_out): then no
 * cross-stream event is needed at all.  _end makes `stream` wait for the slot's lane - only after
 * _end (and the usual stream ordering) are d_out / d_max_norm_dev complete.  The caller must not reuse
 * the buffers of a slot for anything else between _begin and _end.  Results are identical to the
 * serial call.  Keep the number of busy streams (lanes + the caller's + RCCL's) at 4 or fewer: beyond
 * the 4 hardware queues HIP uses, launches slow down by 3x on this stack.
 *
 * Deferred second pass.  When recent launches left few hard rows (the launch-plan feedback: longest
 * list <= 192), _begin only enqueues the FIRST pass, alternating between the lanes of slots 0 and 1,
 * and keeps the launch's hard rows for a batch: every TSIM_AMD_DEFER_GROUP (default 4; 8 for programs with
 * more than 4 MB of chunk tables, whose hard-row pass is long) launches - or
 * when _end / _begin / tsim_synchronize needs a slot whose rows are still waiting - ONE grid
 * (k_sample4h_multi) serves the hard rows of all waiting launches on the lane of slot 2, after their
 * first passes.  No lane then waits for a second pass before its next first pass; a slot's next
 * launch still waits for the batch that served its previous one (an event query, a stream wait only
 * if needed), so the number of SLOTS in flight - 8 is enough on C2 - hides the batch latency, not the
 * number of lanes.  The number of hard-row lists follows the load (4..64, about 40 rows each) so that
 * the 64-row blocks of the hard-row kernel are filled.  TSIM_AMD_DEFER_HARD=0 / TSIM_AMD_MERGE_LISTS=0
 * switch these off.  Results never depend on any of it.
 */
#define TSIM_PIPELINE_SLOTS 32
#define TSIM_PIPE_INPUTS_READY 1u
/* d_out receives the reference's bit_packed rows, uint8 [B, ceil(num_outputs/8)] (sampler.py:665-669), INSTEAD of
 * the padded 8-byte words: ceil(n/8) bytes written per shot (3 instead of 8 for 20 outputs); any alignment */
#define TSIM_PIPE_OUT_BIT_PACKED 2u
int tsim_sample_batch_device_begin(tsim_program *p, int32_t slot, const uint64_t *d_f, int64_t B,
                                   int32_t num_f, uint32_t key_hi, uint32_t key_lo, int64_t shot_offset,
                                   uint64_t *d_out, float *d_max_norm_dev, void *stream, uint32_t flags);
int tsim_sample_batch_device_end(tsim_program *p, int32_t slot, void *stream);
/* Non-consuming form of _end: `stream` (NULL: the handle's stream) waits for the launch `slot` carried LAST, whether or
 * not a stream has been joined to it before (_end marks the slot as joined and a second _end adds no wait).  What a
 * producer needs before it overwrites the slot's INPUT buffer - the reference has no such hazard, its batches are
 * synchronous (sampler.py:398-404) - when a consumer stream already took the slot's output. */
int tsim_pipeline_wait_slot(tsim_program *p, int32_t slot, void *stream);
/* Make every pipeline lane wait for the work already queued on `stream` (NULL: the handle's stream) -
 * one event for all lanes.  Launches whose buffers depend only on that work may then pass
 * TSIM_PIPE_INPUTS_READY (bench.py: once per gather group instead of once per launch). */
int tsim_pipeline_wait_stream(tsim_program *p, void *stream);
/* tsim_sample_batch_device_end(slot, stream) for every slot that has a launch in flight: `stream` is then behind everything the
 * pipeline has been given (one call; NULL = the handle's stream). */
int tsim_pipeline_join(tsim_program *p, void *stream);
/* The stream of lane `lane` (= of slot `lane`; created on demand).  Lane 2 is where the deferred
 * hard-row batches run, i.e. where results complete: a consumer that joins its slots on THAT stream
 * (tsim_sample_batch_device_end(p, slot, lane2)) and queues its own work there (bench.py: the RCCL
 * gather) never makes a first-pass lane wait. */
int tsim_pipeline_lane_stream(tsim_program *p, int32_t lane, void **stream);
/* The NEXT _begin on `slot` also writes its rows as uint8[B, ceil(num_outputs/8)] into d_compact (the
 * reference's bit_packed layout, sampler.py:665-669) straight from the sampling kernels - no separate
 * compaction kernel.  One-shot: cleared by that launch. */
int tsim_pipeline_set_compact_output(tsim_program *p, int32_t slot, uint8_t *d_compact);
/* The same for a run of launches: the next `count` pipelined launches (any slot, in call order) write
 * their bit_packed rows to d_base, d_base + stride_bytes, ... - one call per gather group instead of one
 * per launch.  A per-slot buffer set with tsim_pipeline_set_compact_output takes precedence. */
int tsim_pipeline_set_compact_series(tsim_program *p, uint8_t *d_base, int64_t stride_bytes, int32_t count);
/* Between _begin and _end of `slot`: tsim_compact_rows_device of the launch's output rows, enqueued
 * on the slot's lane behind the launch; _end then also covers d_out. */
int tsim_sample_batch_device_compact(tsim_program *p, int32_t slot, const uint64_t *d_rows, int64_t B,
                                     int32_t nbits, uint8_t *d_out, void *stream);

/*
 * Device-side post-selection (the shot-skipping of src/tsim/sampler.py:422-545, done in HBM):
 *   tsim_postselect_device writes every row's DIRECT output bits to d_out (compiled columns 0),
 *   tests ((row ^ ref) & mask) != 0 per row (mask/ref: packed uint64 [ceil(num_outputs/64)] in final
 *   column order; ref may be NULL) and appends the surviving row numbers to d_row_index
 *   (uint32 [B], unordered), their count to d_row_count; d_discarded (uint8 [B]) is optional.
 *   tsim_sample_rows_device then samples only the listed rows (overwriting their d_out rows with
 *   direct + compiled bits).  The Threefry counter of a row stays its own in-batch index, so the
 *   result does not depend on the order of the list.
 */
int tsim_postselect_device(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f,
                           const uint64_t *d_mask, const uint64_t *d_ref, uint64_t *d_out,
                           uint32_t *d_row_index, uint32_t *d_row_count, uint8_t *d_discarded, void *stream);
int tsim_sample_rows_device(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f,
                            uint32_t key_hi, uint32_t key_lo, int64_t shot_offset, uint64_t *d_out,
                            float *d_max_norm_dev, const uint32_t *d_row_index, const uint32_t *d_row_count,
                            void *stream);

/*
 * evaluate(circuit, param_vals) for level `level` of component `component`
 * (replaces src/tsim/compile/evaluate.py:15-59).
 *   params        uint8 [B, n_params]
 *   re, im        float [B]  complex64 amplitude
 *   abs_out       optional float [B]: |amplitude| as jnp.abs(complex64) forms it
 *                 (the marginal of src/tsim/sampler.py:54,67,945,951)
 *   coeffs_power  optional int32 [B,5]: exact (a,b,c,d,power) of the summed
 *                 amplitude on the exact branch (zeros on the approximate one)
 */
int tsim_evaluate(tsim_program *p, int32_t component, int32_t level, const uint8_t *params,
                  int64_t B, float *re, float *im, float *abs_out, int32_t *coeffs_power);

/* ---- device-side data-format kernels either side of the path ---------- */

/* uint8 [B,num_f] (device) -> packed uint64 [B,ceil(num_f/64)] (device)      */
int tsim_pack_bits_device(tsim_program *p, const uint8_t *d_in, int64_t B, int32_t nbits,
                          uint64_t *d_out, void *stream);
/* packed uint64 [B,ceil(nbits/64)] (device) -> uint8 [B,nbits] (device)      */
int tsim_unpack_bits_device(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t nbits,
                            uint8_t *d_out, void *stream);
/* uint64[B, in_words] padded rows -> uint8[B, ceil(nbits/8)] rows holding the first nbits columns:
 * the reference's bit_packed=True layout (np.packbits(bits[:, :nbits], axis=1, bitorder="little"),
 * sampler.py:665-669) - what a gather or a packed D2H has to move (3 instead of 8 bytes per shot for
 * 20 outputs).  in_words = 0: ceil(nbits/64).  d_out 4-byte aligned. */
int tsim_compact_rows_device(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t in_words,
                             int32_t nbits, uint8_t *d_out, void *stream);
/* Post-selection after sampling (sampler.py:422-545 for device-side noise, where no host stream decides which shots reach
 * sample_program): rows of `row_bytes` bytes each (padded words or bit_packed); d_masks = 5 x row_bytes bytes: test mask,
 * reference XORed before the test, columns a discarded row keeps, XOR for surviving rows, XOR for discarded rows;
 * d_gone (optional) receives 0/1 per row. */
int tsim_postselect_rows_device(tsim_program *p, uint8_t *d_rows, int64_t B, int32_t row_bytes, const uint8_t *d_masks,
                                uint8_t *d_gone, void *stream);
/* The survivors of a chunk (d_gone[i] == 0, from tsim_postselect_device), IN SHOT ORDER, appended to a device-resident
 * queue of shot ids: queue[*d_tail ...] = base + i, *d_tail += their number (the reference's compacted batches,
 * sampler.py:466-508: the order fixes the survivors' Threefry counters).  d_scratch: ceil(n / 1024) uint32. */
int tsim_survivors_append_device(tsim_program *p, const uint8_t *d_gone, int64_t n, uint32_t base, uint32_t *d_scratch,
                                 uint32_t *d_queue, uint32_t *d_tail, void *stream);
/* The epilogue of CompiledDetectorSampler.sample (sampler.py:850-868: detectors / observables prepended, appended,
 * separate; reference-sample flips) and _maybe_bit_pack (:665-669) on the device: out column c = in column
 * (d_cols[c] & 0x7FFFFFFF) XOR (d_cols[c] >> 31) of the padded rows, one byte per column (packed = 0) or
 * ceil(n_cols / 8) bytes per row (packed = 1, np.packbits little-endian). */
int tsim_arrange_rows_device(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t in_words, const uint32_t *d_cols,
                             int32_t n_cols, int32_t packed, uint8_t *d_out, void *stream);

/* Row gather / scatter by index on packed rows of `words` 64-bit words - the data movement of the
 * reference's host-noise post-selection (src/tsim/sampler.py:466-508: survivors are compacted into dense
 * batches of batch_size, the last one padded with its first row; result rows go back to the shots they
 * came from):
 *   gather : d_dst[i] = d_src[d_index[i < n_valid ? i : 0]]   for i < n_total
 *   scatter: d_dst[d_index[i]] = d_src[i]                      for i < n                    */
int tsim_gather_rows_device(tsim_program *p, const uint64_t *d_src, int32_t words, const uint32_t *d_index,
                            int64_t n_valid, int64_t n_total, uint64_t *d_dst, void *stream);
int tsim_scatter_rows_device(tsim_program *p, const uint64_t *d_src, int32_t words, const uint32_t *d_index,
                             int64_t n, uint64_t *d_dst, void *stream);

/* ---- host-side noise sampler ON NUMPY'S STREAM (bit-exact replacement of ChannelSampler.sample,
 *      src/tsim/noise/channels.py:624-658, for a numpy Generator backed by PCG64; no device involved) --------- */

/* PCG64 state and increment as numpy's bit_generator.state reports them (128-bit integers, split in halves) */
typedef struct tsim_pcg64 {
  uint64_t state_lo, state_hi, inc_lo, inc_hi;
} tsim_pcg64;

#define TSIM_PCG_RAW 0          /* Generator.bit_generator.random_raw  -> uint64 */
#define TSIM_PCG_DOUBLE 1       /* Generator.random / uniform(0, 1)    -> double */
#define TSIM_PCG_EXPONENTIAL 2  /* Generator.standard_exponential      -> double (ziggurat) */
#define TSIM_PCG_GEOMETRIC 3    /* Generator.geometric(p)              -> int64  */
/* n draws of one kind; `rng` is advanced exactly as numpy advances it */
int tsim_pcg_draw(tsim_pcg64 *rng, int32_t kind, double p, int64_t n, void *out);

/* One ChannelSampler.sample(num_samples) call: per channel c (tables as _precompute_sparse builds them,
 * channels.py:578-622) n_draws = int(B p + 7 sqrt(B p (1 - p))) + 100 geometric gaps, positions = cumsum - 1
 * (those < B fire), one uniform per fired row, outcome = searchsorted(cond_cdf_c, u), row ^= pattern.
 *   n_outcomes[c]  non-identity outcomes of channel c;  cond_cdf  concatenated conditional CDFs;
 *   patterns       uint64 [sum n_outcomes, words] packed XOR patterns;  rows  uint64 [num_samples, words] (overwritten)
 *   threads        worker threads of the scatter pass (0: up to 8).  The stream consumption is sequential. */
int tsim_pcg_sample_channels(tsim_pcg64 *rng, int32_t n_channels, const double *p_fire, const int32_t *n_outcomes,
                             const double *cond_cdf, const uint64_t *patterns, int32_t words, int64_t num_samples,
                             uint64_t *rows, int32_t threads);

/* ---- device-side noise sampler (statistical replacement of ChannelSampler.sample,
 *      src/tsim/noise/channels.py:578-658; the numpy PCG64 stream is not reproduced) ---------- */
typedef struct tsim_noise tsim_noise;

/* Channel tables exactly as ChannelSampler._precompute_sparse builds them (channels.py:578-622):
 *   p_fire[n_channels]; n_outcomes[c] = number of non-identity outcomes of channel c;
 *   cond_cdf = the concatenated conditional CDFs; xor_patterns = uint8 [sum n_outcomes, num_f].
 * The sampler lives on the program's device and stream. */
int tsim_noise_create(tsim_program *p, int32_t num_f, int32_t n_channels, const double *p_fire,
                      const int32_t *n_outcomes, const double *cond_cdf, const uint8_t *xor_patterns,
                      tsim_noise **out);
/* d_f: uint64 [B, ceil(num_f/64)] (device); overwritten with a fresh batch keyed by (key_hi, key_lo). */
int tsim_noise_sample_device(tsim_noise *n, int64_t B, uint32_t key_hi, uint32_t key_lo, uint64_t *d_f,
                             void *stream);
/* tsim_sample_steps_device with the f rows drawn on the device, in the same call (the reference's batch loop,
 * src/tsim/sampler.py:393-400: channel sampler, then sample_program, per batch).  Batch j's noise key is the j-th
 * split of noise_key (advanced like `key`); its f rows are written to d_f[j] - the bytes tsim_noise_sample_device
 * writes for that key.  Programs of one component of at most 8 outputs over f rows of at most 128 bits draw the noise
 * INSIDE their first pass (one kernel: csrc/tsim_noise_fused.hip.h); every other program runs the noise kernel in
 * front of its first pass.  Results do not depend on which. */
int tsim_sample_steps_noise_device(tsim_program *p, tsim_noise *n, int32_t n_steps, uint64_t *const *d_f, int64_t B,
                                   int32_t num_f, uint32_t key[2], uint32_t noise_key[2], int64_t shot_offset,
                                   void *const *d_out, float *const *d_max_norm_dev, uint32_t flags);
void tsim_noise_destroy(tsim_noise *n);

/* ---- multi-GPU: RCCL over xGMI, issued by the library (no PyTorch) --------------------------------------
 *
 * The path shards over shots: rank r of R evaluates in-batch rows [r B/R, (r+1) B/R)
 * with shot_offset = r B/R; the Threefry counter is the global row index, so results do not depend on R and
 * no data-path exchange exists.  The one collective assembles the finished bit_packed rows.  The reference
 * has no multi-device path (src/tsim/sampler.py:310 uses jax.devices()[0]).
 *
 * One process per GPU.  Rank 0 makes a unique id, the host processes pass its 128 bytes to every rank by
 * any out-of-band channel (tsim_amd/dist.py: file or TCP rendezvous), every rank calls tsim_dist_init.
 * Collectives are asynchronous on `stream` (NULL: the communicator's own stream); order them after the
 * sampling kernels by passing the stream those ran on (tsim_get_stream / tsim_pipeline_lane_stream), or with
 * tsim_dist_stream_wait.
 *   tsim_dist_gather_rows   every rank sends nbytes; rank `root` receives world*nbytes, rank-major (ncclGather)
 *   tsim_dist_alltoall_rows chunk j of every rank's send buffer (nbytes_per_peer each) lands on rank j at
 *                           offset rank*nbytes_per_peer: a gather whose roots are spread over the node
 *   tsim_dist_allreduce_max / tsim_dist_barrier   host-value helpers (blocking) for timing harnesses       */
#define TSIM_DIST_ID_BYTES 128
typedef struct tsim_dist tsim_dist;
int tsim_dist_unique_id(uint8_t id[TSIM_DIST_ID_BYTES]);
int tsim_dist_init(int32_t device, const uint8_t id[TSIM_DIST_ID_BYTES], int32_t rank, int32_t world, tsim_dist **out);
void tsim_dist_destroy(tsim_dist *d);
int tsim_dist_info(const tsim_dist *d, int32_t *rank, int32_t *world);
int tsim_dist_gather_rows(tsim_dist *d, const void *d_send, int64_t nbytes, void *d_recv, int32_t root, void *stream);
int tsim_dist_alltoall_rows(tsim_dist *d, const void *d_send, void *d_recv, int64_t nbytes_per_peer, void *stream);
int tsim_dist_allreduce_max(tsim_dist *d, double *value);
int tsim_dist_barrier(tsim_dist *d);
/* make `waiting_stream` wait for the work queued so far on `signalling_stream` (NULL = the communicator's) */
int tsim_dist_stream_wait(tsim_dist *d, void *waiting_stream, void *signalling_stream);
/* Cross-stream ordering around a collective without draining anything: tsim_dist_mark records marker `mark`
 * (0 .. TSIM_DIST_MARKS-1) on `stream` - e.g. right after a gather was queued there - and
 * tsim_dist_wait_mark makes `stream` wait for that marker only (a marker never recorded is a no-op). */
#define TSIM_DIST_MARKS 8
int tsim_dist_mark(tsim_dist *d, int32_t mark, void *stream);
int tsim_dist_wait_mark(tsim_dist *d, int32_t mark, void *stream);
/* hipDeviceSynchronize on `device` (timing harnesses that must not depend on torch.cuda.synchronize) */
int tsim_device_synchronize(int32_t device);

/* ---- plumbing: memory, streams, timing (replaces utils/cuda_helpers.py:73-141) */

int tsim_device_count(int32_t *count);
/* free / total bytes of the handle's device (hipMemGetInfo): what the reference's batch sizing reads from
 * jax's memory_stats (src/tsim/sampler.py:308-320) */
int tsim_mem_info(tsim_program *p, int64_t *free_bytes, int64_t *total_bytes);
/* Buffers are owned by the handle: tsim_program_destroy frees whatever was not returned. */
int tsim_malloc_device(tsim_program *p, int64_t nbytes, void **d_ptr);
int tsim_free_device(tsim_program *p, void *d_ptr);
int tsim_malloc_pinned(int64_t nbytes, void **h_ptr);      /* hipHostMalloc   */
int tsim_free_pinned(void *h_ptr);
int tsim_memcpy_h2d(tsim_program *p, void *d_dst, const void *h_src, int64_t nbytes);
int tsim_memcpy_d2h(tsim_program *p, void *h_dst, const void *d_src, int64_t nbytes);
int tsim_synchronize(tsim_program *p);                      /* handle's stream */
/* Asynchronous variants on a caller-chosen stream (NULL: the handle's) and the matching wait: the end-to-end sampler
 * downloads finished batches while later ones are sampled (utils/cuda_helpers.py:105-141 copies once, at the end).
 * Pageable host memory is allowed; the call may then block until the copy is done. */
int tsim_memcpy_d2h_async(tsim_program *p, void *h_dst, const void *d_src, int64_t nbytes, void *stream);
int tsim_memcpy_h2d_async(tsim_program *p, void *d_dst, const void *h_src, int64_t nbytes, void *stream);
int tsim_stream_synchronize(tsim_program *p, void *stream);
/* Auxiliary hipStream_t owned by the handle (index 0 .. TSIM_AUX_STREAMS-1; created on first use): work beside the
 * sampling lanes - the device-side channel sampler, transfers. */
#define TSIM_AUX_STREAMS 4
int tsim_aux_stream(tsim_program *p, int32_t index, void **stream);
/* The pipeline slot the next batch of tsim_sample_steps_device will take (it rotates over all TSIM_PIPELINE_SLOTS): pass
 * it to tsim_sample_batch_device_end later to make a stream wait for exactly that batch. */
int tsim_pipeline_next_slot(tsim_program *p, int32_t *slot);
/* the handle's hipStream_t, e.g. to order a collective after the sampling kernel */
int tsim_get_stream(tsim_program *p, void **stream);

/* Host helper, no device involved: `new_key, subkey = jax.random.split(key)` for the threefry2x32
 * key layout (the once-per-batch split of sampler.py:399); out = {new_hi, new_lo, sub_hi, sub_lo}. */
void tsim_key_split(uint32_t key_hi, uint32_t key_lo, uint32_t out[4]);
/* The per-batch idiom of the reference's sample loop in one call (sampler.py:399-401):
 * `key, subkey = split(key)` then tsim_sample_batch_device_begin with `subkey`; key = {hi, lo} is
 * updated in place.  Saves a foreign-function round trip per batch for Python hosts. */
int tsim_sample_batch_device_begin_split(tsim_program *p, int32_t slot, const uint64_t *d_f, int64_t B,
                                         int32_t num_f, uint32_t key[2], int64_t shot_offset, uint64_t *d_out,
                                         float *d_max_norm_dev, void *stream, uint32_t flags);

/* SEVERAL consecutive batches of the reference's batch loop in one call (src/tsim/sampler.py:340-420: per batch
 * `key, subkey = split(key)`, sampler.py:399, then one sample_program, sampler.py:117-167).  For j < n_steps:
 * d_out[j] = sample_program(d_f[j], subkey_j) - exactly what n_steps calls of tsim_sample_batch_device_begin_split
 * on consecutive pipeline slots give, bit for bit; `key` = {hi, lo} is advanced n_steps times.  The call only
 * enqueues (slots rotate through all TSIM_PIPELINE_SLOTS; tsim_synchronize, or tsim_sample_batch_device_end on the
 * slots, completes the results); all batches share B, num_f, shot_offset and `flags` (TSIM_PIPE_*; the inputs are
 * ordered after the handle's stream unless TSIM_PIPE_INPUTS_READY).  d_max_norm_dev may be NULL, and so may its
 * entries.  Where the register first pass applies (f rows of at most 128 bits, at most 64 outputs and 16 compiled
 * outputs, pattern tables on, short hard-row lists) the first passes of up to 8 batches are ONE grid
 * (k_sample_lw_multi: no kernel boundary, no host call between batches, one ramp and one tail per group) and their
 * hard rows one k_sample4h_multi batch behind it; any other program or launch plan goes batch by batch through
 * tsim_sample_batch_device_begin.  TSIM_AMD_FUSED_STEPS=0 forces the latter, TSIM_AMD_FUSED_MAX caps a group. */
int tsim_sample_steps_device(tsim_program *p, int32_t n_steps, const uint64_t *const *d_f, int64_t B, int32_t num_f,
                             uint32_t key[2], int64_t shot_offset, void *const *d_out, float *const *d_max_norm_dev,
                             uint32_t flags);

/* HIP-event timing of the sampling kernel launches on the handle's stream.
 * on = 1: every kernel of a launch; on = 2: only the first kernel of a launch (the pattern-table
 * pass when tables are active) - timing events drain the queue they are recorded on, which costs
 * pipelined launches ~10 us per step when every side-stream kernel is bracketed.            */
int tsim_profile_enable(tsim_program *p, int32_t on);
/* Bracket only one launch in `every` (default 1): each timing event costs a queue drain.        */
int tsim_profile_set_sampling(tsim_program *p, int32_t every);
/* Sum of kernel durations (ms) and number of launches since the last reset;
 * synchronises the stream.                                                   */
int tsim_profile_read(tsim_program *p, double *kernel_ms, int64_t *launches, int32_t reset);
/* The same total split by kernel: [0] pattern-table pass (k_sample_lw), [1] hard-row kernel
 * (k_sample4h), [2] full kernel (k_sample4 / k_sample); call before a resetting tsim_profile_read. */
int tsim_profile_read_stages(tsim_program *p, double stage_ms[3]);
/* Batches covered by the bracketed first passes since the last reset: a fused first pass
 * (tsim_sample_steps_device) is ONE launch in tsim_profile_read that serves several batches. */
int tsim_profile_read_steps(tsim_program *p, int64_t *steps, int32_t reset);

/* ---- introspection ---------------------------------------------------- */

int tsim_program_info(const tsim_program *p, int32_t *n_components, int32_t *num_outputs,
                      int64_t *image_bytes, int64_t *total_graphs, int64_t *total_rows);

/* packer statistics: out[0] fast formulation selected, [1] levels, [2] fixed-frame levels,
 * [3] product pairs, [4] counted NodePhases rows, [5] table entries, [6] graphs with tabled
 * PhasePairs, [7] low 4 bits: full-evaluation kernel: 0 row kernel, 1 LDS chunk tables (k_sample4), 2 sparse columns for wide
 * components (k_sample4w); + 32: the program has a wide record (k_sample_wide serves it); + 16: with the shared column table;
 * + 64: the packer could NOT rule out that the reference's int32 running sum of some level wraps (exact_scalar.py:74-84,173-189) - the
 * exact formulation then differs from the reference on the inputs where the reference's own arithmetic wraps; TSIM_AMD_MODE=faithful
 * (every program) or =strict (exactly these programs) mirrors the wrap on the int32-faithful formulation */
int tsim_program_stats(const tsim_program *p, int64_t out[8]);

/* Which kernel family served the launches of this handle so far (diagnostics: scripts/shape_map.py, tests of the
 * eligibility rules - the reference takes every shape through one code path, src/tsim/sampler.py:117-167; here the shape
 * picks the kernel).  out[k] = launches of family k since creation / the last reset, TSIM_PATH_COUNT entries:
 * 0 k_sample_lw_fast (fused group), 1 k_sample_lw_fastm, 2 k_sample_lw_multi, 3 k_direct_multi, 4 k_sample_wide,
 * 5 k_sample_lw_fast (one batch), 6 k_sample_lw_reg, 7 k_sample_lw<false>, 8 k_sample_lw<true>, 9 k_sample4w,
 * 10 k_sample4, 11 k_sample4h, 12 k_sample_hw, 13 k_sample4_over, 14 row kernel (k_sample<W>), 15 k_sample4h_multi,
 * 16 k_sample_gen (fused group, any shape). */
#define TSIM_PATH_COUNT 24
int tsim_program_path_counts(tsim_program *p, int64_t out[TSIM_PATH_COUNT], int32_t reset);

const char *tsim_last_error(void);
const char *tsim_version(void);
/* The keys TSIM_AMD_TUNE="key=value,..." understands, comma separated (the launch planner's A/B switches; results never depend
 * on them - tests/test_gpu_knobs.py runs an oracle slice under every one).  No counterpart in the reference. */
const char *tsim_tune_keys(void);

#ifdef __cplusplus
}
#endif
#endif /* TSIM_HIP_H */
