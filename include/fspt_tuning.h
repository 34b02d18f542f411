/*
 * fspt_tuning.h — scheduling knobs and measurement hooks of libfspt.
 *
 * Nothing here has a counterpart in the reference (one fragment-shader invocation walks a whole path,
 * tracer.fs:436-518; its whole state is two accumulators and two ray textures, main.js:598-617) and nothing here can
 * change a rendered value: every setting gives bit-identical results.  The drop-in boundary is include/fspt.h; a host
 * that only wants what main.js does never includes this file.  bench.py, the tests and tools/ do.
 */
#ifndef FSPT_TUNING_H
#define FSPT_TUNING_H

#include "fspt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Execution strategy of fspt_trace / fspt_render:
 *   1 "wavefront, batches" (default): primary -> [trace <-> logic] x rounds [-> tail] -> resolve, queue-driven kernels
 *      over batch_ticks ticks at a time (0 keeps the current batch size; default and max 128); path state = every
 *      (pixel, tick) of a batch, 216 bytes each;
 *   2 "wavefront, stream": the same kernels over a FIXED pool of live paths that is kept full (path regeneration
 *      between launches): every launch is pool-sized whatever the call's tick count, path state is the pool
 *      (fspt_target_set_pool) and a ring of finished samples; a run covers batch_ticks (<= 128) ticks;
 *   0 "megakernel": one persistent kernel per tick (path regeneration in place), no path state in memory. */
int fspt_target_set_pipeline(fspt_target *target, int pipeline, uint32_t batch_ticks);
/* Stream scheduler: `paths` = live paths each state set holds (0 = default, 16 Mi; 204 bytes per path; never more than
 * the call's samples); `drain_iterations` = trace/logic iterations after the last one that generated samples before
 * the tail kernel runs the rest to completion (-1 = default); `max_iterations` caps the iterations of a run (0 = no
 * cap; a test hook: the finishing launch then generates what the cursor has not handed out); `overlap` = 1: plan /
 * primary / resolve of an iteration on a second HIP stream beside the previous iteration's trace, 0: one stream,
 * -1: default. */
int fspt_target_set_pool(fspt_target *target, uint32_t paths, int drain_iterations, uint32_t max_iterations, int overlap);
/* Suspended traversals: a trace launch ends on its longest ray.  A wave that can get no more rays walks on for `steps`
 * traversal steps, then writes its unfinished traversals (node, t, hit, stack) to records and ends; the next trace
 * launch resumes them first (the path lags a round, at most four times).  0 = never; default 24.  Not used by the
 * counting kernel variants. */
int fspt_target_set_trace_budget(fspt_target *target, uint32_t steps);
/* When the batch scheduler hands the remaining live paths to the tail kernel (one launch that alternates traversal and
 * shading per path until it ends): -1 (default) decides from the previous batch's live-path counts and measured
 * launch times, 0 never (except for paths that refraction keeps alive beyond NUM_BOUNCES rounds, tracer.fs:488),
 * r >= 1 after round r. */
int fspt_target_set_tail(fspt_target *target, int round);
/* Node form of the traversal, per kernel class.  Every interior node has a 64-byte record (the boxes of its two
 * children: one of the reference's traversal steps, tracer.fs:372-392, per memory round trip) and - when
 * fspt_scene_create found every box of the tree to be the exact union of its children's boxes, which holds for every
 * tree bvh.js builds (bvh.js:120-126) - a 128-byte two-LEVEL record (the boxes of the four grandchildren, from which the
 * children's are derived exactly): two steps per round trip at twice the requests per fetch, the same nodes visited in
 * the same order.  The first is faster where the vector-memory request rate binds (large trace launches on a
 * cache-resident scene), the second where a launch is a bundle of dependent chains (tail kernel, small trace launches,
 * scenes beyond the L2).  primary / trace / tail: -1 the library's choice, 0 the 64-byte nodes, 1 the two-level nodes;
 * tail = 2: ADAPTIVE - the 64-byte nodes while a wave can refill its lanes from the list of live paths, the two-level
 * nodes from then on (a traversal changes form in mid-walk: node references and stack entries mean the same in both);
 * trace_below >= 0: the library's choice for a trace launch is "two-level when it expects fewer paths than this" (from
 * the previous batch's live-path counts); < 0 keeps the current threshold.  Ignored on a scene without two-level nodes
 * and by the counting kernel variants. */
int fspt_target_set_node_form(fspt_target *target, int primary, int trace, int tail, int64_t trace_below);
/* Whether the scene has two-level nodes, and their size in bytes (either pointer may be NULL). */
int fspt_scene_two_level_nodes(const fspt_scene *scene, int *present, uint64_t *bytes);
/* fspt_intersect (fspt.h) walking the two-level nodes (two_level != 0; FSPT_E_INVALID when the scene has none): t, index
 * and the per-ray step / leaf counts must equal the one-level walk's (tests). */
int fspt_intersect_form(fspt_scene *scene, int two_level, const float *rays, uint32_t n, float *t_out, int32_t *index_out,
                        uint32_t *steps_out, uint32_t *leaves_out);
/* The primary launch (ray generation + the camera ray's traversal + its shading) has two forms of its traversal phase:
 * 1 = one traversal per lane (a wave waits for its longest ray), 2 = per-lane refill over 2 x 64 samples per wave.  0
 * (default): the batch scheduler times both on the target's own batches (HIP events around the launch, read back
 * without waiting): per batch size the first batch runs the form the scene's size suggests, the second the other one,
 * a third batch is spent only when the first form lost by no more than its cold start explains; then the faster form
 * runs - form 1 on the 70 k-triangle scene, form 2 on the 1 M-triangle one. */
int fspt_target_set_primary_form(fspt_target *target, int form);
/* The form the next batch of `batch_ticks` ticks will use and what has been measured for that batch size so far
 * (best ms per sample of form 1, form 2; < 0: not measured yet).  Blocking (waits for a measurement in flight). */
int fspt_target_get_primary_form(fspt_target *target, uint32_t batch_ticks, int *form, double ms_per_sample[2]);
/* fspt_trace executes at once (0) instead of being recorded and batched (1, default; fspt.h: fspt_camera). */
int fspt_target_set_deferred(fspt_target *target, int enable);
/* Cap on the target's path-state bytes (0 = none): state sets, ray results, finished samples and suspension records
 * together.  A batch that does not fit the cap - or the free device memory - is halved, down to 8 ticks (or the call's
 * own tick count, if that is less); a frame that cannot hold that much runs its calls on the STREAM scheduler instead,
 * whose path state is a fixed pool sized to the cap (its runs shortened until two units of the pool fit); suspension
 * records that would take more than a quarter of the cap are not used.  Results never depend on any of it.
 * FSPT_E_NOMEM when not even a pool of two one-tick units (128 paths) fits. */
int fspt_target_set_memory_limit(fspt_target *target, uint64_t bytes);
/* Path-state bytes currently allocated by this target (state sets, ray results, finished samples, suspension
 * records) and the batch size in use (after any halving). */
int fspt_target_path_state_bytes(fspt_target *target, uint64_t *bytes, uint32_t *batch_ticks);
/* Allocate (and touch) the path state for the current resolution / shard / batch_ticks now instead of lazily inside
 * the first fspt_trace / fspt_render.  Blocking. */
int fspt_target_prepare(fspt_target *target);
/* Live paths after wavefront round r (r = 1: the primary launch) as a fraction of the batch's samples, from the most
 * recent batch: frac[r] for r < n_rounds (frac[0] unused).  Blocking. */
int fspt_target_live_paths(fspt_target *target, double *frac, uint32_t n_rounds);
/* Memory fspt_scene_create may spend on INTERLEAVED material textures (process-wide; scenes created afterwards;
 * default 8 GiB): a material that samples two or more image layers at one uv (tracer.fs:453-456) gets one image with
 * 16-byte texels, so that a shading event's 16 taps lie in ~2 cache lines instead of ~6; materials beyond the budget
 * read their layers from single-layer images. */
int fspt_set_texture_interleave_budget(uint64_t bytes);

/* ---- measurement ---------------------------------------------------------------------------------------------------- */
/* Per-launch stage timing: a HIP event pair around every kernel launch of the wavefront pipeline, what fspt_last_stage_ms
 * reads (default on).  The events are not free - two timestamp markers per launch, ~30 launches per 20-tick batch: 1.3 % of
 * a 20-tick region (profiles/r05/ab_stage_events.log) - so a host that only wants frames switches them off; the events
 * around the whole call (fspt_last_kernel_ms) stay.  With them off fspt_last_stage_ms reports zeros. */
int fspt_target_set_stage_timing(fspt_target *target, int enable);
/* HIP-event time of the most recent fspt_trace / fspt_render on this target (total ms, kernel launches).  Blocking. */
int fspt_last_kernel_ms(fspt_target *target, float *ms, uint32_t *launches);
/* ... per kernel class {primary, trace, logic, resolve, tail}: summed HIP-event durations and launch counts.  With
 * the stream scheduler's two HIP streams the classes overlap (their sum exceeds fspt_last_kernel_ms).  Blocking. */
int fspt_last_stage_ms(fspt_target *target, float ms[5], uint32_t launches[5]);
/* Of the traversal steps counted since fspt_counters_reset, how many k_wf_trace served from its LDS copy of the top of
 * the tree instead of the vector-memory pipeline (bench.py's request-rate roofline). */
int fspt_get_trace_lds_steps(fspt_target *target, uint64_t *steps);

/* Free and total memory of a device as the HIP runtime reports it (hipMemGetInfo): what a host sizes its batches against,
 * and how the tests check that dropped tracers give their memory back. */
int fspt_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- test hook ------------------------------------------------------------------------------------------------------ */
/* Device-side evaluation of the deterministic math primitives (DESIGN.md
 * "fspt-math"), for bitwise comparison against the oracle's C versions.
 * op: see FSPT_MATH_* ; a, b: n inputs each (b may be NULL for unary ops). */
enum {
  FSPT_MATH_SIN = 0, FSPT_MATH_COS = 1, FSPT_MATH_ATAN2 = 2, FSPT_MATH_ASIN = 3,
  FSPT_MATH_EXP2 = 4, FSPT_MATH_DIV = 5, FSPT_MATH_SQRT = 6, FSPT_MATH_RND = 7,
  FSPT_MATH_FRACT = 8, FSPT_MATH_LOG2 = 9, FSPT_MATH_POW = 10
};
int fspt_math_eval(int device, int op, const float *a, const float *b, uint32_t n,
                   float *out);

#ifdef __cplusplus
}
#endif
#endif /* FSPT_TUNING_H */
