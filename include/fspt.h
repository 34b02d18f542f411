/*
 * fspt.h — C ABI of libfspt, the MI355X-native replacement for the WebGL2
 * draw-call boundary of apbodnar/FSPT's path-trace hot path.
 *
 * The reference has no plugin/FFI interface: its hot path (shader/tracer.fs)
 * sits behind the WebGL2 calls issued by main.js. Every entry point below
 * names the reference call site it replaces (file:line in /root/reference).
 * INTEGRATION.md shows the N-API stub a maintainer of the reference would add
 * to main.js to call these instead of gl.*.
 *
 * Conventions
 *   - plain C, no C++/torch/HIP types in any signature;
 *   - every function returns 0 on success, <0 (FSPT_E_*) on error;
 *     fspt_last_error() returns a thread-local message (reference just
 *     console.log()s / throws: main.js:91-94, 564-569);
 *   - the library COPIES every host array it is given (caller may free after
 *     the call returns);
 *   - all calls for one target come from one host thread at a time (the reference
 *     is a single JS thread: main.js:838-857); work is enqueued on HIP streams -
 *     the two-call ticks even later, see fspt_camera - and only fspt_read_*,
 *     fspt_draw*, fspt_sync, fspt_get_counters and the fspt_last_* timers block;
 *   - there is NO CPU fallback: without a HIP device every device entry point
 *     fails with FSPT_E_NO_DEVICE.
 */
#ifndef FSPT_H
#define FSPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSPT_ABI_VERSION 2

enum {
  FSPT_OK = 0,
  FSPT_E_INVALID = -1,   /* bad argument / inconsistent sizes            */
  FSPT_E_NO_DEVICE = -2, /* no HIP device or HIP runtime error at init   */
  FSPT_E_HIP = -3,       /* HIP runtime error (message has the details)  */
  FSPT_E_NOMEM = -4,
  FSPT_E_PARSE = -5,     /* scene pipeline: malformed OBJ / image        */
  FSPT_E_STATE = -6      /* call order (e.g. trace before camera)        */
};

typedef struct fspt_scene fspt_scene;
typedef struct fspt_target fspt_target;
typedef struct fspt_builder fspt_builder;

/* ------------------------------------------------------------------------
 * Scene upload.  Replaces the texture uploads of initBVH (main.js:408-437),
 * initAtlas (main.js:548-560) and createEnvironmentMapImg (main.js:170-180).
 * Arrays are exactly what the reference hands to gl.texImage2D, WITHOUT the
 * padBuffer() padding (main.js:143-154) — pass the un-padded element counts.
 * ---------------------------------------------------------------------- */
typedef struct fspt_scene_desc {
  /* bvhTex (main.js:369-370,272-282): 9 x 32-bit words per node, pre-order,
   * [left:i32 right:i32 triStart:i32 | min.xyz | max.xyz]; the first three
   * words are raw int bits stored in float slots (maskBVHBuffer).            */
  const float *bvh;
  uint32_t n_nodes;
  /* triTex (main.js:374): 9 floats per triangle v1 v2 v3, leaf order.        */
  const float *tri;
  uint32_t n_tris;
  /* matTex (main.js:377-382): 12 floats per triangle
   * [diffuseLayer emissiveLayer normalLayer | mrLayer 0 0 | emittance.rgb |
   *  ior dielectric 0].                                                      */
  const float *mat;
  /* normTex (main.js:383-385): 27 floats per triangle, per vertex n,t,bt.    */
  const float *norm;
  /* uvTex (main.js:386): 6 floats per triangle.                              */
  const float *uv;
  /* texArray (main.js:548-559): RGBA8, atlas_res^2 * atlas_layers texels,
   * layer-major, row 0 first (GL order).                                     */
  const uint8_t *atlas;
  uint32_t atlas_res;
  uint32_t atlas_layers;
  /* envTex (main.js:170-180): RGBE in RGBA8, env_w * env_h texels, row 0
   * first.  NULL => black environment (main.js:303-307).                     */
  const uint8_t *env;
  uint32_t env_w;
  uint32_t env_h;
  /* radianceBins uniform (tracer.fs:21, env_sampler.js:73): n_bins x
   * (x0,y0,x1,y1).  n_bins must be >= 1 (main.js:292: [0,0,1,2048]).         */
  const uint32_t *bins;
  uint32_t n_bins;
  /* '#define LEAF_SIZE' spliced into the shader (main.js:45,895).            */
  uint32_t leaf_size;
} fspt_scene_desc;

/* device = HIP device ordinal.  Builds the MI355X-native layouts (64-byte
 * two-child nodes, 144-byte leaf records, 192-byte hit records, material texture sets in 128-byte tiles). */
int fspt_scene_create(const fspt_scene_desc *desc, int device, fspt_scene **out);
int fspt_scene_destroy(fspt_scene *scene);
/* Memory fspt_scene_create may spend on INTERLEAVED material textures (process-wide; applies to scenes created
 * afterwards; default 8 GiB).  A material that samples two or more image layers at one uv (tracer.fs:453-456) gets one
 * image with 16-byte texels (its four layers side by side), res^2 * 16 bytes, so that a shading event's 16 taps lie in
 * ~2 cache lines instead of ~6; materials beyond the budget (or bytes = 0) read their layers from single-layer images.
 * The rendered values do not depend on it. */
int fspt_set_texture_interleave_budget(uint64_t bytes);
/* Maximum depth of the uploaded tree (root = 0); sizes the LDS stacks. */
int fspt_scene_depth(const fspt_scene *scene, uint32_t *depth);

/* ------------------------------------------------------------------------
 * Render target.  Replaces initBuffers (main.js:598-617): two RGBA32F screen
 * textures (a single accumulator here: each pixel reads/writes only itself,
 * tracer.fs:516-517) and the two RGBA32F camera textures.
 *
 * Sharding: the frame is cut into tile x tile pixel tiles dealt
 * round-robin (tile index % n_shards == shard) to the shards; a target only
 * traces its own tiles and leaves every other pixel of its full-size
 * accumulator at zero, so that a sum-reduce over shards yields the frame.
 * n_shards = 1, shard = 0 traces everything.
 * ---------------------------------------------------------------------- */
int fspt_target_create(fspt_scene *scene, uint32_t width, uint32_t height,
                       fspt_target **out);
int fspt_target_destroy(fspt_target *target);
int fspt_target_set_shard(fspt_target *target, uint32_t shard, uint32_t n_shards,
                          uint32_t tile);
/* Use caller-owned device memory (e.g. a torch tensor's data_ptr, W*H*4
 * floats) as the accumulator so that a collective can run on it in place.
 * NULL restores the internally allocated buffer.  The library works asynchronously (recorded two-call ticks, kernels
 * on its own streams): the bound buffer is only current after fspt_sync, fspt_read_radiance or fspt_draw - anything
 * else that reads it (a torch op, a collective) must call fspt_sync first.  fspt_target_destroy executes the ticks
 * still recorded for a caller-owned accumulator before it lets go of it. */
int fspt_target_bind_accumulator(fspt_target *target, void *device_ptr);
/* The size the target was created with (what fspt_read_radiance / fspt_draw write: W*H*4 elements); bindings check
 * the caller's array against it (canvas.width/height of main.js:598-617). */
int fspt_target_size(fspt_target *target, uint32_t *width, uint32_t *height);
/* Device pointer of the accumulator currently in use (W*H*4 floats). */
int fspt_target_accumulator(fspt_target *target, void **device_ptr);

/* drawCamera (main.js:741-756) -> camera.fs:37-46.  Writes the pos/dir ray
 * buffers.  lens = lensFeatures = [1 - 1/focalDepth, apertureSize].
 *
 * DEFERRED EXECUTION of the two-call form.  The reference's tick() issues drawCamera + drawTracer and moves on; WebGL
 * runs them whenever it likes, and nothing is observable before the next read of a render target.  libfspt uses the
 * same freedom: fspt_camera records its arguments, fspt_trace records the tick, and the recorded ticks run - runs of
 * consecutive ticks with unchanged camera / envTheta / num_bounces as ONE wavefront batch, rays generated inside the
 * path kernel from the recorded randBase values - when something observes or changes what they depend on
 * (fspt_read_radiance, fspt_draw, fspt_read_rays, fspt_sync, fspt_clear, fspt_get_counters, every fspt_target_set_*,
 * fspt_render, fspt_set_rays), or when batch_ticks of them have accumulated.  A host loop of 128 tick()s therefore
 * costs what fspt_render(128) costs (4.8 Gsamples/s at 1920x1080 instead of 0.77 one launch set per tick).  Results
 * are bit-identical either way; an error of a deferred tick is reported by the call that flushes it.
 * fspt_target_set_deferred(target, 0) makes every fspt_trace execute at once.  Rays injected with fspt_set_rays are
 * always traced immediately, from the buffers. */
int fspt_camera(fspt_target *target, const float P[3], const float I[3],
                float fov_scale, const float lens[2], float rand_base);
int fspt_target_set_deferred(fspt_target *target, int enable);
/* Inject ray buffers instead (W*H*4 floats each, RGBA32F rows bottom-up) —
 * used to feed the GLSL oracle's camera output to the tracer. */
int fspt_set_rays(fspt_target *target, const float *pos, const float *dir);
int fspt_read_rays(fspt_target *target, float *pos, float *dir);

/* drawTracer(i) (main.js:758-807) -> tracer.fs:436-518.  One sample for every
 * pixel from the current ray buffers, running-mean accumulate with weight
 * tick.  num_bounces is tracer.fs:9's compile-time NUM_BOUNCES made a
 * run-time argument (reference value 4).                                     */
int fspt_trace(fspt_target *target, uint32_t tick, float rand_base,
               float env_theta, uint32_t num_bounces);
/* The refraction branch does not advance the bounce counter (tracer.fs:488 `i--`), so the reference's loop is
 * unbounded; libfspt ends every path after FSPT_MAX_BOUNCES loop iterations.  A num_bounces above
 * that therefore cannot change any sample: fspt_trace / fspt_render treat it as FSPT_MAX_BOUNCES. */
#define FSPT_MAX_BOUNCES 64

/* drawTracer in the reference's `mode=test` (main.js:879-883 swaps tracer.fs for bvh_test.fs): every
 * pixel's camera ray is traced once and the number of traversal-loop iterations x 0.001 is folded into
 * the accumulator's running mean (bvh_test.fs:224-232; no clamp).  Same ray buffers, accumulator, shard
 * and read-out as fspt_trace. */
int fspt_trace_test(fspt_target *target, uint32_t tick);

/* tick() loop (main.js:838-857): n_ticks x (drawCamera + drawTracer) starting
 * at tick first_tick, with Math.random()*10000 (main.js:748,777) replaced by
 * the documented xorshift64* stream seeded with seed: per tick two draws,
 * camera first.  Ray generation is fused into the path kernel (no ray-buffer
 * round trip); results are identical to the fspt_camera + fspt_trace pair.   */
typedef struct fspt_camera_params {
  float P[3];
  float I[3];
  float fov_scale;
  float lens[2];
  float env_theta;
  uint32_t num_bounces;
} fspt_camera_params;
int fspt_render(fspt_target *target, const fspt_camera_params *cam,
                uint32_t first_tick, uint32_t n_ticks, uint64_t seed);
/* The host PRNG used by fspt_render: state' = xorshift64*(state); returns a
 * float in [0,10000) as float(u >> 40) * 2^-24 * 10000.  Exposed so a host can
 * reproduce the stream for the two-call (camera + trace) form. */
float fspt_rand_base_next(uint64_t *state);

/* gl.viewport(0, 0, w, h) of drawCamera / drawTracer (main.js:744,761): only pixels x < w, y < h are generated and
 * traced; the rest of the ray buffers and of the accumulator keep their contents.  The reference shrinks the viewport
 * to resolution * 0.25 while the camera is being dragged (resScale, main.js:840) and shows that corner magnified
 * through draw.fs's `scale` (fspt_draw_scaled).  0, 0 restores the whole target. */
int fspt_target_set_viewport(fspt_target *target, uint32_t w, uint32_t h);

/* Execution strategy of fspt_trace / fspt_render (results are bit-identical).  The reference has one: a fragment
 * shader invocation per pixel and draw call (main.js:758-807); its whole path state is two accumulators and two ray
 * textures (main.js:598-617).
 *   pipeline 1 "wavefront, batches": primary -> [trace <-> logic] x rounds -> resolve, queue-driven
 *              kernels over batch_ticks ticks at a time (0 keeps the current batch size; default and max 128);
 *              path state = every (pixel, tick) of a batch;
 *   pipeline 3 "wavefront, stream": the same kernels over a FIXED pool of live paths that is kept full (path
 *              regeneration between launches): every launch is pool-sized whatever the call's tick count, path state
 *              is the pool (fspt_target_set_pool) and a ring of finished samples; a run covers up to 128 ticks;
 *   pipeline 4 "wavefront, stream, two pools": pipeline 3 on two halves of the frame's 8x8 patches, interleaved on
 *              separate HIP streams;
 *   pipeline 0 "megakernel": one persistent kernel per tick (path regeneration in place);
 *   pipeline 2 "wavefront, batches, two lanes": pipeline 1 with the batch split in two halves that run
 *              concurrently on two HIP streams (separate path state, resolves chained in tick order). */
int fspt_target_set_pipeline(fspt_target *target, int pipeline, uint32_t batch_ticks);
/* Stream scheduler (pipeline 3 / 4): `paths` = live paths each state set of a pool holds (0 = default, 16 Mi; 204 bytes
 * per path; never more than the call's samples); `drain_iterations` = trace/logic iterations after the last one that
 * generated samples before the tail kernel runs the rest to completion (-1 = default); `max_iterations` caps the
 * iterations of a run (0 = no cap; a test hook: the finishing launch then generates what the cursor has not handed
 * out); `overlap` = 1: plan / primary / resolve of an iteration on a second HIP stream, beside the previous
 * iteration's trace, 0: one stream, -1: default.  None of them changes a result. */
int fspt_target_set_pool(fspt_target *target, uint32_t paths, int drain_iterations, uint32_t max_iterations, int overlap);
/* Wavefront pipelines: a trace launch ends on its longest ray - up to a few hundred dependent node fetches walked by a
 * handful of lanes.  A wave that can get no more rays walks on for `steps` traversal steps, then writes the state of its
 * unfinished traversals (node, t, hit, stack) to memory and ends; the next trace launch resumes them first, beside its
 * new rays (the path lags a round, at most four times).  Same traversal, same result.  0 = never suspend; the default
 * is 24.  Not used by the counting kernel variants.  (The reference has no counterpart: one fragment-shader invocation
 * walks its whole path, tracer.fs:436-518.) */
int fspt_target_set_trace_budget(fspt_target *target, uint32_t steps);
/* Wavefront pipelines: who finishes the paths that END in a round (two in three: the extension ray left the scene -
 * environment lookup, tracer.fs:509-512 - or the bounce budget is used up; the NEE result before that, :500-505).
 * 0: the logic kernel, beside its shading; 1: a kernel of its own (8 waves/SIMD instead of the 4 the shading code's
 * registers allow) in front of the logic launch; 2: the same on a second HIP stream beside the logic launch (batch
 * scheduler).  Same arithmetic, same result. */
int fspt_target_set_finish_kernel(fspt_target *target, int mode);
/* Wavefront path state lives in device memory: 216 bytes per (pixel, tick) of a batch.  It is sized for the largest
 * n_ticks any call on this target has asked for so far (at most batch_ticks; fspt_trace = 1 tick = 0.46 GB at
 * 1920x1080, a 128-tick fspt_render = 58 GB) and grows when a longer call arrives.  fspt_target_set_memory_limit caps
 * it (bytes; 0 = no cap): a batch that does not fit the cap - or the free device memory - is halved until it does,
 * which only costs speed (results do not depend on the batch size).  FSPT_E_NOMEM when even one tick does not fit.
 * The stream scheduler (pipeline 3 / 4) holds a pool instead - 204 bytes per pool path + a ring of finished samples,
 * 3.4 + 1.5 GB at the default 16 Mi paths whatever the frame and tick count - and the limit shrinks the pool
 * (FSPT_E_NOMEM below two units of 64 pixels x the call's ticks). */
int fspt_target_set_memory_limit(fspt_target *target, uint64_t bytes);
/* Path-state bytes currently allocated by this target and the batch size in use (after any halving). */
int fspt_target_path_state_bytes(fspt_target *target, uint64_t *bytes, uint32_t *batch_ticks);
/* Live paths after wavefront round r (r = 1: the primary launch) as a fraction of the batch's samples, from the most
 * recent batch: frac[r] for r < n_rounds (frac[0] unused).  What the adaptive tail setting decides on.  Blocking. */
int fspt_target_live_paths(fspt_target *target, double *frac, uint32_t n_rounds);
/* Allocate (and touch) the pipeline's path-state buffers for the current resolution / shard / batch now,
 * (sized for the full configured batch_ticks) instead of lazily inside the first fspt_trace / fspt_render.  Blocking. */
int fspt_target_prepare(fspt_target *target);
/* Per-kernel-class timing of the most recent fspt_trace / fspt_render (wavefront pipeline):
 * summed HIP-event durations and launch counts for {primary, trace, logic, resolve, tail}: primary = the first launch
 * of a batch (ray generation + the camera ray's traversal + its shading in one kernel), trace / logic = the later
 * rounds' traversal and shading launches, resolve = the running-mean fold, tail = the kernel that runs the last live
 * paths to completion. Blocking. */
int fspt_last_stage_ms(fspt_target *target, float ms[5], uint32_t launches[5]);
/* When the wavefront pipeline hands the remaining live paths to the tail kernel (one launch that alternates traversal
 * and shading per path until it ends, instead of one trace + one logic launch per bounce): -1 (default) decides from
 * the live-path counts of the previous batch, 0 never (except for paths that refraction keeps alive beyond
 * NUM_BOUNCES rounds, tracer.fs:488), r >= 1 after round r.  Results are bit-identical for every setting. */
int fspt_target_set_tail(fspt_target *target, int round);

/* ------------------------------------------------------------------------
 * One frame over several GPUs of a node, driven by ONE host thread (the reference's host is a single JS thread;
 * README.md:28 lists "tiled rendering" as a TODO).  fspt_multi_create uploads the scene to every listed device and
 * makes one render target per device that owns every n_devices-th 32x32 tile (fspt_target_set_shard; the RNG depends
 * on pixel coordinates and randBase only - camera.fs:38, tracer.fs:458 - so the assembled frame is bit-identical to a
 * single-GPU render).  The draw calls below enqueue on every device and return; NO data moves between devices while
 * rendering.  fspt_multi_read_radiance / fspt_multi_draw do the one exchange: every device packs its own tiles, the
 * packed tiles travel to devices[0] with peer-to-peer copies (xGMI) on the devices' own streams, devices[0] scatters
 * them into its full-size accumulator.  A device may be listed more than once (its share of the tiles is then traced
 * by several targets one after the other) - that is how a 1-GPU box exercises the path.
 * ---------------------------------------------------------------------- */
typedef struct fspt_multi fspt_multi;
int fspt_multi_create(const fspt_scene_desc *desc, const int *devices, uint32_t n_devices,
                      uint32_t width, uint32_t height, fspt_multi **out);
int fspt_multi_destroy(fspt_multi *m);
/* the per-device target i (for fspt_target_set_pipeline / _set_tail / _prepare / counters); owned by m */
int fspt_multi_target(fspt_multi *m, uint32_t i, fspt_target **out);
int fspt_multi_camera(fspt_multi *m, const float P[3], const float I[3], float fov_scale, const float lens[2],
                      float rand_base);                                     /* drawCamera on every device   */
int fspt_multi_trace(fspt_multi *m, uint32_t tick, float rand_base, float env_theta, uint32_t num_bounces);
int fspt_multi_render(fspt_multi *m, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks,
                      uint64_t seed);                                       /* fspt_render on every device  */
int fspt_multi_clear(fspt_multi *m);
int fspt_multi_sync(fspt_multi *m);
/* Gather (see above), then what fspt_read_radiance / fspt_draw do on the assembled frame.  Blocking. */
int fspt_multi_read_radiance(fspt_multi *m, float *out);
int fspt_multi_draw(fspt_multi *m, float exposure, float saturation, int denoise, float max_sigma, uint8_t *out_rgba8);
/* Bytes that crossed between devices in the most recent gather (the exchange's payload: 16 bytes per foreign pixel). */
int fspt_multi_last_gather_bytes(fspt_multi *m, uint64_t *bytes);
int fspt_multi_size(fspt_multi *m, uint32_t *width, uint32_t *height);  /* the frame fspt_multi_create was given */
/* How target i's tiles reach devices[0] (the reference has no counterpart: README.md:28 "Tiled rendering" is a TODO):
 * bit 0 = target i's device can write devices[0]'s memory directly (hipDeviceCanAccessPeer(devices[i], devices[0]):
 * the direction the gather copy runs, issued on the sending device's stream), bit 1 = the reverse mapping.  0 = the
 * copy is staged through the host.  A target on devices[0] itself reports 3. */
int fspt_multi_peer_access(fspt_multi *m, uint32_t i, int *mask);

/* clear() (main.js:826-836). */
int fspt_clear(fspt_target *target);
int fspt_sync(fspt_target *target);
/* What draw.fs:87 reads: RGBA32F, W*H*4 floats, row 0 = bottom, a = 1.
 * Blocking (syncs the stream first). */
int fspt_read_radiance(fspt_target *target, float *out);

/* drawQuad (main.js:809-824) -> draw.fs:82-93: exposure, ACES fit, saturation, gamma 1/2.2 and the
 * optional 5x5 firefly filter (draw.fs:52-80, max_sigma = the `sigma` slider) on the current
 * accumulator; writes what the canvas would hold: RGBA8, W*H*4 bytes, row 0 = bottom.  Blocking. */
int fspt_draw(fspt_target *target, float exposure, float saturation, int denoise, float max_sigma,
              uint8_t *out_rgba8);
/* The same with draw.fs's `scale` uniform (draw.fs:59,87: texel = ivec2(gl_FragCoord * scale)); the reference
 * draws with scale 0.25 while the camera is being dragged (main.js:819,840), 1.0 otherwise. */
int fspt_draw_scaled(fspt_target *target, float exposure, float saturation, int denoise,
                     float max_sigma, float scale, uint8_t *out_rgba8);

/* ------------------------------------------------------------------------
 * intersectScene (tracer.fs:366-404) as a stand-alone entry: n rays
 * (origin xyz, dir xyz: 6 floats each) -> closest hit t and triangle index
 * (-1 = miss, t = 1e5), optionally the loop-iteration and leaf-visit counts
 * per ray (what bvh_test.fs:173-231 visualises).  Host pointers.
 * ---------------------------------------------------------------------- */
int fspt_intersect(fspt_scene *scene, const float *rays, uint32_t n, float *t_out,
                   int32_t *index_out, uint32_t *steps_out, uint32_t *leaves_out);

/* Work counters for the byte accounting of SURVEY 8d, summed over every
 * sample traced since the last fspt_clear / fspt_counters_reset when
 * counting is enabled (slower kernel variants; off by default).  enable = 1 counts the REFERENCE's work: NEE shadow
 * rays are traced to their closest hit like tracer.fs:501 does, so the counters equal the oracle's.  enable = 2 counts
 * the work of the production kernels, whose shadow rays stop at the first hit (only `shadow.index == -1` is consumed,
 * tracer.fs:502): fewer steps / leaves, everything else - and every radiance value - identical. */
typedef struct fspt_counters {
  uint64_t samples;     /* tracer.fs main() invocations                      */
  uint64_t rays;        /* intersectScene calls            (tracer.fs:366)   */
  uint64_t steps;       /* while(idx>-1) iterations        (tracer.fs:373)   */
  uint64_t leaves;      /* processLeaf calls               (tracer.fs:380)   */
  uint64_t shades;      /* bounce-loop iterations          (tracer.fs:446)   */
  uint64_t env_lookups; /* envSample calls                 (tracer.fs:416)   */
} fspt_counters;
int fspt_enable_counters(fspt_target *target, int enable);
int fspt_get_counters(fspt_target *target, fspt_counters *out);
/* Measurement only (bench.py's request-rate roofline of the trace kernel): of the traversal steps counted since
 * fspt_counters_reset, how many k_wf_trace served from its LDS copy of the top of the tree instead of the vector-memory
 * pipeline.  No counterpart in the reference. */
int fspt_get_trace_lds_steps(fspt_target *target, uint64_t *steps);
int fspt_counters_reset(fspt_target *target);

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

/* Timing of the most recent fspt_trace / fspt_render on this target, measured
 * with HIP events on the target's own stream around the path-trace kernel(s):
 * total milliseconds and number of kernel launches. Blocking. */
int fspt_last_kernel_ms(fspt_target *target, float *ms, uint32_t *launches);

/* ------------------------------------------------------------------------
 * Scene pipeline (host side, CPU).  A native equivalent of
 * obj_loader.js + bvh.js + the packing loops of initBVH, making the same
 * decisions in the same float64 arithmetic, for scenes too large for the
 * JS builder (1M triangles: 2.5 min / 4 GB in Node).
 * ---------------------------------------------------------------------- */
typedef struct fspt_prop_desc {
  /* transforms of one scene-JSON prop (obj_loader.js:20-38): rotations
   * (axis xyz, angle) applied in order, then scale, then translate.         */
  const double *rotate; /* n_rotate x 4: axis.x axis.y axis.z angle          */
  uint32_t n_rotate;
  double scale;
  double translate[3];
  /* normals: 0 = "flat"/default, 1 = "smooth", 2 = "mesh" (obj_loader.js:144,196) */
  uint32_t normals_mode;
  /* resolved material (getMaterial, main.js:206-270): atlas layer ids etc.  */
  double diffuse_layer, emissive_layer, normal_layer, mr_layer;
  double emittance[3];
  double ior, dielectric;
} fspt_prop_desc;

/* one element of scene.worldTransforms (obj_loader.js:26-36): `if (t.rotate) rotations (positions and
 * normals) else if (t.translate) translation (positions only)`; has_rotate = the key is present. */
typedef struct fspt_world_transform {
  const double *rotate; /* n_rotate x 4: axis.x axis.y axis.z angle */
  uint32_t n_rotate;
  uint32_t has_rotate;
  double translate[3];
  uint32_t has_translate;
} fspt_world_transform;

/* getMaterial's result for one OBJ group (main.js:206-270, packed by main.js:376-382) */
typedef struct fspt_group_material {
  double diffuse_layer, emissive_layer, normal_layer, mr_layer;
  double emittance[3];
  double ior, dielectric;
} fspt_group_material;

int fspt_builder_create(fspt_builder **out);
int fspt_builder_destroy(fspt_builder *b);
/* parseMesh (obj_loader.js:6-215) for one prop: v / vt / vn / f lines, fan
 * triangulation, negative indices, per-prop transforms, normals, tangents. */
int fspt_builder_add_obj(fspt_builder *b, const char *obj_text, size_t len,
                         const fspt_prop_desc *prop);
/* The same in two steps, for OBJs whose groups (`usemtl`) carry their own materials (mtl_loader.js,
 * getMaterial main.js:206-270): parse (scene.worldTransforms, prop.skips; the material fields of `prop` are
 * ignored), list the groups in the reference's iteration order (Object.entries: array-index names first),
 * let the host resolve one material per group, commit.  mtllib = ordinal of the `mtllib` line that was
 * current when the group's first face was read (-1: none; the group's material is then `{}`). */
int fspt_builder_parse_obj(fspt_builder *b, const char *obj_text, size_t len, const fspt_prop_desc *prop,
                           const fspt_world_transform *world, uint32_t n_world,
                           const char *const *skips, uint32_t n_skips, uint32_t *n_groups);
int fspt_builder_group_info(const fspt_builder *b, uint32_t group, const char **name,
                            uint32_t *n_tris, int32_t *mtllib);
int fspt_builder_mtllib_name(const fspt_builder *b, uint32_t index, const char **name);
int fspt_builder_commit_obj(fspt_builder *b, const fspt_group_material *mats, uint32_t n_groups);
/* scene.normalize (main.js:337-348): centre on the scene bounds and scale the longest side to 2*size. */
int fspt_builder_normalize(fspt_builder *b, double size);
/* new BVH(geometry, leaf_size) + serializeTree + packing loops
 * (bvh.js:5-91, main.js:355-392). */
int fspt_builder_build(fspt_builder *b, uint32_t leaf_size);
int fspt_builder_counts(const fspt_builder *b, uint32_t *n_nodes, uint32_t *n_tris,
                        uint32_t *depth);
/* shootAutoFocusRay (main.js:447-546) on the built tree, in float64: distance along (eye, dir) to the first
 * triangle, 1e6 when there is none; the reference then sets lensFeatures[0] = 1 - 1/dist. */
int fspt_builder_autofocus(const fspt_builder *b, const double eye[3], const double dir[3], double *dist);
/* Copies the packed reference-layout arrays (sizes from fspt_builder_counts:
 * bvh 9*n_nodes, tri 9*n_tris, mat 12*n_tris, norm 27*n_tris, uv 6*n_tris). */
int fspt_builder_get(const fspt_builder *b, float *bvh, float *tri, float *mat,
                     float *norm, float *uv);
/* ProcessEnvRadiance (env_sampler.js:1-74) on raw RGBE bytes.  Writes up to
 * cap bins (4 uint32 each) and the real count to *n_bins. */
int fspt_env_bins(const uint8_t *rgbe, uint32_t w, uint32_t h, uint32_t *bins,
                  uint32_t cap, uint32_t *n_bins);

const char *fspt_last_error(void);
int fspt_abi_version(void);
/* Number of visible HIP devices (0 when none / no driver). */
int fspt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* FSPT_H */
