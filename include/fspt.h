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

/* 4 (round 6): the entry points rounds 5-6 added (fspt_intersect_form, fspt_scene_two_level_nodes, fspt_target_set_node_form,
 * fspt_target_set_stage_timing, fspt_target_shard_slots / _pack_tiles / _unpack_tiles, fspt_multi_set_exchange /
 * _get_exchange, fspt_multi_last_stage_ms) and two changes of behaviour: fspt_target_destroy DROPS recorded ticks (a host
 * that wants them executed calls fspt_sync first - PathTracer.close() does), and under fspt_target_set_memory_limit a frame
 * that cannot hold 8 ticks of path state runs on the stream scheduler's bounded pool.  The bindings compare
 * fspt_abi_version() with this value when they load the library. */
#define FSPT_ABI_VERSION 4

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
  const float *bvh;       /* bvhTex (main.js:369-370,272-282): 9 words per node, pre-order: left, right, triStart as  */
  uint32_t n_nodes;       /* raw int32 bits in float slots (maskBVHBuffer) | min.xyz | max.xyz                        */
  const float *tri;       /* triTex (main.js:374): 9 floats per triangle v1 v2 v3, leaf order                          */
  uint32_t n_tris;
  const float *mat;       /* matTex (main.js:377-382): 12 floats per triangle [diffuseLayer emissiveLayer normalLayer  */
                          /* | mrLayer 0 0 | emittance.rgb | ior dielectric 0]                                         */
  const float *norm;      /* normTex (main.js:383-385): 27 floats per triangle, per vertex n, t, bt                    */
  const float *uv;        /* uvTex (main.js:386): 6 floats per triangle                                                */
  const uint8_t *atlas;   /* texArray (main.js:548-559): RGBA8, atlas_res^2 * atlas_layers texels, layer-major, GL rows */
  uint32_t atlas_res;
  uint32_t atlas_layers;
  const uint8_t *env;     /* envTex (main.js:170-180): RGBE in RGBA8, env_w * env_h texels; NULL = black (main.js:303-307) */
  uint32_t env_w;
  uint32_t env_h;
  const uint32_t *bins;   /* radianceBins (tracer.fs:21, env_sampler.js:73): n_bins x (x0,y0,x1,y1); n_bins >= 1       */
  uint32_t n_bins;
  uint32_t leaf_size;     /* '#define LEAF_SIZE' spliced into the shader (main.js:45,895)                              */
} fspt_scene_desc;

/* device = HIP device ordinal.  Builds the MI355X-native layouts (64-byte
 * two-child nodes, 144-byte leaf records, 192-byte hit records, material texture sets in 128-byte tiles). */
int fspt_scene_create(const fspt_scene_desc *desc, int device, fspt_scene **out);
int fspt_scene_destroy(fspt_scene *scene);
/* Maximum depth of the uploaded tree (root = 0); sizes the LDS stacks. */
int fspt_scene_depth(const fspt_scene *scene, uint32_t *depth);

/* ------------------------------------------------------------------------
 * Render target.  Replaces initBuffers (main.js:598-617): the RGBA32F screen textures (a single accumulator here: each
 * pixel reads/writes only itself, tracer.fs:516-517) and the two RGBA32F camera textures.
 * Sharding: tile x tile pixel tiles dealt round-robin (tile index % n_shards == shard); a target traces
 * only its own tiles and leaves every other pixel of its full-size accumulator at zero (a sum over shards = the frame).
 * ---------------------------------------------------------------------- */
int fspt_target_create(fspt_scene *scene, uint32_t width, uint32_t height,
                       fspt_target **out);
int fspt_target_destroy(fspt_target *target);
int fspt_target_set_shard(fspt_target *target, uint32_t shard, uint32_t n_shards,
                          uint32_t tile);
/* Use caller-owned device memory (W*H*4 floats, e.g. a torch tensor) as the accumulator so that a collective can run
 * on it in place; NULL restores the library's own buffer.  The bound buffer is only current after fspt_sync,
 * fspt_read_radiance or fspt_draw (recorded ticks, the library's own streams); re-binding flushes the recorded ticks
 * into the old buffer, fspt_target_destroy DROPS them and never writes to a caller-owned buffer. */
int fspt_target_bind_accumulator(fspt_target *target, void *device_ptr);
/* The size the target was created with (fspt_read_radiance / fspt_draw write W*H*4 elements). */
int fspt_target_size(fspt_target *target, uint32_t *width, uint32_t *height);
/* Device pointer of the accumulator currently in use (W*H*4 floats). */
int fspt_target_accumulator(fspt_target *target, void **device_ptr);

/* drawCamera (main.js:741-756) -> camera.fs:37-46.  lens = lensFeatures = [1 - 1/focalDepth, apertureSize].
 * Like WebGL's draw calls the two-call form is DEFERRED: fspt_camera records its arguments, fspt_trace records the
 * tick, and recorded ticks run - consecutive ticks with an unchanged view as ONE wavefront batch - when something
 * observes or changes what they depend on (fspt_read_*, fspt_draw, fspt_sync, fspt_clear, fspt_get_counters, every
 * setter, fspt_render, fspt_set_rays) or when a batch of them has accumulated.  Results are bit-identical; an error of
 * a deferred tick is reported by the call that flushes it.  (fspt_tuning.h: fspt_target_set_deferred.) */
int fspt_camera(fspt_target *target, const float P[3], const float I[3],
                float fov_scale, const float lens[2], float rand_base);
/* Inject ray buffers instead (W*H*4 floats each, rows bottom-up; traced at once, from the buffers). */
int fspt_set_rays(fspt_target *target, const float *pos, const float *dir);
int fspt_read_rays(fspt_target *target, float *pos, float *dir);

/* drawTracer(i) (main.js:758-807) -> tracer.fs:436-518.  One sample for every
 * pixel from the current ray buffers, running-mean accumulate with weight
 * tick.  num_bounces is tracer.fs:9's compile-time NUM_BOUNCES made a
 * run-time argument (reference value 4).                                     */
int fspt_trace(fspt_target *target, uint32_t tick, float rand_base,
               float env_theta, uint32_t num_bounces);
/* Refraction does not advance the bounce counter (tracer.fs:488 `i--`: the reference's loop is unbounded); libfspt
 * ends every path after FSPT_MAX_BOUNCES loop iterations, and treats a larger num_bounces as that. */
#define FSPT_MAX_BOUNCES 64

/* drawTracer in the reference's `mode=test` (main.js:879-883: bvh_test.fs): the camera ray's traversal-loop
 * iterations x 0.001 folded into the running mean (bvh_test.fs:224-232; no clamp). */
int fspt_trace_test(fspt_target *target, uint32_t tick);

/* tick() loop (main.js:838-857): n_ticks x (drawCamera + drawTracer) from tick first_tick, Math.random()*10000
 * (main.js:748,777) replaced by the xorshift64* stream below seeded with `seed`: two draws per tick, camera first.
 * Results are identical to the fspt_camera + fspt_trace pair. */
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
/* The host PRNG of fspt_render: state' = xorshift64*(state); returns float(u >> 40) * 2^-24 * 10000 in [0, 10000). */
float fspt_rand_base_next(uint64_t *state);

/* gl.viewport(0, 0, w, h) of drawCamera / drawTracer (main.js:744,761; resScale 0.25 while dragging, main.js:840):
 * only pixels x < w, y < h are traced, the rest keeps its contents.  0, 0 restores the whole target. */
int fspt_target_set_viewport(fspt_target *target, uint32_t w, uint32_t h);

/* Several GPUs (one frame cut into 32x32 tiles over the devices of a node; the reference has nothing here - README.md:28
 * lists "Tiled rendering" as a TODO): include/fspt_multi.h, included at the end of this file. */

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

/* intersectScene (tracer.fs:366-404) as a stand-alone entry: n rays (origin xyz, dir xyz) -> closest hit t and
 * triangle index (-1 = miss, t = 1e5), optionally loop-iteration and leaf-visit counts per ray.  Host pointers. */
int fspt_intersect(fspt_scene *scene, const float *rays, uint32_t n, float *t_out,
                   int32_t *index_out, uint32_t *steps_out, uint32_t *leaves_out);

/* Work counters for the byte accounting of SURVEY 8d, summed since the last fspt_clear / fspt_counters_reset while
 * counting is enabled (slower kernel variants).  enable = 1 counts the REFERENCE's work (NEE shadow rays traced to
 * their closest hit like tracer.fs:501: equal to the oracle's counters), 2 the production kernels' (shadow rays stop
 * at the first hit - only `shadow.index == -1` is consumed, tracer.fs:502); radiance is identical. */
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
int fspt_counters_reset(fspt_target *target);

/* ------------------------------------------------------------------------
 * Scene pipeline (host side, CPU): obj_loader.js + bvh.js + the packing loops of initBVH with the
 * same decisions in the same float64 arithmetic (1M triangles take the JS builder 2.5 min / 4 GB).
 * ---------------------------------------------------------------------- */
typedef struct fspt_prop_desc {
  const double *rotate; /* one scene-JSON prop (obj_loader.js:20-38): n_rotate x (axis.xyz, angle) in order, ... */
  uint32_t n_rotate;
  double scale;         /* ... then scale, then translate */
  double translate[3];
  uint32_t normals_mode; /* 0 = "flat"/default, 1 = "smooth", 2 = "mesh" (obj_loader.js:144,196) */
  double diffuse_layer, emissive_layer, normal_layer, mr_layer; /* getMaterial (main.js:206-270): atlas layer ids etc. */
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
/* parseMesh (obj_loader.js:6-215) for one prop: v / vt / vn / f, fan triangulation, negative indices, transforms,
 * normals, tangents. */
int fspt_builder_add_obj(fspt_builder *b, const char *obj_text, size_t len, const fspt_prop_desc *prop);
/* The same in two steps, for OBJs whose `usemtl` groups carry their own materials (mtl_loader.js, getMaterial): parse
 * (scene.worldTransforms, prop.skips), list the groups in the reference's iteration order (Object.entries: array-index
 * names first), let the host resolve one material per group, commit.  mtllib = ordinal of the `mtllib` line current at
 * the group's first face (-1: none). */
int fspt_builder_parse_obj(fspt_builder *b, const char *obj_text, size_t len, const fspt_prop_desc *prop,
                           const fspt_world_transform *world, uint32_t n_world,
                           const char *const *skips, uint32_t n_skips, uint32_t *n_groups);
int fspt_builder_group_info(const fspt_builder *b, uint32_t group, const char **name,
                            uint32_t *n_tris, int32_t *mtllib);
int fspt_builder_mtllib_name(const fspt_builder *b, uint32_t index, const char **name);
int fspt_builder_commit_obj(fspt_builder *b, const fspt_group_material *mats, uint32_t n_groups);
/* scene.normalize (main.js:337-348): centre on the scene bounds and scale the longest side to 2*size. */
int fspt_builder_normalize(fspt_builder *b, double size);
/* new BVH(geometry, leaf_size) + serializeTree + packing loops (bvh.js:5-91, main.js:355-392). */
int fspt_builder_build(fspt_builder *b, uint32_t leaf_size);
int fspt_builder_counts(const fspt_builder *b, uint32_t *n_nodes, uint32_t *n_tris, uint32_t *depth);
/* shootAutoFocusRay (main.js:447-546) on the built tree, in float64: distance along (eye, dir) to the first
 * triangle, 1e6 when there is none; the reference then sets lensFeatures[0] = 1 - 1/dist. */
int fspt_builder_autofocus(const fspt_builder *b, const double eye[3], const double dir[3], double *dist);
/* Copies the packed reference-layout arrays (bvh 9*n_nodes, tri 9*n_tris, mat 12*n_tris, norm 27*n_tris, uv 6*n_tris). */
int fspt_builder_get(const fspt_builder *b, float *bvh, float *tri, float *mat, float *norm, float *uv);
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
#include "fspt_multi.h"
#endif /* FSPT_H */
