// fspt_device.hpp — device data layouts + launch parameter blocks shared by
// fspt_kernels.hip (device code) and fspt_api.cpp (host side of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fspt {

// Traversal references.  ref >= 0: interior node index (native numbering,
// root = 0); ref < 0: leaf, leaf record = ~ref; REF_SENTINEL: empty stack.
constexpr int REF_SENTINEL = (int)0x80000000;

constexpr float MAX_T = 100000.0f;       // tracer.fs:10
constexpr float EPSILON = 0.000001f;     // tracer.fs:11
constexpr float M_PI_F = 3.14159265f;    // tracer.fs:12
constexpr float M_TAU_F = M_PI_F * 2.0f; // tracer.fs:13
constexpr float INV_PI_F = 1.0f / M_PI_F;
constexpr int MAX_PATH_ITERS = 64;       // cap on tracer.fs:488's unbounded i--

// Textures (environment, atlas layers with res > 1) are stored in tiles of 8 x 4 texels = 128 bytes = one cache line
// (rows of tiles, a tile's texels row-major), images padded to whole tiles.
#ifndef FSPT_TEX_TILE_W_LOG2
#define FSPT_TEX_TILE_W_LOG2 3
#define FSPT_TEX_TILE_H_LOG2 2
#endif
constexpr int TEX_TILE_W_LOG2 = FSPT_TEX_TILE_W_LOG2, TEX_TILE_H_LOG2 = FSPT_TEX_TILE_H_LOG2; // 0, 0 = plain row-major (A/B)
constexpr int TEX_TILE_W = 1 << TEX_TILE_W_LOG2, TEX_TILE_H = 1 << TEX_TILE_H_LOG2;

#ifndef FSPT_ENV_APRON
#define FSPT_ENV_APRON 1 // the environment map in overlapping tiles (fspt_kernels.hip env_taps); 0: disjoint tiles (A/B)
#endif
constexpr uint32_t LAYER_CONST = 0xFFFFFFFFu;
// Material texture set (3 x uint4 per set, DScene::tex_sets; a triangle's hit record names its set): the four atlas
// layers a shading event samples at one uv (tracer.fs:453-456: diffuse, emissive, metallic-roughness, normal), resolved
// at scene creation (layer = clamp(floor(id + 0.5)), the same binary32 arithmetic):
//   [0] = kind, base of the interleaved image (in 128-byte tiles), 0, 0
//   [1] = the four layers' texel where the layer is constant (every flat colour)
//   [2] = SEPARATE form: per layer the base of its tiled single-layer image (in 128-byte tiles) or LAYER_CONST
constexpr uint32_t TEXSET_CONST = 0u;    // four flat colours: no fetch at all
constexpr uint32_t TEXSET_SEPARATE = 1u; // image layers fetched from single-layer images (8 x 4-texel tiles)
constexpr uint32_t TEXSET_QUAD = 2u;     // the four layers interleaved texel by texel: 4 x 2-texel tiles of 16-byte texels

constexpr int NODE_F4 = 4;   // 64-byte two-child node = 4 x float4
constexpr int QUAD_F4 = 8;   // 128-byte two-LEVEL node (below) = 8 x float4 = one cache line
constexpr int TRI_FLOATS = 9;  // pre-edged triangle: v1, e1 = v2 - v1, e2 = v3 - v1 (tracer.fs:301-302 precomputed)
constexpr int HITREC_F4 = 12; // 192-byte hit record (shading) = 12 x float4 = exactly 3 cache lines

// Node (64 B), the slab bounds of both children in the pairs the packed-FP32 box test consumes (fspt_kernels.hip node_test):
//                f4[0] = lmin.xy lmax.xy   f4[1] = rmin.xy rmax.xy
//                f4[2] = lmin.z lmax.z rmin.z rmax.z   f4[3] = (int) left_ref right_ref 0 0
//   (tracer.fs:374-378 fetches the header of `current` and then, dependently,
//    the boxes of both children: 3 round trips; here one.)
// Two-level node ("quad", 128 B = ONE cache line, one per interior node N with children L, R; DScene::quads): what a lane
//   needs to take TWO of the reference's traversal steps (tracer.fs:372-392 at idx = N, then at idx = the child it
//   descends into) on one memory round trip - for the launches that are bundles of DEPENDENT chains (tail kernel, small
//   trace launches, the 1 M-triangle scene), where a step costs a cache-miss latency, not a request slot:
//                f4[0..2] = the node record of L (the boxes of L's children LL, LR, in the layout above)
//                f4[3]    = (int) LL_ref LR_ref L_ref R_ref
//                f4[4..6] = the node record of R (boxes of RL, RR)      f4[7] = (int) RL_ref RR_ref 0 0
//   The boxes of L and R themselves - what the step at N tests - are NOT stored: bvh.js builds every node's box from ITS
//   triangles' vertices (BoundingBox.addNode, bvh.js:120-126), so box(L) = union(box(LL), box(LR)) exactly, in binary32
//   (min / max are exact and rounding is monotone); the lane derives them with 12 v_min / v_max.  fspt_scene_create
//   VERIFIES that identity bit for bit on every interior node of the arrays it is given and builds no quads otherwise
//   (DScene::quads = NULL: every launch then uses the 64-byte nodes).  A child that is a leaf has no children: its part
//   holds its own box twice (the union is the box).  Same nodes tested against the same t in the same order, the far
//   child pushed as a child, the far grandchild as a grandchild: the node sequence, the counters and every result are
//   those of the one-level walk.
// Leaf record (LEAF_SIZE x 36 B = 144 B for LEAF_SIZE 4): the LEAF_SIZE consecutive triangles the reference's processLeaf
//   reads from the leaf's first one (tracer.fs:355-364 - a leaf with fewer triangles reads on into its successors, the
//   last leaves into the "-1" padding, main.js:150-152), pre-edged, COMPONENT-major: floats [c * LEAF_SIZE + i] =
//   component c (v1.xyz e1.xyz e2.xyz) of triangle i.  For LEAF_SIZE 4 that is nine 16-byte loads whose halves are the
//   operand pairs of the two-triangles-at-once intersection (ray_tri2).  A leaf's ref is ~(its record index);
//   slot_tri[record * LEAF_SIZE + i] = that triangle's index in the reference's order (what tracer.fs:360 reports).
// HitRec (192 B, 64-byte aligned = exactly 3 cache lines per shaded hit), one per leaf SLOT - the traversal's hit index
//   addresses it directly, no translation on the path (a triangle that is also an earlier leaf's over-read has a copy there):
//   floats 0..8 the same v1,e1,e2; 9..35 normTex record (n,t,bt per vertex,
//   main.js:383-385); 36..41 uv (main.js:386); 42 (uint bits) the material texture set of the triangle's four
//   layer ids (main.js:377-379); 46 ior; 47 dielectric.
struct DScene {
  const float4 *nodes;
  const float4 *quads;  // two-level nodes, indexed like `nodes` (NULL: the scene's boxes are not unions of their children's)
  const float *leaves;  // leaf records (see above)
  const uint32_t *slot_tri; // leaf slot -> triangle index
  const float4 *hitrec; // 12 x float4 per triangle
  const uint32_t *atlas; // RGBA8 texels of the single-layer images (SEPARATE sets), each tiled (TEX_TILE_*)
  const uint4 *atlas4;   // interleaved four-layer images (QUAD sets)
  const uint4 *tex_sets; // 3 x uint4 per material texture set
  const uint32_t *env;   // RGBE texels, tiled (NULL = black default environment)
  const uint4 *bins;
  uint32_t atlas_res, atlas_layers;
  uint32_t n_tex_sets;
  uint32_t env_w, env_h;
  uint32_t n_bins;
  uint32_t leaf_size;
  int root_ref;
  uint32_t stack_n; // LDS stack entries per lane (tree depth + 1)
  uint32_t n_top;   // nodes [0, n_top) are the top of the tree in breadth-first order
};

struct CameraP {
  float P[3];
  float I[3];
  float fov_scale;
  float lens[2];
};

struct TraceP {
  DScene scene;
  uint32_t W, H;
  uint32_t vw, vh; // gl.viewport(0, 0, vw, vh): only these pixels are drawn (main.js:744,761)
  uint32_t tick;
  float rand_base;     // drawTracer's randBase
  float rand_base_cam; // drawCamera's randBase (fused ray generation only)
  float env_theta;
  uint32_t num_bounces;
  CameraP cam;
  const float4 *ray_pos; // ray buffers (two-call form)
  const float4 *ray_dir;
  float4 *accum;
  uint32_t *work_counter;       // zeroed before the launch
  unsigned long long *counters; // 6 x u64 (COUNT variants) or NULL
  // sharding
  uint32_t shard, n_shards, tile;
  uint32_t tiles_x, tiles_y, n_owned_tiles;
};

struct IntersectP {
  DScene scene;
  const float *rays; // 6 floats per ray
  uint32_t n;
  uint32_t wide;     // walk the two-level nodes (scene.quads must exist)
  float *t_out;
  int *index_out;
  uint32_t *steps_out;
  uint32_t *leaves_out;
};

// ---------------------------------------------------------------------------
// Wavefront pipeline (fspt_render's default): the same per-path arithmetic cut
// into queue-driven kernels so that every lane of a wave does the same kind of
// work:  primary (ray generation + camera-ray traversal + its shading) -> [ trace <-> logic ] x rounds
//        [-> tail: the last few live paths run to completion in one kernel] -> resolve.
//   slot s = w * n_batch + j  : sample of tick (first_tick + j) for work index w
//   (pixel via work_to_pixel); a batch holds n_batch ticks.  Pixel-major on purpose:
//   the ticks of one pixel are neighbours in the primary launch, so a wave's camera rays are
//   near-identical (coherent traversal, broadcast node loads) and shade the same triangle.
// Path state is DENSE: round r's logic launch writes the state of its survivors to consecutive indices
//   k = 0 .. n_r-1 of state set (r & 1) (the other set is its input), so every later access - the trace kernel's
//   ray loads and result stores, the next round's loads - touches whole cache lines.  (Round 1 kept the state at
//   the path's fixed slot: after a few rounds the survivors are sparse and every 16-byte access moved a 64-byte
//   line - 4x the bytes, profiles/r01.)  float4 SoA arrays, per set:
//   A  ro.xyz, slot (uint bits)          B  rd.xyz, flags (bits: 0-7 bounce, 8-15 iters, 16 primary, 17 hasShadow, 18 colour is +0)
//   C  accumulatedReflectance.xyz, weights.y
//   E  color.xyz, -   (only touched while the colour is non-zero)
//   D  envDir.xyz, weights.x             P  reflectance*envThroughput.xyz, -     (only when the path has a NEE shadow ray)
// Per round (indexed like the state): hit (t, index) of the extension ray, shadow_hit = index of the NEE ray.
// fin[slot]: the finished sample colour, slot-major (a wave's finishing paths are neighbours).
// ---------------------------------------------------------------------------
constexpr int WF_MAX_BATCH = 128;
constexpr int WF_HIT_TERMINAL = -2; // hit[k].index of an extension ray that hit something but whose path has no bounce left
constexpr int WF_HIT_PENDING = -3;  // hit[k].index of a path whose traversal was SUSPENDED by the trace launch (hit[k].t = its record)
constexpr uint32_t WF_FLAG_PRIMARY = 1u << 16, WF_FLAG_SHADOW = 1u << 17, WF_FLAG_COLZERO = 1u << 18;
// Suspended traversals (k_wf_trace): what a trace launch costs beyond its work is its LONGEST ray - a dependent chain of
// up to a few hundred node fetches that a handful of lanes walk while the rest of the chip idles (~0.2 ms per launch,
// profiles/r03).  A wave that can get no more work and has been walking for WfP::susp_budget steps writes the state of
// its unfinished rays (node, t, hit, the LDS stack) to a record and ends; k_wf_carry (in front of the logic launch) moves
// such a path on to the next state set unchanged (flag WF_FLAG_SUSP: the next trace launch must not start its rays afresh), and the next
// trace launch resumes the records FIRST, beside its bulk of new rays.  Same traversal, same result; a path lags one
// round per suspension (at most WF_LAG_MAX: the count lives in bits 20-22 of the flags).
constexpr uint32_t WF_FLAG_SUSP = 1u << 19;
#ifndef FSPT_WF_LAG_MAX
#define FSPT_WF_LAG_MAX 4
#endif
constexpr uint32_t WF_LAG_SHIFT = 20, WF_LAG_MASK = 7u, WF_LAG_MAX = FSPT_WF_LAG_MAX;
constexpr int WF_SUSP_HEADER = 8; // ints before the stack entries of a record: state index, node, t, hit, sp | ray << 8, shadow result, -, -
#ifndef FSPT_WF_HEADS
#define FSPT_WF_HEADS 16
#endif
constexpr int WF_HEADS = FSPT_WF_HEADS; // work-pool heads of the trace kernel (one per item segment; waves steal from the others)

#ifndef FSPT_WF_HEAD_STRIDE
#define FSPT_WF_HEAD_STRIDE 64 // uint32 words between two pool heads (256 bytes)
#endif
constexpr int WF_HEAD_STRIDE = FSPT_WF_HEAD_STRIDE;
struct alignas(128) WfCounts { // one per round, zeroed before the batch
  uint32_t n_ext;          // live paths written by this round's primary / logic launch
  uint32_t n_susp;         // traversals this round's trace launch suspended (records for the next one)
  uint32_t pad[30];
};
// Trace / tail kernel work-pool heads: WF_HEADS per round, each in a memory line (and channel) of its own
// (WfP::heads[(round * WF_HEADS + stripe) * WF_HEAD_STRIDE]): device-scope atomics execute at the memory side and
// same-LINE atomics serialise like same-address ones (~15 ns each) - 16 heads packed in one 64-byte line behaved
// exactly like a single head (profiles/r02/ab_trace_pool_heads.log).

struct WfSet { float4 *A, *B, *C, *E, *D, *P; };

// ---------------------------------------------------------------------------
// Streaming scheduler (fspt_target_set_pipeline code 2): a FIXED pool of path state that is kept full.
//   The (work index, tick) samples of a run of n_batch ticks are numbered pixel-major exactly like a batch's slots
//   (g = w * n_batch + j) and cut into UNITS of 64 work indices x n_batch ticks (one 8x8 pixel patch, all its ticks).
//   Iteration i:   plan(i)    takes as many units from the device-side cursor as are GUARANTEED to fit the state set:
//                             cap - (live paths of set i-1)  - every live path may survive, every new sample may too
//                  primary(i) camera rays + their traversal + shading for those units; survivors -> set i&1
//                  logic(i)   one S step for the live paths of set (i-1)&1; survivors -> set i&1 (same counter)
//                  trace(i)   the rays of set i&1
//                  resolve    folds the units whose every path has ended (generated >= num_bounces iterations ago;
//                             MAX_PATH_ITERS when a material can refract) into the accumulator, ticks in order
//   so every trace / logic launch is about the size of the pool whatever the call's tick count, paths of all ages
//   mixed; path state is cap x 204 bytes + a ring of finished colours, not (pixels x ticks) x 216 bytes.  plan /
//   primary / resolve run on a second stream beside the previous iteration's trace.  After the last iteration the tail
//   kernel runs whatever is alive to completion AND generates whatever the cursor has not handed out (the host only
//   estimates the iteration count; nothing depends on the estimate but speed).
//   fin is a RING of ring_slots entries (a multiple of the unit size, > everything in flight); the slot id a path
//   carries is its ring position (slot % n_batch is still its tick).
// ---------------------------------------------------------------------------
constexpr int WF_RING = 4;    // iterations whose counters / pool heads exist at the same time (even: set parity = index parity)
constexpr int WF_HIST = 128;  // cursor history kept for the resolve (> MAX_PATH_ITERS + 2)
struct alignas(128) WfStreamCtl {
  uint32_t cursor;       // next unit nobody has generated yet (may overshoot the run's units, work_total / 64, at the very end)
  uint32_t reserved;
  uint32_t n_iters;      // statistics of the run, read back by the host without waiting (tunes its iteration estimate)
  uint32_t last_gen_it;  // last iteration that took units
  uint32_t max_live;
  uint32_t fin_gen_units; // units the finishing tail launch had to generate itself
  uint32_t pad[26];
  unsigned long long sum_live;
  unsigned long long pad2[15];
  uint32_t plan_start[WF_RING], plan_units[WF_RING]; // iteration i's chunk of units
  uint32_t hist[WF_HIST]; // cursor after plan(i), i % WF_HIST
};

struct WfP {
  uint32_t gen_rays; // primary launch: 1 = camera.fs in the kernel, 0 = read the ray buffers (two-call form)
  uint32_t lds_top; // top-of-tree nodes k_wf_trace keeps in LDS (<= scene.n_top; set by launch_wf)
  DScene scene;
  WfSet set[2];
  float *fin;  // finished sample colours [slot][3]
  float2 *hit;
  int *shadow_hit;
  WfCounts *counts;
  uint32_t *heads; // pool heads, see above
  uint32_t round;  // batch scheduler: the round (1 = primary); stream: the iteration
  // which counter / state set a launch reads and writes (batch: round - 1 / round; stream: iteration mod WF_RING)
  uint32_t cnt_in, cnt_out, set_in, set_out;
  WfStreamCtl *ctl;     // NULL = batch scheduler
  uint32_t ring_slots;  // entries of the fin ring (batch: n_batch * work_total = no wrap)
  uint32_t cap;         // stream: paths a state set holds
  uint32_t take_max;    // stream: units plan(i) takes at most
  int res_from, res_to; // resolve (stream): fold the units between hist[res_from] (< 0: unit 0) and hist[res_to] (-2: all)
  uint32_t finish;      // tail (stream): also generate what the cursor has not handed out
  uint32_t serial;      // plan (stream): it runs AFTER logic(i) (one HIP stream) and sees the real survivors in counts[cnt_out]
  int *susp[2];         // suspended-traversal records: trace(i) writes susp[cnt_out & 1], resumes susp[cnt_in & 1]
  uint32_t susp_stride; // ints per record (WF_SUSP_HEADER + stack entries, a multiple of 4)
  uint32_t susp_budget; // traversal steps a wave walks on after its last refill before it suspends (0: never)
  // batch scheduler, resolve launch (the last kernel of a batch): block 0 copies the rounds' live-path counts to
  // `live_out` (pinned host memory: the tail heuristic's statistics) and clears the counters and pool heads of
  // `zero_rounds` rounds for the next batch - instead of two fill commands and a copy command behind every batch
  uint32_t *live_out;
  uint32_t zero_rounds;
  uint32_t carry_blocks; // logic launch: this many trailing blocks do k_wf_carry's work instead (one launch less per round)
  uint32_t tail_slice;    // tail kernel: traversal steps of a lane per T phase (0: WF_TAIL_SLICE); the host picks it by scene size
  uint32_t tail_adaptive; // tail kernel with its `wide` bit set: two-level nodes only once the wave's list is used up (k_wf_tail WIDE = 2)
  uint32_t wide;        // bit k: kernel class k (WF_K_PRIMARY / WF_K_TRACE / WF_K_TAIL) walks the two-level nodes (scene.quads; same results)
  uint32_t primary_r;   // k_wf_primary: 1 = one traversal per lane, 2 = per-lane refill over 2 x 64 samples per wave (same results)
  uint32_t W, H;
  uint32_t vw, vh; // viewport, as in TraceP
  uint32_t work_total; // work indices per tick (owned tiles * tile^2)
  uint32_t n_batch;    // ticks in this batch (<= WF_MAX_BATCH)
  uint32_t first_tick;
  float rb_cam[WF_MAX_BATCH];
  float rb_trace[WF_MAX_BATCH];
  float env_theta;
  uint32_t num_bounces;
  CameraP cam;
  const float4 *ray_pos, *ray_dir; // ray buffers (two-call form, n_batch == 1)
  float4 *accum;
  unsigned long long *counters;
  uint32_t shard, n_shards, tile, tiles_x, tiles_y, n_owned_tiles;
};

// kernel classes; also the slots of fspt_last_stage_ms
enum { WF_K_PRIMARY = 0, WF_K_TRACE = 1, WF_K_LOGIC = 2, WF_K_RESOLVE = 3, WF_K_TAIL = 4, WF_K_KINDS = 5 /* timed classes */, WF_K_PLAN = 6 /* not timed */, WF_K_CARRY = 7 /* not timed */ };
// count: 0 = production kernels; 1 = counting variants doing the reference's work (NEE shadow rays traced to the closest
// hit, tracer.fs:501); 2 = counting variants of the production work (shadow rays stop at the first hit)
hipError_t launch_wf(int kernel, const WfP &p, int count, int num_cus, hipStream_t stream);

// multi-device read-out: a shard's own pixels <-> a packed array in work-index order (work_to_pixel)
struct TilePackP {
  float4 *accum;   // full-size W x H accumulator
  float4 *packed;  // n_owned_tiles * tile^2 entries (channels 3: that many RGB triples, alpha = 1 on unpack, tracer.fs:517)
  uint32_t channels; // 0 / 4: RGBA, 3: RGB
  uint32_t W, H, vw, vh;
  uint32_t shard, n_shards, tile, tiles_x, tiles_y, n_owned_tiles;
};
hipError_t launch_tile_pack(const TilePackP &p, bool unpack, hipStream_t stream);

// launchers (fspt_kernels.hip)
hipError_t launch_trace(const TraceP &p, bool gen_rays, bool count, int num_cus, hipStream_t stream);
size_t wf_max_stack_entries(); // deepest tree (entries per lane) whose traversal stacks fit the LDS of every kernel
hipError_t launch_camera(uint32_t W, uint32_t H, uint32_t vw, uint32_t vh, const CameraP &cam, float rand_base, float4 *pos, float4 *dir,
                         hipStream_t stream);
hipError_t launch_intersect(const IntersectP &p, hipStream_t stream);
hipError_t launch_bvh_test(const TraceP &p, hipStream_t stream);
hipError_t launch_draw(const float4 *acc, uint32_t W, uint32_t H, float exposure, float saturation, int denoise,
                       float max_sigma, float scale, uint32_t *out, hipStream_t stream);
hipError_t launch_math(int op, const float *a, const float *b, uint32_t n, float *out, hipStream_t stream);

} // namespace fspt
