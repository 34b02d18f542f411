// fspt_api.cpp — host side of the libfspt C ABI (include/fspt.h).
//
// Mirrors the WebGL2 resource/draw-call layer of the reference's main.js:
// scene upload (initBVH 408-437, initAtlas 548-560), render targets
// (initBuffers 598-617), drawCamera (741-756), drawTracer (758-807), clear
// (826-836) and the tick loop (838-857).  No CPU fallback: every device entry
// point fails with FSPT_E_NO_DEVICE when there is no HIP device.
#include "../../include/fspt.h"
#include "../../include/fspt_tuning.h"
#include "fspt_device.hpp"

#include <cstdarg>
#include <cstdio>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <array>
#include <atomic>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#ifndef FSPT_NODE_TREELET
#define FSPT_NODE_TREELET 0 // nodes per treelet below the breadth-first top of the tree; 0 = pre-order (profiles/r02: A/B on the 1 M-triangle scene)
#endif

static thread_local char g_err[512] = "";

void fspt_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      fspt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
      return FSPT_E_HIP;                                                                          \
    }                                                                                             \
  } while (0)

struct fspt_scene {
  int device = 0;
  int num_cus = 256;
  fspt::DScene d{};
  void *nodes = nullptr, *quads = nullptr /* two-level nodes, or NULL */, *tris = nullptr /* leaf records */, *slot_tri = nullptr, *shade = nullptr, *atlas = nullptr, *atlas4 = nullptr, *tex_sets = nullptr, *env = nullptr, *bins = nullptr;
  uint32_t depth = 0, n_nodes = 0, n_tris = 0, n_interior = 0;
  bool has_dielectric = false; // some triangle can refract (tracer.fs:481-488: unbounded path length)
};

static const int WF_ARRAYS = 15;
#ifndef FSPT_SUSP_BUDGET
#define FSPT_SUSP_BUDGET 24 // profiles/r03/ab_trace_suspend_budget.log: 0 / 16 / 24 / 32 / 48 -> 3 883 / 3 938 / 3 940 / 3 935 / 3 921 Msamples/s in 20-step regions (same box, twice)
#endif
static const uint32_t ST_DEFAULT_SUSP_BUDGET = FSPT_SUSP_BUDGET;
// Library's choice of the node form (fspt_target::node_form = -1), per kernel class.  Measured (profiles/r05/ab_two_level_*.log,
// one box, interleaved): the two-level nodes LOSE in every regime they were built for - tail kernel 0.037 -> 0.040-0.045 ms
// per tick (C2, 20 ticks), 0.82 -> 0.80-0.92 (single tick), 0.125 -> 0.145 (1 M triangles); trace launches 0.170 -> 0.21 /
// 0.215 -> 0.26; primary 0.127 -> 0.138 / 0.176 -> 0.193.  Halving the dependent round trips buys nothing because a step's
// time is not a cache-miss latency: it is the CU's vector-memory front end working through the lane-requests of all its
// resident waves (16 waves x 4 instructions x (4.6 + 0.63 x active lanes) cycles = the 1 900 clocks per step round-4
// measured in the tail kernel), and a two-level fetch issues 8 requests where the walk needs 4 or 8.  So: everything off.
#ifndef FSPT_WIDE_PRIMARY
#define FSPT_WIDE_PRIMARY 0
#endif
#ifndef FSPT_WIDE_TAIL
#define FSPT_WIDE_TAIL 0
#endif
#ifndef FSPT_CARRY_BLOCKS
#define FSPT_CARRY_BLOCKS 4u // trailing blocks of a logic launch that do k_wf_carry's work (0: a separate launch per round, as in rounds 3-4)
#endif
#ifndef FSPT_RESOLVE_CLEARS
#define FSPT_RESOLVE_CLEARS 1 // the batch's resolve launch hands the live-path counts to the host and clears counters + pool heads (0: fill / copy commands)
#endif
#ifndef FSPT_WIDE_TRACE_BELOW
#define FSPT_WIDE_TRACE_BELOW 0u // paths
#endif
struct fspt_target {
  fspt_scene *scene = nullptr;
  uint32_t W = 0, H = 0;
  float4 *accum_own = nullptr;
  float4 *accum = nullptr;
  float4 *ray_pos = nullptr, *ray_dir = nullptr;
  bool rays_valid = false;
  uint32_t *work_counters = nullptr; // ring of zeroed work counters, one per launch
  uint32_t n_work_counters = 0;
  unsigned long long *counters = nullptr; // 6 x u64 on device
  int count = 0; // fspt_enable_counters: 0 off, 1 the reference's work, 2 the production kernels' work
  uint32_t shard = 0, n_shards = 1, tile = 32;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  uint32_t last_launches = 0;
  // wavefront pipeline
  uint32_t vw = 0, vh = 0;    // viewport (gl.viewport of the two draws); default = the whole target
  int pipeline = 1;           // 0 = megakernel, 1 = wavefront
  int sched = 0;              // wavefront pipeline: 0 = batch scheduler (all ticks x all pixels per batch), 1 = stream (fixed pool)
  uint32_t pool_paths = 0;    // stream: paths per state set and lane (0 = default)
  int stream_drain = -1;      // stream: iterations after the last generating one before the tail kernel takes over (-1 = default)
  uint32_t stream_iter_cap = 0; // stream, test hook: at most this many iterations per run (the finishing launch does the rest)
  uint32_t susp_budget = ST_DEFAULT_SUSP_BUDGET; // traversal steps a starved trace wave walks on before it parks its rays (0 = never)
  int stream_overlap = -1;      // stream: plan / primary / resolve on a second HIP stream beside the previous trace (1), everything on one stream (0), default (-1)
  uint32_t batch_ticks = 128; // ticks traced together by the wavefront pipeline (58 GB of path state at 1080p;
                              // measured 64 / 128 / 256 -> 3 619 / 3 794 / 3 750 Msamples/s, profiles/r01)
  // Path state and streams of the wavefront pipeline (either scheduler).  (Two such lanes with overlapped half-batches
  // were measured in rounds 1-3 and gained nothing worth their memory: profiles/r02, profiles/r03/ab_staggered_lanes.log.)
  struct WfLane {
    void *mem[WF_ARRAYS] = {};
    fspt::WfCounts *counts = nullptr;
    uint32_t *heads = nullptr;             // trace pool heads (fspt_device.hpp)
    fspt::WfCounts *counts_host = nullptr; // pinned copy of the last batch's per-round counts (tail heuristic)
    uint32_t *live_host = nullptr, *live_dev = nullptr; // ... or (counts_live) the live paths per round as the resolve launch wrote them: pinned host memory and its device address
    bool counts_live = false;
    hipEvent_t counts_ready = nullptr;
    bool counts_pending = false;
    uint32_t counts_slots = 0;             // slots of the batch the copy describes
    uint32_t slots = 0;        // allocated path slots (batch scheduler)
    hipStream_t stream = nullptr;
    hipEvent_t resolved = nullptr; // this lane's most recent resolve has finished
    // stream scheduler (fspt_device.hpp: WfStreamCtl): a pool of st_cap paths per state set + a ring of st_fin finished colours
    uint32_t st_cap = 0, st_fin = 0;
    hipStream_t stream_b = nullptr;          // plan / primary / resolve run here, beside the previous iteration's trace
    hipEvent_t ev_logic[fspt::WF_RING] = {}, ev_b[fspt::WF_RING] = {}, ev_run = nullptr, ev_b_last = nullptr;
    fspt::WfStreamCtl *ctl = nullptr;
    fspt::WfStreamCtl *ctl_host = nullptr;   // pinned copy of the last run's statistics (never waited for)
    hipEvent_t ctl_ready = nullptr;
    bool ctl_pending = false;
    uint64_t ctl_key = 0, stat_key = 0;      // what the pending copy / the known statistics describe (units, ticks, pool, bounces)
    uint32_t ctl_units = 0;                  // units of the run the pending copy describes
    uint32_t stat_gen_iters = 0;             // iterations the last such run needed to hand out all its units
    uint64_t bytes = 0;                      // path-state bytes this lane holds (either scheduler)
    int *susp[2] = {nullptr, nullptr};       // suspended-traversal records of the trace launches (fspt_device.hpp), ping-pong
    uint32_t susp_stride = 0;
    size_t susp_recs = 0;                    // records per buffer
    uint64_t susp_bytes = 0;                 // both buffers
    bool zeroed = false;                     // counts / heads / ctl are zero (cleared behind the previous batch, off the next one's critical path)
  } wf;
  // Deferred two-call ticks (fspt_camera + fspt_trace): recorded, executed in batches at the next flush point
  struct Deferred { fspt_camera_params cam; float rb_cam; uint32_t tick; float rb_trace; };
  std::vector<Deferred> pending;
  fspt_camera_params last_cam{}; // the most recent fspt_camera call (num_bounces / env_theta filled in by fspt_trace)
  float last_rb_cam = 0.0f;
  bool cam_recorded = false;     // last_cam is valid and newer than the ray buffers' contents
  bool rays_injected = false;    // the ray buffers hold caller-supplied rays (fspt_set_rays): trace them as they are
  bool defer = true;             // fspt_target_set_deferred
  // Primary-form tuner (batch scheduler): k_wf_primary has two forms of its traversal phase with identical results
  // (fspt_kernels.hip).  Which is faster depends on the scene and the batch size, so the target measures: HIP events
  // around the primary launch of a batch, read back without waiting at the start of a later batch.  Per batch size: the
  // first batch runs the form the scene's size suggests (X), the second the other one (Y), and as a rule that settles it
  // - see prim_choose for the one case that takes a third batch.
  int primary_form = 0;      // fspt_target_set_primary_form: 0 measure and choose, 1 / 2 forced
  // batch ticks -> [form] {best ms per sample so far (< 0: none), measurements taken}
  struct PrimStat { double best[3] = {-1.0, -1.0, -1.0}; uint32_t runs[3] = {0, 0, 0}; };
  std::map<uint32_t, PrimStat> prim_ms;
  hipEvent_t prim_ev[2] = {nullptr, nullptr};
  bool prim_pending = false;
  uint32_t prim_pending_form = 0, prim_pending_ticks = 0;
  double prim_pending_samples = 0.0;
  // Node form per kernel class (fspt_target_set_node_form): -1 the library's choice, 0 the 64-byte nodes, 1 the two-level
  // nodes (fspt_device.hpp "quad"; only where the scene has them).  [0] primary launch, [1] trace launches, [2] tail kernel.
  int node_form[3] = {-1, -1, -1};
  uint32_t wide_trace_below = FSPT_WIDE_TRACE_BELOW; // library's choice for a trace launch: two-level nodes when it expects fewer paths than this
  int tail_round = -1;       // fspt_target_set_tail: -1 adaptive, 0 never, r >= 1 after round r
  float live_frac[80] = {};  // live paths after round r / slots of the batch, from the most recent finished batch
  bool live_known = false;
  uint32_t ticks_seen = 0;   // largest n_ticks of any call so far: path state is sized for min(batch_ticks, ticks_seen)
  uint64_t mem_limit = 0;    // fspt_target_set_memory_limit: cap on the path-state bytes of this target (0 = none)
  hipEvent_t ev_start = nullptr;
  // per-launch stage timing (HIP events on the target's stream)
  std::vector<hipEvent_t> ev_pool;
  std::vector<int> ev_kind;   // kernel class of pair i
  uint32_t ev_used = 0;       // pairs used by the last render
  bool ev_overflow = false;
};

// RGBA8 image (row-major, w x h) -> 8 x 4-texel tiles (fspt_device.hpp: TEX_TILE_*), padded to whole tiles.
// Returns the number of texels of the tiled image; with src == nullptr only that.
// Bytes of interleaved four-layer texture images (TEXSET_QUAD) a scene may allocate; sets beyond it fetch their image
// layers from single-layer images (fspt_set_texture_interleave_budget).
static std::atomic<uint64_t> g_texset_budget{8ull << 30};

static size_t tile_image(const uint8_t *src, uint32_t w, uint32_t h, std::vector<uint32_t> &out) {
  const uint32_t tx = (w + fspt::TEX_TILE_W - 1) / fspt::TEX_TILE_W, ty = (h + fspt::TEX_TILE_H - 1) / fspt::TEX_TILE_H;
  const size_t n = (size_t)tx * ty * fspt::TEX_TILE_W * fspt::TEX_TILE_H;
  if (!src) return n;
  out.assign(n, 0u);
  for (uint32_t j = 0; j < h; ++j)
    for (uint32_t i = 0; i < w; ++i) {
      uint32_t v;
      std::memcpy(&v, src + ((size_t)j * w + i) * 4, 4);
      out[((size_t)(j / fspt::TEX_TILE_H) * tx + i / fspt::TEX_TILE_W) * (fspt::TEX_TILE_W * fspt::TEX_TILE_H) +
          (j % fspt::TEX_TILE_H) * fspt::TEX_TILE_W + (i % fspt::TEX_TILE_W)] = v;
    }
  return n;
}

static int flush_pending(fspt_target *t);
static void prim_reset(fspt_target *t);
static int materialise_rays(fspt_target *t);
#define FLUSH_OR_RETURN(t) do { int rc_f = flush_pending(t); if (rc_f) return rc_f; } while (0)

static int check_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    fspt_set_error("no HIP device available (%s); libfspt has no CPU fallback",
                   e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return FSPT_E_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    fspt_set_error("device %d out of range (have %d)", device, n);
    return FSPT_E_INVALID;
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) {
    fspt_set_error("hipSetDevice(%d): %s", device, hipGetErrorString(e));
    return FSPT_E_NO_DEVICE;
  }
  return FSPT_OK;
}

extern "C" {

const char *fspt_last_error(void) { return g_err; }
int fspt_abi_version(void) { return FSPT_ABI_VERSION; }

int fspt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

float fspt_rand_base_next(uint64_t *state) {
  uint64_t x = *state;
  x ^= x >> 12;
  x ^= x << 25;
  x ^= x >> 27;
  *state = x;
  uint64_t r = x * 2685821657736338717ULL;
  return ((float)(r >> 40) * (1.0f / 16777216.0f)) * 10000.0f;
}

// ---------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------
int fspt_set_texture_interleave_budget(uint64_t bytes) {
  g_texset_budget = bytes;
  return FSPT_OK;
}

int fspt_scene_create(const fspt_scene_desc *desc, int device, fspt_scene **out) {
  if (!desc || !out) { fspt_set_error("fspt_scene_create: NULL argument"); return FSPT_E_INVALID; }
  *out = nullptr;
  if (!desc->bvh || !desc->tri || !desc->mat || !desc->norm || !desc->uv || desc->n_nodes == 0 || desc->n_tris == 0) {
    fspt_set_error("fspt_scene_create: bvh/tri/mat/norm/uv must be non-empty");
    return FSPT_E_INVALID;
  }
  if (!desc->atlas || desc->atlas_res == 0 || desc->atlas_layers == 0) {
    fspt_set_error("fspt_scene_create: atlas must have at least one layer");
    return FSPT_E_INVALID;
  }
  if (!desc->bins || desc->n_bins == 0) {
    fspt_set_error("fspt_scene_create: radianceBins must hold at least one bin (main.js:292)");
    return FSPT_E_INVALID;
  }
  if (desc->env && (desc->env_w == 0 || desc->env_h == 0)) {
    fspt_set_error("fspt_scene_create: env given with zero size");
    return FSPT_E_INVALID;
  }
  if (desc->leaf_size == 0 || desc->leaf_size > 64) {
    fspt_set_error("fspt_scene_create: leaf_size %u out of range [1,64]", desc->leaf_size);
    return FSPT_E_INVALID;
  }
  const uint32_t N = desc->n_nodes, T = desc->n_tris;
  auto word = [&](uint32_t node, int w) -> int32_t {
    int32_t v;
    std::memcpy(&v, desc->bvh + (size_t)node * 9 + w, 4);
    return v;
  };
  // ---- validate + renumber interior nodes ------------------------------------------
  // The first TOP_BFS interior nodes in breadth-first order get the lowest numbers (every ray walks the top of
  // the tree: the traversal kernel keeps a prefix of them in LDS); the rest keep their pre-order.
  std::vector<int32_t> ref(N);
  std::vector<uint32_t> leaf_first; // first triangle of every leaf, in node order
  uint32_t n_interior = 0;
  for (uint32_t i = 0; i < N; ++i) {
    int32_t l = word(i, 0), r = word(i, 1), ts = word(i, 2);
    if (ts > -1) {
      if ((uint32_t)ts > T) { fspt_set_error("node %u: triStart %d > n_tris %u", i, ts, T); return FSPT_E_INVALID; }
      ref[i] = ~(int32_t)leaf_first.size(); // leaf record index
      leaf_first.push_back((uint32_t)ts);
    } else {
      // serializeTree is pre-order (bvh.js:33-50): children come after their parent.
      if (l <= (int32_t)i || r <= (int32_t)i || (uint32_t)l >= N || (uint32_t)r >= N) {
        fspt_set_error("node %u: child indices (%d,%d) violate pre-order / range [%u,%u)", i, l, r, i + 1, N);
        return FSPT_E_INVALID;
      }
      ref[i] = INT32_MAX; // interior, numbered below
      n_interior++;
    }
  }
  {
    const uint32_t TOP_BFS = 256;
    uint32_t next = 0;
    std::vector<uint32_t> queue;
    if (N && word(0, 2) <= -1) queue.push_back(0);
    for (size_t q = 0; q < queue.size() && next < TOP_BFS; ++q) {
      uint32_t i = queue[q];
      ref[i] = (int32_t)next++;
      uint32_t l = (uint32_t)word(i, 0), r = (uint32_t)word(i, 1);
      if (word(l, 2) <= -1) queue.push_back(l);
      if (word(r, 2) <= -1) queue.push_back(r);
    }
#if FSPT_NODE_TREELET > 1
    // Below the breadth-first top: TREELETS.  A treelet = a subtree root and its descendants in breadth-first order, up to
    // FSPT_NODE_TREELET nodes, stored contiguously; the treelets hanging off it follow, depth-first.  A ray that enters
    // a treelet finds the next few levels of its descent - and the sibling it will pop later - in the same or the next
    // 128-byte lines, instead of one line per level (pre-order keeps only the LEFT child next to its parent).  Only the
    // numbering changes: same nodes, same boxes, same traversal order, bit-identical results.
    {
      std::vector<uint32_t> roots; // subtree roots waiting to be laid out (a stack: depth-first over treelets)
      for (size_t q = queue.size(); q-- > 0;)
        if (ref[queue[q]] == INT32_MAX) roots.push_back(queue[q]); // discovered by the top's BFS but beyond its budget
      std::vector<uint32_t> local;
      while (!roots.empty()) {
        const uint32_t root = roots.back();
        roots.pop_back();
        local.assign(1, root);
        for (size_t q = 0; q < local.size(); ++q) {
          const uint32_t i = local[q];
          ref[i] = (int32_t)next++;
          const uint32_t ch[2] = {(uint32_t)word(i, 0), (uint32_t)word(i, 1)};
          for (uint32_t c : ch)
            if (word(c, 2) <= -1 && local.size() < (size_t)FSPT_NODE_TREELET) local.push_back(c);
        }
        // children of the treelet's nodes that did not fit: roots of the next treelets (right before left on the
        // stack, so the left subtree is laid out first, like pre-order)
        for (size_t q = local.size(); q-- > 0;) {
          const uint32_t i = local[q];
          const uint32_t ch[2] = {(uint32_t)word(i, 1), (uint32_t)word(i, 0)};
          for (uint32_t c : ch)
            if (word(c, 2) <= -1 && ref[c] == INT32_MAX) roots.push_back(c);
        }
      }
    }
#endif
    for (uint32_t i = 0; i < N; ++i)
      if (ref[i] == INT32_MAX) ref[i] = (int32_t)next++; // (pre-order for whatever is left: nothing, with treelets)
  }
  std::vector<float> nodes((size_t)(n_interior ? n_interior : 1) * 16, 0.0f);
  // depth of every node (root 0); a child's depth = parent's + 1
  std::vector<uint32_t> depth(N, 0);
  uint32_t max_depth = 0;
  for (uint32_t i = 0; i < N; ++i) {
    int32_t ts = word(i, 2);
    if (ts > -1) continue;
    int32_t l = word(i, 0), r = word(i, 1);
    depth[l] = depth[i] + 1;
    depth[r] = depth[i] + 1;
    if (depth[i] + 1 > max_depth) max_depth = depth[i] + 1;
    float *n = &nodes[(size_t)ref[i] * 16];
    const float *lb = desc->bvh + (size_t)l * 9 + 3, *rb = desc->bvh + (size_t)r * 9 + 3;
    n[0] = lb[0]; n[1] = lb[1]; n[2] = lb[3]; n[3] = lb[4];   // lmin.xy lmax.xy
    n[4] = rb[0]; n[5] = rb[1]; n[6] = rb[3]; n[7] = rb[4];   // rmin.xy rmax.xy
    n[8] = lb[2]; n[9] = lb[5]; n[10] = rb[2]; n[11] = rb[5]; // lmin.z lmax.z rmin.z rmax.z
    int32_t lr[4] = {ref[l], ref[r], 0, 0};
    std::memcpy(n + 12, lr, 16);
  }
  // ---- two-level nodes (fspt_device.hpp "quad"): the node records of both children side by side, one cache line ----
  // Usable only when every interior node's box IS the union of its children's boxes, bit for bit (true for bvh.js trees:
  // a node's box is built from its own triangles); checked here on the caller's arrays, no quads otherwise.
  bool quad_ok = n_interior > 0;
  std::vector<float> quads;
  if (quad_ok) {
    quads.assign((size_t)n_interior * 32, 0.0f);
    auto bits = [](float x) { uint32_t u; std::memcpy(&u, &x, 4); return u; };
    for (uint32_t i = 0; i < N && quad_ok; ++i) {
      if (word(i, 2) > -1) continue;
      const int32_t ch[2] = {word(i, 0), word(i, 1)};
      float *q = &quads[(size_t)ref[i] * 32];
      int32_t refs[8] = {fspt::REF_SENTINEL, fspt::REF_SENTINEL, ref[ch[0]], ref[ch[1]], fspt::REF_SENTINEL, fspt::REF_SENTINEL, 0, 0};
      for (int k = 0; k < 2 && quad_ok; ++k) {
        const int32_t c = ch[k];
        float *part = q + 16 * k;
        const float *cb = desc->bvh + (size_t)c * 9 + 3; // the child's own box: min.xyz max.xyz
        if (word(c, 2) > -1) { // a leaf: its own box, twice
          part[0] = part[4] = cb[0]; part[1] = part[5] = cb[1]; part[2] = part[6] = cb[3]; part[3] = part[7] = cb[4];
          part[8] = part[10] = cb[2]; part[9] = part[11] = cb[5];
        } else {
          const float *cn = &nodes[(size_t)ref[c] * 16];
          std::memcpy(part, cn, 48);
          std::memcpy(&refs[4 * k], cn + 12, 8);
        }
        // box(c) == union of the two boxes of its part, exactly?  (the comparison the device's v_min / v_max make; a pair
        // of candidates that compare equal must be the same bits: -0 / +0)
        const float lo[3][2] = {{part[0], part[4]}, {part[1], part[5]}, {part[8], part[10]}};
        const float hi[3][2] = {{part[2], part[6]}, {part[3], part[7]}, {part[9], part[11]}};
        for (int a = 0; a < 3 && quad_ok; ++a) {
          const float mn = lo[a][0] < lo[a][1] ? lo[a][0] : lo[a][1], mx = hi[a][0] > hi[a][1] ? hi[a][0] : hi[a][1];
          if (std::isnan(lo[a][0]) || std::isnan(lo[a][1]) || std::isnan(hi[a][0]) || std::isnan(hi[a][1])) quad_ok = false;
          if (lo[a][0] == lo[a][1] && bits(lo[a][0]) != bits(lo[a][1])) quad_ok = false;
          if (hi[a][0] == hi[a][1] && bits(hi[a][0]) != bits(hi[a][1])) quad_ok = false;
          if (bits(mn) != bits(cb[a]) || bits(mx) != bits(cb[3 + a])) quad_ok = false;
        }
      }
      std::memcpy(q + 12, &refs[0], 16);
      std::memcpy(q + 28, &refs[4], 16);
    }
    if (!quad_ok) quads.clear();
  }
  if (max_depth + 1 > 64 || max_depth + 1 > fspt::wf_max_stack_entries()) {
    // the reference's stack is int[64] (tracer.fs:368); here one entry per level, in LDS (all 64 fit: 128 KB of the CU's
    // 160 KB under the 512-thread primary launch)
    fspt_set_error("BVH depth %u exceeds the traversal stack (64)", max_depth);
    return FSPT_E_INVALID;
  }
  // ---- pre-edged triangles, padded by leaf_size "-1" triangles (main.js:150-152) ----
  const uint32_t TP = T + desc->leaf_size;
  std::vector<float> tris((size_t)TP * 9, 0.0f);
  for (uint32_t i = 0; i < TP; ++i) {
    float v[9];
    if (i < T) std::memcpy(v, desc->tri + (size_t)i * 9, 36);
    else for (int k = 0; k < 9; ++k) v[k] = -1.0f;
    float *o = &tris[(size_t)i * 9];
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    o[3] = v[3] - v[0]; o[4] = v[4] - v[1]; o[5] = v[5] - v[2]; // e1 = v2 - v1 (tracer.fs:301)
    o[6] = v[6] - v[0]; o[7] = v[7] - v[1]; o[8] = v[8] - v[2]; // e2 = v3 - v1 (tracer.fs:302)
  }
  // ---- leaf records: the leaf_size triangles processLeaf reads from each leaf's first one, component-major ----
  const uint32_t LS = desc->leaf_size;
  const size_t n_leaves = leaf_first.size();
  std::vector<float> leaves((n_leaves ? n_leaves : 1) * (size_t)LS * 9, 0.0f);
  std::vector<uint32_t> slot_tri((n_leaves ? n_leaves : 1) * (size_t)LS, 0u);
  for (size_t L = 0; L < n_leaves; ++L) {
    float *rec = &leaves[L * LS * 9];
    for (uint32_t k = 0; k < LS; ++k) {
      const uint32_t ti = leaf_first[L] + k; // <= T - 1 + leaf_size: inside the padded array
      for (int c = 0; c < 9; ++c) rec[(size_t)c * LS + k] = tris[(size_t)ti * 9 + c];
      slot_tri[L * LS + k] = ti;
    }
  }
  // ---- material texture sets: the four atlas layers a triangle samples at one uv (tracer.fs:453-456) ----
  // layer = clamp(floor(id + 0.5), 0, layers - 1) as texture(sampler2DArray) selects it; same binary32 arithmetic here
  const uint32_t n_layers = desc->atlas_layers;
  auto layer_of = [&](float id) -> uint32_t {
    const float x = std::floor(id + 0.5f);
    if (!(x >= 0.0f)) return 0u; // negative, NaN (the device's float -> int conversion gives 0 for NaN)
    if (x >= (float)(n_layers - 1u)) return n_layers - 1u;
    return (uint32_t)x;
  };
  std::map<std::array<uint32_t, 4>, uint32_t> set_ids;
  std::vector<std::array<uint32_t, 4>> set_keys;
  std::vector<uint32_t> tri_set(T);
  for (uint32_t i = 0; i < T; ++i) {
    const float *m = desc->mat + (size_t)i * 12;
    const std::array<uint32_t, 4> key = {layer_of(m[0]), layer_of(m[1]), layer_of(m[3]), layer_of(m[2])}; // diffuse, emissive, mr, normal
    auto it = set_ids.find(key);
    if (it == set_ids.end()) {
      it = set_ids.emplace(key, (uint32_t)set_keys.size()).first;
      set_keys.push_back(key);
    }
    tri_set[i] = it->second;
  }
  // ---- 192-byte hit records, one per leaf SLOT (what the traversal reports): slot (L, k) holds triangle leaf_first[L] + k ----
  bool has_dielectric = false;
  const size_t n_slots = (n_leaves ? n_leaves : 1) * (size_t)LS;
  std::vector<float> shade(n_slots * 48, 0.0f);
  for (size_t sl = 0; sl < n_leaves * LS; ++sl) {
    const uint32_t i = slot_tri[sl];
    if (i >= T) continue; // "-1" padding: never hit (det = 0)
    float *o = &shade[sl * 48];
    std::memcpy(o, &tris[(size_t)i * 9], 36);
    std::memcpy(o + 9, desc->norm + (size_t)i * 27, 27 * 4);
    std::memcpy(o + 36, desc->uv + (size_t)i * 6, 6 * 4);
    const float *m = desc->mat + (size_t)i * 12;
    std::memcpy(&o[42], &tri_set[i], 4);                     // material texture set (diffuse, emissive, mr, normal layers)
    o[46] = m[9]; o[47] = m[10];                             // ior, dielectric
  }
  for (uint32_t i = 0; i < T; ++i)
    if (desc->mat[(size_t)i * 12 + 10] >= 0.0f) has_dielectric = true;

  int rc = check_device(device);
  if (rc) return rc;
  fspt_scene *s = new fspt_scene();
  s->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->num_cus = prop.multiProcessorCount;
  auto upload = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    if (bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    return e;
  };
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = upload(&s->nodes, nodes.data(), nodes.size() * 4);
  if (e == hipSuccess && quad_ok) e = upload(&s->quads, quads.data(), quads.size() * 4);
  if (e == hipSuccess) e = upload(&s->tris, leaves.data(), leaves.size() * 4);
  if (e == hipSuccess) e = upload(&s->slot_tri, slot_tri.data(), slot_tri.size() * 4);
  if (e == hipSuccess) e = upload(&s->shade, shade.data(), shade.size() * 4);
  // Atlas.  A layer whose texels are all equal - every flat colour: TexturePacker fills whole layers with them
  // (texture_packer.js:36-42), and a colours-only atlas is 1 x 1 - is never stored: its texel sits in the sets that use
  // it.  A set with two or more image layers gets ONE interleaved image (16-byte texels: diffuse, emissive, mr, normal;
  // 4 x 2-texel tiles = 128 bytes), as long as the interleaving budget lasts; the image layers of the other sets are
  // stored once each as single-layer images in 8 x 4-texel tiles.
  uint32_t n_sets = (uint32_t)set_keys.size();
  if (e == hipSuccess) {
    const uint32_t res = desc->atlas_res;
    const size_t per_layer = (size_t)res * res;
    std::vector<uint8_t> is_const(n_layers, 1);
    std::vector<uint32_t> first(n_layers, 0u);
    for (uint32_t l = 0; l < n_layers; ++l) {
      const uint8_t *src = desc->atlas + (size_t)l * per_layer * 4;
      std::memcpy(&first[l], src, 4);
      for (size_t k = 1; k < per_layer && is_const[l]; ++k) is_const[l] = std::memcmp(src + k * 4, &first[l], 4) == 0;
    }
    const uint32_t qtx = (res + 3u) / 4u, qty = (res + 1u) / 2u;
    const size_t quad_tiles = (size_t)qtx * qty;             // 128-byte tiles per interleaved image
    std::vector<uint32_t> tab((size_t)n_sets * 12, 0u), kind(n_sets, fspt::TEXSET_CONST);
    std::vector<int64_t> layer_base(n_layers, -1);            // single-layer image of a layer, in tiles (-1: not stored)
    std::vector<uint32_t> single;                              // the single-layer images, tiled
    std::vector<uint32_t> tiled;
    uint64_t quad_bytes = 0;
    uint32_t n_quad = 0;
    for (uint32_t si = 0; si < n_sets; ++si) {
      const auto &key = set_keys[si];
      uint32_t n_img = 0, distinct[4];
      for (int k = 0; k < 4; ++k) {
        if (is_const[key[k]]) continue;
        bool seen = false;
        for (uint32_t q = 0; q < n_img; ++q) seen = seen || distinct[q] == key[k];
        if (!seen) distinct[n_img++] = key[k];
      }
      if (n_img >= 2 && quad_bytes + quad_tiles * 128u <= g_texset_budget && (n_quad + 1ull) * quad_tiles < 0xFFFFFFFFull) {
        kind[si] = fspt::TEXSET_QUAD;
        quad_bytes += quad_tiles * 128u;
        n_quad++;
      } else if (n_img >= 1) {
        kind[si] = fspt::TEXSET_SEPARATE;
      }
    }
    // single-layer images: the image layers of SEPARATE sets
    for (uint32_t si = 0; si < n_sets; ++si) {
      if (kind[si] != fspt::TEXSET_SEPARATE) continue;
      for (int k = 0; k < 4; ++k) {
        const uint32_t l = set_keys[si][k];
        if (is_const[l] || layer_base[l] >= 0) continue;
        layer_base[l] = (int64_t)(single.size() / (fspt::TEX_TILE_W * fspt::TEX_TILE_H));
        tile_image(desc->atlas + (size_t)l * per_layer * 4, res, res, tiled);
        single.insert(single.end(), tiled.begin(), tiled.end());
      }
    }
    if (single.size() / (fspt::TEX_TILE_W * fspt::TEX_TILE_H) >= 0xFFFFFFFFull) {
      fspt_set_error("atlas too large: %zu texels of image layers", single.size());
      fspt_scene_destroy(s);
      return FSPT_E_INVALID;
    }
    if (e == hipSuccess) e = upload(&s->atlas, single.data(), single.size() * 4);
    if (e == hipSuccess) e = hipMalloc(&s->atlas4, quad_bytes ? quad_bytes : 16);
    std::vector<uint32_t> quad;
    uint32_t qi = 0;
    for (uint32_t si = 0; si < n_sets && e == hipSuccess; ++si) {
      const auto &key = set_keys[si];
      uint32_t *q = &tab[(size_t)si * 12];
      q[0] = kind[si];
      for (int k = 0; k < 4; ++k) {
        q[4 + k] = first[key[k]];
        q[8 + k] = (kind[si] == fspt::TEXSET_SEPARATE && !is_const[key[k]]) ? (uint32_t)layer_base[key[k]] : fspt::LAYER_CONST;
      }
      if (kind[si] != fspt::TEXSET_QUAD) continue;
      q[1] = (uint32_t)(qi * quad_tiles);
      quad.assign(quad_tiles * 32, 0u);
      for (int k = 0; k < 4; ++k) {
        const uint32_t l = key[k];
        const uint8_t *src = desc->atlas + (size_t)l * per_layer * 4;
        for (uint32_t jy = 0; jy < res; ++jy)
          for (uint32_t ix = 0; ix < res; ++ix) {
            uint32_t v = first[l];
            if (!is_const[l]) std::memcpy(&v, src + ((size_t)jy * res + ix) * 4, 4);
            quad[(((size_t)(jy >> 1) * qtx + (ix >> 2)) * 8 + ((jy & 1u) << 2) + (ix & 3u)) * 4 + k] = v;
          }
      }
      e = hipMemcpy((char *)s->atlas4 + (size_t)qi * quad_tiles * 128u, quad.data(), quad.size() * 4, hipMemcpyHostToDevice);
      qi++;
    }
    if (e == hipSuccess) e = upload(&s->tex_sets, tab.data(), tab.size() * 4);
  }
  if (e == hipSuccess && desc->env) {
    std::vector<uint32_t> tiled;
#if FSPT_ENV_APRON
    // overlapping 8 x 4-texel tiles: tile (a, b) = texels [7a, 7a + 8) x [3b, 3b + 4), REPEAT in s, CLAMP in t (main.js:174-178)
    const uint32_t w = desc->env_w, h = desc->env_h, tx = (w + 6u) / 7u, ty = (h + 2u) / 3u;
    tiled.assign((size_t)tx * ty * 32u, 0u);
    for (uint32_t b = 0; b < ty; ++b)
      for (uint32_t a = 0; a < tx; ++a)
        for (uint32_t lb = 0; lb < 4; ++lb)
          for (uint32_t la = 0; la < 8; ++la) {
            const uint32_t i = (7u * a + la) % w;
            uint32_t j = 3u * b + lb;
            if (j > h - 1u) j = h - 1u;
            std::memcpy(&tiled[((size_t)b * tx + a) * 32u + lb * 8u + la], desc->env + ((size_t)j * w + i) * 4, 4);
          }
#else
    tile_image(desc->env, desc->env_w, desc->env_h, tiled);
#endif
    e = upload(&s->env, tiled.data(), tiled.size() * 4);
  }
  if (e == hipSuccess) e = upload(&s->bins, desc->bins, (size_t)desc->n_bins * 16);
  if (e != hipSuccess) {
    fspt_set_error("scene upload failed: %s", hipGetErrorString(e));
    fspt_scene_destroy(s);
    return FSPT_E_HIP;
  }
  s->d.nodes = (const float4 *)s->nodes;
  s->d.quads = (const float4 *)s->quads; // NULL when the boxes are not unions (see above)
  s->d.leaves = (const float *)s->tris;
  s->d.slot_tri = (const uint32_t *)s->slot_tri;
  s->d.hitrec = (const float4 *)s->shade;
  s->d.atlas = (const uint32_t *)s->atlas;
  s->d.atlas4 = (const uint4 *)s->atlas4;
  s->d.tex_sets = (const uint4 *)s->tex_sets;
  s->d.n_tex_sets = n_sets;
  s->d.env = (const uint32_t *)s->env;
  s->d.bins = (const uint4 *)s->bins;
  s->d.atlas_res = desc->atlas_res;
  s->d.atlas_layers = desc->atlas_layers;
  s->d.env_w = desc->env ? desc->env_w : 0;
  s->d.env_h = desc->env ? desc->env_h : 0;
  s->d.n_bins = desc->n_bins;
  s->d.leaf_size = desc->leaf_size;
  s->d.root_ref = ref[0];
  s->d.stack_n = max_depth + 1;
  s->d.n_top = n_interior < 256u ? n_interior : 256u; // interior nodes numbered breadth-first
  s->depth = max_depth;
  s->n_nodes = N;
  s->n_tris = T;
  s->n_interior = n_interior;
  s->has_dielectric = has_dielectric;
  *out = s;
  return FSPT_OK;
}

int fspt_scene_destroy(fspt_scene *s) {
  if (!s) return FSPT_OK;
  hipSetDevice(s->device);
  hipFree(s->nodes); hipFree(s->quads); hipFree(s->tris); hipFree(s->slot_tri); hipFree(s->shade); hipFree(s->atlas); hipFree(s->atlas4); hipFree(s->tex_sets); hipFree(s->env); hipFree(s->bins);
  delete s;
  return FSPT_OK;
}

int fspt_scene_depth(const fspt_scene *s, uint32_t *depth) {
  if (!s || !depth) { fspt_set_error("fspt_scene_depth: NULL argument"); return FSPT_E_INVALID; }
  *depth = s->depth;
  return FSPT_OK;
}

// ---------------------------------------------------------------------------
// target
// ---------------------------------------------------------------------------
static const uint32_t WORK_RING = 4096;
static const uint32_t WF_ROUNDS_MAX = fspt::MAX_PATH_ITERS + 4;
static const uint32_t EV_PAIRS = 4096;
static const size_t WF_HEADS_BYTES = (size_t)(WF_ROUNDS_MAX + 2) * fspt::WF_HEADS * fspt::WF_HEAD_STRIDE * sizeof(uint32_t);
static const uint64_t WF_SLOT_BUDGET = 448ull << 20; // path slots, 216 B each (up to 101 GB of the 288 GB HBM: a 4K frame x 56 ticks)
static_assert(WF_SLOT_BUDGET < (1ull << 29), "k_wf_trace keeps a path's state index in 29 bits");

int fspt_target_create(fspt_scene *scene, uint32_t W, uint32_t H, fspt_target **out) {
  if (!scene || !out || W == 0 || H == 0) { fspt_set_error("fspt_target_create: bad argument"); return FSPT_E_INVALID; }
  if ((uint64_t)W * H > (1ull << 30)) { fspt_set_error("fspt_target_create: %ux%u too large", W, H); return FSPT_E_INVALID; }
  int rc = check_device(scene->device);
  if (rc) return rc;
  fspt_target *t = new fspt_target();
  t->scene = scene;
  t->W = W; t->H = H;
  t->vw = W; t->vh = H;
  size_t px = (size_t)W * H;
  hipError_t e = hipMalloc((void **)&t->accum_own, px * 16);
  if (e == hipSuccess) e = hipMalloc((void **)&t->ray_pos, px * 16);
  if (e == hipSuccess) e = hipMalloc((void **)&t->ray_dir, px * 16);
  if (e == hipSuccess) e = hipMalloc((void **)&t->work_counters, WORK_RING * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&t->counters, 8 * 8); // 6 work counters (fspt_counters) + [6] = traversal steps the trace kernel served from LDS
  if (e == hipSuccess) e = hipStreamCreate(&t->stream);
  if (e == hipSuccess) e = hipEventCreate(&t->ev0);
  if (e == hipSuccess) e = hipEventCreate(&t->ev1);
  if (e == hipSuccess) e = hipEventCreate(&t->ev_start);
  if (e == hipSuccess) e = hipEventCreate(&t->prim_ev[0]);
  if (e == hipSuccess) e = hipEventCreate(&t->prim_ev[1]);
  {
    fspt_target::WfLane &ln = t->wf;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ln.stream_b, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.resolved, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ev_run, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ev_b_last, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ctl_ready, hipEventDisableTiming);
    for (int k = 0; k < fspt::WF_RING; ++k) {
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ev_logic[k], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ev_b[k], hipEventDisableTiming);
    }
  }
  if (e == hipSuccess) e = hipMemsetAsync(t->accum_own, 0, px * 16, t->stream);
  if (e == hipSuccess) e = hipMemsetAsync(t->counters, 0, 64, t->stream);
  if (e != hipSuccess) {
    fspt_set_error("fspt_target_create: %s", hipGetErrorString(e));
    fspt_target_destroy(t);
    return FSPT_E_HIP;
  }
  t->accum = t->accum_own;
  *out = t;
  return FSPT_OK;
}

int fspt_target_destroy(fspt_target *t) {
  if (!t) return FSPT_OK;
  hipSetDevice(t->scene->device);
  // Recorded ticks are dropped: nothing can observe the library's own accumulator any more, and a caller-owned one
  // (fspt_target_bind_accumulator) may already have been freed by its owner - destroy never writes to it.  A caller
  // that wants the recorded ticks in its buffer calls fspt_sync (or re-binds, which flushes) first.
  t->pending.clear();
  if (t->stream) hipStreamSynchronize(t->stream);
  hipFree(t->accum_own); hipFree(t->ray_pos); hipFree(t->ray_dir); hipFree(t->work_counters); hipFree(t->counters);
  {
    fspt_target::WfLane &ln = t->wf;
    for (void *m : ln.mem) hipFree(m);
    hipFree(ln.counts);
    hipFree(ln.heads);
    if (ln.counts_host) hipHostFree(ln.counts_host);
    if (ln.live_host) hipHostFree(ln.live_host);
    if (ln.counts_ready) hipEventDestroy(ln.counts_ready);
    if (ln.resolved) hipEventDestroy(ln.resolved);
    if (ln.stream_b) hipStreamSynchronize(ln.stream_b);
    hipFree(ln.ctl);
    for (int *b : ln.susp) hipFree(b);
    if (ln.ctl_host) hipHostFree(ln.ctl_host);
    for (hipEvent_t ev : {ln.ev_run, ln.ev_b_last, ln.ctl_ready}) if (ev) hipEventDestroy(ev);
    for (int k = 0; k < fspt::WF_RING; ++k) { if (ln.ev_logic[k]) hipEventDestroy(ln.ev_logic[k]); if (ln.ev_b[k]) hipEventDestroy(ln.ev_b[k]); }
    if (ln.stream_b) hipStreamDestroy(ln.stream_b);
    if (ln.stream) hipStreamDestroy(ln.stream);
  }
  if (t->ev_start) hipEventDestroy(t->ev_start);
  for (hipEvent_t ev : t->prim_ev) if (ev) hipEventDestroy(ev);
  for (hipEvent_t e : t->ev_pool) hipEventDestroy(e);
  if (t->ev0) hipEventDestroy(t->ev0);
  if (t->ev1) hipEventDestroy(t->ev1);
  if (t->stream) hipStreamDestroy(t->stream);
  delete t;
  return FSPT_OK;
}

int fspt_target_set_shard(fspt_target *t, uint32_t shard, uint32_t n_shards, uint32_t tile) {
  if (!t) { fspt_set_error("fspt_target_set_shard: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (n_shards == 0 || shard >= n_shards) { fspt_set_error("shard %u of %u invalid", shard, n_shards); return FSPT_E_INVALID; }
  if (tile == 0 || tile % 8 != 0 || tile > 256) { fspt_set_error("tile %u must be a multiple of 8 in [8,256]", tile); return FSPT_E_INVALID; }
  if (t->shard != shard || t->n_shards != n_shards || t->tile != tile) prim_reset(t);
  t->shard = shard; t->n_shards = n_shards; t->tile = tile;
  return FSPT_OK;
}

int fspt_target_bind_accumulator(fspt_target *t, void *device_ptr) {
  if (!t) { fspt_set_error("fspt_target_bind_accumulator: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  t->accum = device_ptr ? (float4 *)device_ptr : t->accum_own;
  return FSPT_OK;
}

int fspt_target_size(fspt_target *t, uint32_t *W, uint32_t *H) {
  if (!t || !W || !H) { fspt_set_error("fspt_target_size: NULL argument"); return FSPT_E_INVALID; }
  *W = t->W; *H = t->H;
  return FSPT_OK;
}

int fspt_target_accumulator(fspt_target *t, void **device_ptr) {
  if (!t || !device_ptr) { fspt_set_error("fspt_target_accumulator: NULL argument"); return FSPT_E_INVALID; }
  *device_ptr = t->accum;
  return FSPT_OK;
}

int fspt_camera(fspt_target *t, const float P[3], const float I[3], float fov_scale, const float lens[2],
                float rand_base) {
  if (!t || !P || !I || !lens) { fspt_set_error("fspt_camera: NULL argument"); return FSPT_E_INVALID; }
  // drawCamera is recorded, not launched: the ticks that use these rays generate them inside the path kernel (the ray
  // textures are only written when somebody looks at them: fspt_read_rays, fspt_trace_test)
  std::memset(&t->last_cam, 0, sizeof(t->last_cam));
  std::memcpy(t->last_cam.P, P, 12); std::memcpy(t->last_cam.I, I, 12);
  t->last_cam.fov_scale = fov_scale; t->last_cam.lens[0] = lens[0]; t->last_cam.lens[1] = lens[1];
  t->last_rb_cam = rand_base;
  t->cam_recorded = true;
  t->rays_injected = false;
  t->rays_valid = true;
  return FSPT_OK;
}

int fspt_set_rays(fspt_target *t, const float *pos, const float *dir) {
  if (!t || !pos || !dir) { fspt_set_error("fspt_set_rays: NULL argument"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  t->cam_recorded = false;
  t->rays_injected = true;
  size_t bytes = (size_t)t->W * t->H * 16;
  HIP_TRY(hipMemcpyAsync(t->ray_pos, pos, bytes, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipMemcpyAsync(t->ray_dir, dir, bytes, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  t->rays_valid = true;
  return FSPT_OK;
}

int fspt_read_rays(fspt_target *t, float *pos, float *dir) {
  if (!t || !pos || !dir) { fspt_set_error("fspt_read_rays: NULL argument"); return FSPT_E_INVALID; }
  if (!t->rays_valid) { fspt_set_error("fspt_read_rays: no rays generated yet"); return FSPT_E_STATE; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  { int rcm = materialise_rays(t); if (rcm) return rcm; }
  size_t bytes = (size_t)t->W * t->H * 16;
  HIP_TRY(hipMemcpyAsync(pos, t->ray_pos, bytes, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipMemcpyAsync(dir, t->ray_dir, bytes, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  return FSPT_OK;
}

static void fill_trace_params(fspt_target *t, fspt::TraceP &p) {
  p.scene = t->scene->d;
  p.W = t->W; p.H = t->H;
  p.vw = t->vw; p.vh = t->vh;
  p.ray_pos = t->ray_pos; p.ray_dir = t->ray_dir;
  p.accum = t->accum;
  p.counters = t->count ? t->counters : nullptr;
  p.shard = t->shard; p.n_shards = t->n_shards; p.tile = t->tile;
  p.tiles_x = (t->W + t->tile - 1) / t->tile;
  p.tiles_y = (t->H + t->tile - 1) / t->tile;
  uint32_t n_tiles = p.tiles_x * p.tiles_y;
  p.n_owned_tiles = (n_tiles > t->shard) ? (n_tiles - t->shard + t->n_shards - 1) / t->n_shards : 0;
}

// Records of suspended traversals: one per lane of the trace grid a launch over `max_paths` paths gets (a lane parks at
// most one ray per launch; launch_wf: min(ceil(paths / 256), 8 blocks per CU) blocks of 256), two buffers, grown on
// demand.  They are part of the target's path state (fspt_target_path_state_bytes, fspt_target_set_memory_limit): when
// they do not fit what the limit leaves, traversals are simply not suspended (*on = false) - same results, a little slower.
static uint64_t susp_need(const fspt_target *t, uint64_t max_paths, uint32_t *stride_out, size_t *recs_out) {
  const uint32_t stride = ((uint32_t)fspt::WF_SUSP_HEADER + t->scene->d.stack_n + 3u) & ~3u;
  const uint64_t grid_max = (uint64_t)t->scene->num_cus * 8u;
  uint64_t blocks = (max_paths + 255u) / 256u;
  if (blocks > grid_max) blocks = grid_max;
  if (blocks < 1) blocks = 1;
  const size_t recs = (size_t)blocks * 256u;
  if (stride_out) *stride_out = stride;
  if (recs_out) *recs_out = recs;
  return 2ull * recs * stride * sizeof(int);
}
static int susp_ensure(fspt_target *t, fspt_target::WfLane &ln, uint64_t max_paths, bool *on) {
  uint32_t stride; size_t recs;
  const uint64_t need = susp_need(t, max_paths, &stride, &recs);
  *on = true;
  if (ln.susp[0] && ln.susp_stride == stride && ln.susp_recs >= recs) {
    // the path state may have grown since the records were made: the limit covers both
    if (!t->mem_limit || ln.bytes + ln.susp_bytes <= t->mem_limit) return FSPT_OK;
  }
  for (int *&b : ln.susp) { if (b) { HIP_TRY(hipFree(b)); b = nullptr; } }
  ln.susp_bytes = 0; ln.susp_recs = 0;
  if (t->mem_limit && ln.bytes + need > t->mem_limit) { *on = false; return FSPT_OK; }
  for (int *&b : ln.susp) {
    hipError_t e = hipMalloc((void **)&b, recs * stride * sizeof(int));
    if (e == hipErrorOutOfMemory) {
      (void)hipGetLastError();
      for (int *&c : ln.susp) { if (c) { (void)hipFree(c); c = nullptr; } }
      *on = false;
      return FSPT_OK;
    }
    HIP_TRY(e);
  }
  ln.susp_stride = stride;
  ln.susp_recs = recs;
  ln.susp_bytes = need;
  return FSPT_OK;
}

// bytes per path slot of every path-state array (fspt_device.hpp: WfP)
// two state sets of A B C E D P (float4) | hit (float2) | shadow_hit (int) | fin (3 floats)   = 216 bytes per slot
static const size_t WF_ARRAY_BYTES[WF_ARRAYS] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 8, 4, 12};
static size_t wf_slot_bytes() {
  size_t b = 0;
  for (size_t x : WF_ARRAY_BYTES) b += x;
  return b;
}

static void wf_release(fspt_target::WfLane &ln) {
  if (ln.stream) hipStreamSynchronize(ln.stream);
  if (ln.stream_b) hipStreamSynchronize(ln.stream_b);
  for (void *&m : ln.mem) { if (m) { hipFree(m); m = nullptr; } }
  ln.slots = 0;
  ln.st_cap = ln.st_fin = 0;
  ln.bytes = 0;
  ln.zeroed = false;
}

// Path state of one lane for `slots` path slots.  `budget_slots` = what fspt_target_set_memory_limit leaves this lane;
// exceeding it is reported exactly like the device running out of memory (FSPT_E_NOMEM: the caller shrinks the batch).
static int wf_ensure(fspt_target *t, fspt_target::WfLane &ln, uint32_t slots, uint64_t budget_slots) {
  (void)t;
  if (ln.slots >= slots && ln.counts && !ln.st_cap) return FSPT_OK;
  wf_release(ln);
  for (int i = 0; i < WF_ARRAYS; ++i) {
    hipError_t e = slots > budget_slots ? hipErrorOutOfMemory : hipMalloc(&ln.mem[i], (size_t)slots * WF_ARRAY_BYTES[i]);
    if (e == hipErrorOutOfMemory) {
      // not enough free HBM (or over the target's memory limit) for this batch size: give everything back
      (void)hipGetLastError();
      wf_release(ln);
      fspt_set_error("path state for %u slots (%zu bytes) does not fit %s", slots, (size_t)slots * wf_slot_bytes(),
                     slots > budget_slots ? "the target's memory limit" : "the free device memory");
      return FSPT_E_NOMEM;
    }
    HIP_TRY(e);
    HIP_TRY(hipMemsetAsync(ln.mem[i], 0, (size_t)slots * WF_ARRAY_BYTES[i], ln.stream)); // touch every page once, now
  }
  if (!ln.counts) HIP_TRY(hipMalloc((void **)&ln.counts, sizeof(fspt::WfCounts) * (WF_ROUNDS_MAX + 2)));
  if (!ln.heads) HIP_TRY(hipMalloc((void **)&ln.heads, WF_HEADS_BYTES));
  if (!ln.counts_host) HIP_TRY(hipHostMalloc((void **)&ln.counts_host, sizeof(fspt::WfCounts) * (WF_ROUNDS_MAX + 2), hipHostMallocDefault));
  if (!ln.live_host) {
    HIP_TRY(hipHostMalloc((void **)&ln.live_host, sizeof(uint32_t) * (WF_ROUNDS_MAX + 2), hipHostMallocMapped));
    std::memset(ln.live_host, 0, sizeof(uint32_t) * (WF_ROUNDS_MAX + 2));
    if (hipHostGetDevicePointer((void **)&ln.live_dev, ln.live_host, 0) != hipSuccess) { (void)hipGetLastError(); ln.live_dev = nullptr; } // (falls back to the copy command)
  }
  if (!ln.counts_ready) HIP_TRY(hipEventCreateWithFlags(&ln.counts_ready, hipEventDisableTiming));
  HIP_TRY(hipStreamSynchronize(ln.stream));
  ln.slots = slots;
  ln.bytes = (uint64_t)slots * wf_slot_bytes();
  return FSPT_OK;
}

// Ticks per batch for a call of n_ticks (0 = the configured steady state, fspt_target_prepare).
// Path state is sized for the largest call seen so far, not for the configured batch: a host that only ever calls
// fspt_trace (one tick at a time, like main.js:842-843) holds one tick of path state, not 128.
static uint32_t wf_plan(const fspt_target *t, uint64_t work_total, uint32_t n_ticks) {
  uint32_t batch = t->batch_ticks;
  if (n_ticks) {
    uint32_t want = n_ticks > t->ticks_seen ? n_ticks : t->ticks_seen;
    if (batch > want) batch = want;
  }
  uint64_t fit = WF_SLOT_BUDGET / work_total;
  if (fit < 1) fit = 1;
  if (batch > fit) batch = (uint32_t)fit;
  if (batch < 1) batch = 1;
  if (batch > (uint32_t)fspt::WF_MAX_BATCH) batch = fspt::WF_MAX_BATCH;
  return batch;
}

// Plan the batch and make sure its path state is allocated.  When the device is short of memory (or the target's
// memory limit is lower) the batch is halved until it fits (results do not depend on the batch size).
static int wf_plan_and_ensure(fspt_target *t, uint64_t work_total, uint32_t n_ticks, uint32_t &batch) {
  if (n_ticks > t->ticks_seen) t->ticks_seen = n_ticks;
  while (true) {
    batch = wf_plan(t, work_total, n_ticks);
    // the trace kernel carries a path's state index in 29 bits (fspt_kernels.hip k_wf_trace: item kind and the
    // no-bounce-left flag share the word); WF_SLOT_BUDGET keeps every batch below that, a single tick of a frame beyond
    // 2^29 pixels does not fit
    if ((uint64_t)batch * work_total > 0x1FFFFFFFull) { fspt_set_error("frame too large for the wavefront pipeline (more than 2^29 paths per batch)"); return FSPT_E_INVALID; }
    // the limit covers the suspension records too (susp_ensure): the slots get what the records this batch needs leave,
    // unless the records alone would take more than a quarter of the limit - then traversals are simply not suspended
    uint64_t budget = ~0ull;
    if (t->mem_limit) {
      uint64_t lim = t->mem_limit;
      const uint64_t rec = (t->susp_budget != 0 && t->count == 0) ? susp_need(t, (uint64_t)batch * work_total, nullptr, nullptr) : 0;
      if (rec <= lim / 4) lim -= rec;
      budget = lim / wf_slot_bytes();
    }
    int rc = wf_ensure(t, t->wf, (uint32_t)(batch * work_total), budget);
    if (rc != FSPT_E_NOMEM) return rc;
    if (batch <= 1) return rc; // one tick does not fit: give up (message set by wf_ensure)
    t->batch_ticks = batch / 2;
  }
}

// Primary-form tuner (fspt_target::prim_ms): fold a finished measurement in (wait = block until it has finished) ...
static void prim_collect(fspt_target *t, bool wait) {
  if (!t->prim_pending) return;
  if (wait) { if (hipEventSynchronize(t->prim_ev[1]) != hipSuccess) return; }
  else if (hipEventQuery(t->prim_ev[1]) != hipSuccess) return;
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, t->prim_ev[0], t->prim_ev[1]) == hipSuccess && t->prim_pending_samples > 0.0) {
    fspt_target::PrimStat &st = t->prim_ms[t->prim_pending_ticks];
    const double v = (double)ms / t->prim_pending_samples;
    const uint32_t f = t->prim_pending_form;
    if (st.best[f] < 0.0 || v < st.best[f]) st.best[f] = v;
    st.runs[f]++;
  }
  t->prim_pending = false;
}
// the measurements describe one launch geometry (shard, viewport, node form, pipeline): a setter that changes it forgets them
static void prim_reset(fspt_target *t) {
  t->prim_pending = false; // (an event pair in flight is simply never read)
  t->prim_ms.clear();
}
// ... and the form for the next batch of `ticks` ticks.  X = the form the scene's size suggests (per-lane refill pays
// where ray lengths scatter: sub-pixel triangles), Y the other one.  Batch 1 of a size runs X - cold: a size's first batch
// is 4-8 % slower (first use of that much path state, clocks, caches) - batch 2 runs Y.  If X won although it ran cold,
// or lost by more than a cold start explains (25 %: the first 128-tick batch - 57 GB of path state used for the first
// time - has been seen 23 % slow), the matter is settled after those two batches; otherwise X gets a
// warm run (batch 3) and the better best-run wins.  (Forms are measured on whole batches: timed on halves of a batch
// the refill form - 512 samples per block iteration - looked 10-20 % worse than it is, profiles/r04/primary_form_tuner_split.log.)
static uint32_t prim_choose(const fspt_target *t, uint32_t ticks) {
  const uint32_t X = t->scene->n_tris >= (1u << 18) ? 2u : 1u, Y = 3u - X;
  const auto it = t->prim_ms.find(ticks);
  if (it == t->prim_ms.end()) return X;
  const fspt_target::PrimStat &st = it->second;
  if (st.runs[X] == 0) return X;
  if (st.runs[Y] == 0) return Y;
  if (st.runs[X] == 1) { // X has only its cold run
    if (st.best[X] <= st.best[Y]) return X;
    if (st.best[X] > 1.25 * st.best[Y]) return Y;
    return X; // its warm run
  }
  return st.best[X] <= st.best[Y] ? X : Y;
}

static int ev_begin(fspt_target *t, int kind, hipStream_t stream) {
  if (t->ev_used >= EV_PAIRS) { t->ev_overflow = true; return -1; }
  if (t->ev_pool.size() < (size_t)(t->ev_used + 1) * 2) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { t->ev_overflow = true; return -1; }
    t->ev_pool.push_back(a); t->ev_pool.push_back(b);
    t->ev_kind.push_back(kind);
  }
  int i = (int)t->ev_used++;
  t->ev_kind[i] = kind;
  hipEventRecord(t->ev_pool[2 * i], stream);
  return i;
}
static void ev_end(fspt_target *t, int i, hipStream_t stream) { if (i >= 0) hipEventRecord(t->ev_pool[2 * i + 1], stream); }

// Live-path statistics of the most recent finished batch (copied to pinned memory behind the batch, never waited for).
static void wf_collect_counts(fspt_target *t, fspt_target::WfLane &ln) {
  if (!ln.counts_pending || hipEventQuery(ln.counts_ready) != hipSuccess) return;
  ln.counts_pending = false;
  if (!ln.counts_slots) return;
  for (uint32_t r = 0; r < WF_ROUNDS_MAX + 2 && r < 80; ++r)
    t->live_frac[r] = (float)(ln.counts_live ? ln.live_host[r] : ln.counts_host[r].n_ext) / (float)ln.counts_slots;
  t->live_known = true;
}

// The round after which the tail kernel takes over (> last: never).  Adaptive, from the previous batch's live-path
// counts.  Two costs are compared per candidate round r (both grow with the rounds still to go, last - r):
//   staying in the wavefront: every further round is a trace launch + a logic launch at their latency floors;
//   the tail kernel: one chain of dependent extension rays per remaining round, times how often its resident lane
//   pairs (4 blocks/CU x 4 waves x 32 pairs) have to be re-filled to get through the n_r live paths.
// Hand over at the first r with  n_r <= 0.9 * (last - r) * resident pairs.  The constant is fitted to hand-over scans
// on two scenes at 1920x1080 (profiles/r02/probe_tail_round_paired.log, probe_tail_round_c3.log): 70 k triangles:
// 1 tick -> after round 2 (562 K paths, 7 rounds to go: 1.74 ms vs 1.93 after round 3), 20 ticks -> round 5, 128 ticks
// -> never (all equal there); 1 M triangles, 20 ticks: round 6 (round 5, 386 K paths with 4 rounds to go, costs 3 %);
// re-scanned with the tail kernel at 4 waves/SIMD: 0.62 / 0.9 / 1.25 / 1.6 (profiles/r02/ab_tail_handover_coefficient.log).
#ifndef FSPT_TAIL_COEF
#define FSPT_TAIL_COEF 0.9
#endif
static uint32_t wf_tail_round(const fspt_target *t, uint64_t slots, uint32_t last) {
  if (t->tail_round == 0) return last + 1;
  if (t->tail_round > 0) return (uint32_t)t->tail_round;
  if (!t->live_known) return last + 1;
  const double pairs = (double)t->scene->num_cus * 4.0 * 4.0 * 32.0;
  for (uint32_t r = 1; r < last && r < 80; ++r)
    if ((double)t->live_frac[r] * (double)slots <= FSPT_TAIL_COEF * (double)(last - r) * pairs) return r;
  return last + 1;
}

// Which node form a launch of kernel class `kind` over (an expected) `paths` paths walks: WfP::wide's bit for it.
static uint32_t wide_bit(const fspt_target *t, int kind, double paths) {
  if (!t->scene->quads || t->count) return 0u;
  const int slot = kind == fspt::WF_K_PRIMARY ? 0 : kind == fspt::WF_K_TRACE ? 1 : kind == fspt::WF_K_TAIL ? 2 : -1;
  if (slot < 0) return 0u;
  bool on;
  if (t->node_form[slot] >= 0) on = t->node_form[slot] != 0;
  else if (slot == 0) on = FSPT_WIDE_PRIMARY != 0;
  else if (slot == 2) on = FSPT_WIDE_TAIL != 0;
  else on = paths >= 0.0 && paths < (double)t->wide_trace_below;
  return on ? 1u << kind : 0u;
}

// n_ticks ticks through the wavefront pipeline.  rays_from_buffers: two-call form (n_ticks == 1).
static int render_wavefront(fspt_target *t, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks,
                            const float *rb_cam, const float *rb_trace, bool rays_from_buffers) {
  fspt::TraceP tp{};
  fill_trace_params(t, tp);
  const uint32_t work_total = tp.n_owned_tiles * tp.tile * tp.tile;
  if (work_total == 0) return FSPT_OK;
  // path state: sized for the largest call so far (fspt_target_prepare sizes it for the configured batch up front, so
  // that a short warm-up call does not cause a reallocation inside a later, longer call)
  uint32_t batch;
  int rc = wf_plan_and_ensure(t, work_total, n_ticks, batch);
  if (rc) return rc;

  fspt::WfP p{};
  p.scene = t->scene->d;
  p.W = t->W; p.H = t->H; p.vw = t->vw; p.vh = t->vh; p.work_total = work_total;
  p.env_theta = cam->env_theta; p.num_bounces = cam->num_bounces;
  std::memcpy(p.cam.P, cam->P, 12); std::memcpy(p.cam.I, cam->I, 12);
  p.cam.fov_scale = cam->fov_scale; p.cam.lens[0] = cam->lens[0]; p.cam.lens[1] = cam->lens[1];
  p.ray_pos = t->ray_pos; p.ray_dir = t->ray_dir;
  p.accum = t->accum;
  p.counters = t->count ? t->counters : nullptr;
  p.shard = tp.shard; p.n_shards = tp.n_shards; p.tile = tp.tile; p.tiles_x = tp.tiles_x; p.tiles_y = tp.tiles_y;
  p.n_owned_tiles = tp.n_owned_tiles;
  const int cus = t->scene->num_cus;
  const bool gen = !rays_from_buffers;
  const uint32_t nb = cam->num_bounces;

  // everything already queued on the target's stream (clear, ray upload, earlier renders) comes first
  HIP_TRY(hipEventRecord(t->ev_start, t->stream));
  fspt_target::WfLane &ln = t->wf;
  hipStream_t st = ln.stream;
  HIP_TRY(hipStreamWaitEvent(st, t->ev_start, 0));

  uint32_t done = 0;
  while (done < n_ticks) {
    auto launch = [&](int kind) -> int {
      int e = kind >= fspt::WF_K_KINDS ? -1 : ev_begin(t, kind, st);
      hipError_t err = fspt::launch_wf(kind, p, t->count, cus, st);
      ev_end(t, e, st);
      if (err != hipSuccess) { fspt_set_error("wavefront launch %d failed: %s", kind, hipGetErrorString(err)); return FSPT_E_HIP; }
      return FSPT_OK;
    };
    for (int k = 0; k < 2; ++k) {
      fspt::WfSet &ws = p.set[k];
      ws.A = (float4 *)ln.mem[6 * k + 0]; ws.B = (float4 *)ln.mem[6 * k + 1]; ws.C = (float4 *)ln.mem[6 * k + 2];
      ws.E = (float4 *)ln.mem[6 * k + 3]; ws.D = (float4 *)ln.mem[6 * k + 4]; ws.P = (float4 *)ln.mem[6 * k + 5];
    }
    p.hit = (float2 *)ln.mem[12]; p.shadow_hit = (int *)ln.mem[13];
    p.fin = (float *)ln.mem[14];
    p.counts = ln.counts;
    p.heads = ln.heads;
    uint32_t nbt = n_ticks - done < batch ? n_ticks - done : batch;
    p.n_batch = nbt;
    p.first_tick = first_tick + done;
    p.ctl = nullptr; p.ring_slots = nbt * work_total; p.finish = 0;
    // suspended traversals: off while counting (the tail kernel re-traces a carried path's rays, which would count twice)
    bool susp_on = t->susp_budget != 0 && t->count == 0;
    if (susp_on && (rc = susp_ensure(t, ln, (uint64_t)nbt * work_total, &susp_on))) return rc;
    p.susp[0] = susp_on ? ln.susp[0] : nullptr; p.susp[1] = susp_on ? ln.susp[1] : nullptr; p.susp_stride = ln.susp_stride; p.susp_budget = susp_on ? t->susp_budget : 0u;
    for (uint32_t j = 0; j < nbt; ++j) { p.rb_cam[j] = rb_cam ? rb_cam[done + j] : 0.0f; p.rb_trace[j] = rb_trace[done + j]; }
    // the previous batch's live-path counts, if their copy has landed: where the tail kernel takes over
    wf_collect_counts(t, ln);
    if (!ln.zeroed) {
      HIP_TRY(hipMemsetAsync(ln.counts, 0, sizeof(fspt::WfCounts) * (WF_ROUNDS_MAX + 2), st));
      HIP_TRY(hipMemsetAsync(ln.heads, 0, WF_HEADS_BYTES, st));
    }
    ln.zeroed = false;
    p.gen_rays = gen ? 1u : 0u;
    // Round 1 = the primary launch (ray generation + primary traversal + its shading); round r >= 2: logic consumes the
    // results of trace r-1 and shades bounce r-1.  After round nb+1 every path has finished unless a refraction kept `i`
    // from advancing (tracer.fs:488).  After round `tail` the tail kernel runs whatever is still alive to completion.
    const uint32_t last = nb + 1;
    uint32_t tail = wf_tail_round(t, (uint64_t)nbt * work_total, last);
    if ((t->scene->has_dielectric || susp_on) && tail > last) tail = last; // refraction / a suspended traversal: paths may outlive `last` rounds
    auto set_round = [&](uint32_t r) { p.round = r; p.cnt_in = r - 1; p.cnt_out = r; p.set_in = (r - 1) & 1u; p.set_out = r & 1u; };
    // the primary launch's form: forced, or measured (see fspt_target::prim_ms)
    prim_collect(t, false);
    uint32_t form = 1;
    if (t->primary_form == 1 || t->primary_form == 2) form = (uint32_t)t->primary_form;
    else if (t->count == 0) form = prim_choose(t, nbt); // (the counting variants are not what is timed: form 1 unless forced)
    p.primary_r = form;
    p.wide = wide_bit(t, fspt::WF_K_PRIMARY, -1.0) | wide_bit(t, fspt::WF_K_TAIL, -1.0);
    const bool time_primary = t->count == 0 && !t->prim_pending;
    bool prev_trace_suspends = false; // (no carry launch behind a trace launch that cannot have suspended anything)
    for (uint32_t r = 1; r <= last && r <= tail; ++r) {
      set_round(r);
      // the paths trace(r-1) suspended move on: a few trailing blocks of the logic launch (FSPT_CARRY_BLOCKS 0: a launch of their own)
      p.carry_blocks = 0u;
      if (r > 1 && susp_on && prev_trace_suspends) {
        if (FSPT_CARRY_BLOCKS) p.carry_blocks = FSPT_CARRY_BLOCKS;
        else if ((rc = launch(fspt::WF_K_CARRY))) return rc;
      }
      if (r == 1 && time_primary) HIP_TRY(hipEventRecord(t->prim_ev[0], st));
      if ((rc = launch(r == 1 ? fspt::WF_K_PRIMARY : fspt::WF_K_LOGIC))) return rc;
      if (r == 1 && time_primary) {
        HIP_TRY(hipEventRecord(t->prim_ev[1], st));
        t->prim_pending = true; t->prim_pending_form = form; t->prim_pending_ticks = nbt;
        t->prim_pending_samples = (double)nbt * (double)work_total;
      }
      if (r < last && r < tail) {
        // (the last trace launch in front of the tail kernel parks its long rays like every other: the carry launch
        // moves those paths on and the tail kernel - a bundle of dependent chains with lanes to spare - traces their rays
        // again from the root.  Letting them finish in the trace launch, as rounds 3 and early 4 did, kept the chip
        // waiting for a handful of rays: 1 M-triangle scene trace 0.249 -> 0.215 ms per tick, a single tick of C2 0.27 ->
        // 0.18, profiles/r04/ab_last_trace_suspends.log)
        // a trace launch expected to be small (the previous batch's live-path counts) is a bundle of dependent chains
        p.wide = (p.wide & ~(1u << fspt::WF_K_TRACE)) |
                 wide_bit(t, fspt::WF_K_TRACE, t->live_known && r < 80 ? (double)t->live_frac[r] * (double)nbt * (double)work_total : -1.0);
        if ((rc = launch(fspt::WF_K_TRACE))) return rc;
        prev_trace_suspends = p.susp_budget != 0;
      }
    }
    if (tail <= last && (tail < last || t->scene->has_dielectric || susp_on)) {
      set_round(tail);
      if ((rc = launch(fspt::WF_K_TAIL))) return rc;
    }
    // The batch's live-path counts go to the host (the tail heuristic's statistics, never waited for) and the counters
    // and pool heads are cleared for the next batch.  Rounds 1-4: a copy command in front of the resolve launch and two
    // fill commands behind it (2.4 % + 1.2 % of a 20-tick batch's GPU time, profiles/r04/final_kernel_stats.csv); now
    // block 0 of the resolve launch does all three (pinned host memory is written by the kernel itself).
    const bool resolve_clears = FSPT_RESOLVE_CLEARS && ln.live_dev != nullptr;
    p.live_out = resolve_clears ? ln.live_dev : nullptr;
    p.zero_rounds = resolve_clears ? WF_ROUNDS_MAX + 2 : 0u;
    if (!resolve_clears) HIP_TRY(hipMemcpyAsync(ln.counts_host, ln.counts, sizeof(fspt::WfCounts) * (WF_ROUNDS_MAX + 2), hipMemcpyDeviceToHost, st));
    // the running mean is order-dependent (tracer.fs:517): batches resolve in tick order - they follow each other on `st`
    if ((rc = launch(fspt::WF_K_RESOLVE))) return rc;
    p.zero_rounds = 0u;
    HIP_TRY(hipEventRecord(ln.counts_ready, st));
    ln.counts_pending = true;
    ln.counts_slots = nbt * work_total;
    ln.counts_live = resolve_clears;
    if (!resolve_clears) {
      HIP_TRY(hipMemsetAsync(ln.counts, 0, sizeof(fspt::WfCounts) * (WF_ROUNDS_MAX + 2), st));
      HIP_TRY(hipMemsetAsync(ln.heads, 0, WF_HEADS_BYTES, st));
    }
    ln.zeroed = true;
    done += nbt;
  }
  HIP_TRY(hipEventRecord(ln.resolved, st));
  HIP_TRY(hipStreamWaitEvent(t->stream, ln.resolved, 0));
  return FSPT_OK;
}

// ---------------------------------------------------------------------------
// Stream scheduler (fspt_device.hpp: WfStreamCtl; fspt_target_set_pipeline code 2)
// ---------------------------------------------------------------------------
static const uint32_t ST_DEFAULT_POOL = 16u << 20; // paths per state set (3.4 GB; profiles/r03/sweep_stream_pool.log)
static const bool ST_DEFAULT_OVERLAP = true; // profiles/r03/ab_stream_overlap.log: 8 Mi pool, 20 / 128 steps: 3 733 / 4 095 Msamples/s against 3 702 / 3 975 on one stream
static const size_t ST_CTL_BYTES = sizeof(fspt::WfStreamCtl);
static const size_t ST_COUNTS_BYTES = sizeof(fspt::WfCounts) * (WF_ROUNDS_MAX + 2);

// Path state of one lane for the stream scheduler: two state sets + ray results for `cap` paths, `fin_slots` finished colours.
static int st_ensure(fspt_target *t, fspt_target::WfLane &ln, uint32_t cap, uint32_t fin_slots, uint64_t budget_bytes) {
  (void)t;
  if (ln.st_cap >= cap && ln.st_fin >= fin_slots && ln.counts && ln.ctl) return FSPT_OK; // (a larger pool from an earlier call is kept)
  wf_release(ln);
  const uint64_t need = (uint64_t)cap * (wf_slot_bytes() - 12) + (uint64_t)fin_slots * 12;
  for (int i = 0; i < WF_ARRAYS; ++i) {
    const size_t bytes = i == 14 ? (size_t)fin_slots * 12 : (size_t)cap * WF_ARRAY_BYTES[i];
    hipError_t e = need > budget_bytes ? hipErrorOutOfMemory : hipMalloc(&ln.mem[i], bytes);
    if (e == hipErrorOutOfMemory) {
      (void)hipGetLastError();
      wf_release(ln);
      fspt_set_error("path pool of %u paths + %u finished samples (%llu bytes) does not fit %s", cap, fin_slots, (unsigned long long)need,
                     need > budget_bytes ? "the target's memory limit" : "the free device memory");
      return FSPT_E_NOMEM;
    }
    HIP_TRY(e);
    HIP_TRY(hipMemsetAsync(ln.mem[i], 0, bytes, ln.stream)); // touch every page once, now
  }
  if (!ln.counts) HIP_TRY(hipMalloc((void **)&ln.counts, ST_COUNTS_BYTES));
  if (!ln.heads) HIP_TRY(hipMalloc((void **)&ln.heads, WF_HEADS_BYTES));
  if (!ln.counts_host) HIP_TRY(hipHostMalloc((void **)&ln.counts_host, ST_COUNTS_BYTES, hipHostMallocDefault));
  if (!ln.counts_ready) HIP_TRY(hipEventCreateWithFlags(&ln.counts_ready, hipEventDisableTiming));
  if (!ln.ctl) HIP_TRY(hipMalloc((void **)&ln.ctl, ST_CTL_BYTES));
  if (!ln.ctl_host) HIP_TRY(hipHostMalloc((void **)&ln.ctl_host, ST_CTL_BYTES, hipHostMallocDefault));
  HIP_TRY(hipStreamSynchronize(ln.stream));
  ln.st_cap = cap; ln.st_fin = fin_slots;
  ln.bytes = need;
  return FSPT_OK;
}

// Geometry of a stream run: n_batch ticks of a lane's share of the frame.
struct StPlan {
  uint32_t cap, unit_slots, take_max, horizon, ring_slots, units;
};
static int st_plan(const fspt_target *t, uint32_t units, uint32_t nbt, uint32_t nb, StPlan &pl) {
  const bool overlap = t->stream_overlap < 0 ? ST_DEFAULT_OVERLAP : t->stream_overlap != 0;
  pl.units = units;
  pl.unit_slots = 64u * nbt;
  uint64_t cap = t->pool_paths ? t->pool_paths : ST_DEFAULT_POOL;
  // a pool larger than the run needs is memory for nothing: everything fits when cap = the run's samples
  const uint64_t all = (uint64_t)units * pl.unit_slots;
  if (cap > all) cap = all;
  if (cap < 2ull * pl.unit_slots) cap = 2ull * pl.unit_slots;
  // every path generated in iteration k has ended after logic(k + horizon): the bounce budget, or - when a material can
  // refract, tracer.fs:488 - the cap on loop iterations
  pl.horizon = t->scene->has_dielectric ? (uint32_t)fspt::MAX_PATH_ITERS : (nb ? nb : 0u);
  // a suspended traversal makes its path lag a round, at most WF_LAG_MAX times (fspt_device.hpp)
  if (t->susp_budget != 0 && t->count == 0) pl.horizon += fspt::WF_LAG_MAX;
  if (t->mem_limit) {
    // what the memory limit leaves per lane: 204 bytes per pool path + its share of the ring, 12 * (horizon + 3) / 2
    // (one stream: / 1) bytes, + one unit of rounding
    const uint64_t per_path = (wf_slot_bytes() - 12) + (overlap ? 6ull : 12ull) * (pl.horizon + 3u);
    uint64_t lane_limit = t->mem_limit;
    // ... minus the suspension records of a pool-sized trace grid (they count as path state: susp_ensure), unless they
    // alone would take more than a quarter of the limit - then this target's traversals are not suspended
    if (t->susp_budget != 0 && t->count == 0) {
      const uint64_t rec = susp_need(t, cap, nullptr, nullptr);
      if (rec <= lane_limit / 4) lane_limit -= rec;
    }
    const uint64_t round_up = 12ull * (pl.horizon + 3u) * pl.unit_slots;
    const uint64_t fit = lane_limit > round_up ? (lane_limit - round_up) / per_path : 0;
    if (cap > fit) cap = fit;
    if (cap < 2ull * pl.unit_slots) { fspt_set_error("the target's memory limit leaves no room for a pool of two units (%u paths)", 2u * pl.unit_slots); return FSPT_E_NOMEM; }
  }
  if (cap > 0x1FFFFFFFull) cap = 0x1FFFFFFFull; // k_wf_trace: 29 bits of state index
  pl.cap = (uint32_t)cap;
  // overlapped: plan(i) runs before logic(i) and has to leave room for every live path; one stream: it runs after
  pl.take_max = (uint32_t)(cap / (overlap ? 2 : 1) / pl.unit_slots);
  if (pl.take_max < 1) pl.take_max = 1;
  const uint64_t ring_units = (uint64_t)(pl.horizon + 3u) * pl.take_max;
  const uint64_t ring = (ring_units < units ? ring_units : units) * pl.unit_slots; // never more than the run itself
  if (ring > 0xFFFFFFFFull) { fspt_set_error("frame too large for the stream scheduler (fin ring of %llu samples)", (unsigned long long)ring); return FSPT_E_INVALID; }
  pl.ring_slots = (uint32_t)ring;
  return FSPT_OK;
}

static void st_collect(fspt_target::WfLane &ln) {
  if (!ln.ctl_pending || hipEventQuery(ln.ctl_ready) != hipSuccess) return;
  ln.ctl_pending = false;
  ln.stat_key = ln.ctl_key;
  uint64_t gen = (uint64_t)ln.ctl_host->last_gen_it + 1u;
  // the finishing launch had to generate units itself: the iterations were too few - scale the estimate up
  const uint32_t fin = ln.ctl_host->fin_gen_units, units = ln.ctl_units;
  if (fin && units > fin) gen = (gen * units + (units - fin) - 1) / (units - fin) + 1;
  else if (fin) gen = gen * 2 + 1;
  ln.stat_gen_iters = (uint32_t)(gen > 100000 ? 100000 : gen);
}

// n_ticks ticks through the stream scheduler.  Everything is enqueued without waiting for the device.
static int render_stream(fspt_target *t, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks,
                         const float *rb_cam, const float *rb_trace, bool rays_from_buffers) {
  fspt::TraceP tp{};
  fill_trace_params(t, tp);
  const uint32_t work_total = tp.n_owned_tiles * tp.tile * tp.tile;
  if (work_total == 0) return FSPT_OK;
  const uint32_t units_total = work_total >> 6; // tile is a multiple of 8: whole 64-pixel patches
  const uint32_t nb = cam->num_bounces;
  const int cus = t->scene->num_cus;
  constexpr uint32_t R = fspt::WF_RING;
  const bool overlap = t->stream_overlap < 0 ? ST_DEFAULT_OVERLAP : t->stream_overlap != 0;
  const bool susp_on = t->susp_budget != 0 && t->count == 0;

  fspt::WfP base{};
  base.scene = t->scene->d;
  base.W = t->W; base.H = t->H; base.vw = t->vw; base.vh = t->vh;
  base.env_theta = cam->env_theta; base.num_bounces = nb;
  std::memcpy(base.cam.P, cam->P, 12); std::memcpy(base.cam.I, cam->I, 12);
  base.cam.fov_scale = cam->fov_scale; base.cam.lens[0] = cam->lens[0]; base.cam.lens[1] = cam->lens[1];
  base.ray_pos = t->ray_pos; base.ray_dir = t->ray_dir;
  base.accum = t->accum;
  base.counters = t->count ? t->counters : nullptr;
  base.shard = tp.shard; base.n_shards = tp.n_shards; base.tile = tp.tile; base.tiles_x = tp.tiles_x; base.tiles_y = tp.tiles_y;
  base.n_owned_tiles = tp.n_owned_tiles;
  base.gen_rays = rays_from_buffers ? 0u : 1u;
  base.primary_r = 1u; // (iterations of varying size: the plain form)
  base.wide = wide_bit(t, fspt::WF_K_PRIMARY, -1.0) | wide_bit(t, fspt::WF_K_TAIL, -1.0);

  // everything already queued on the target's stream (clear, ray upload, earlier renders) comes first
  HIP_TRY(hipEventRecord(t->ev_start, t->stream));
  fspt_target::WfLane &ln = t->wf;
  HIP_TRY(hipStreamWaitEvent(ln.stream, t->ev_start, 0));

  int rc = FSPT_OK;
  auto launch = [&](int kind, const fspt::WfP &p, hipStream_t st) -> int {
    int e = kind >= fspt::WF_K_KINDS ? -1 : ev_begin(t, kind, st);
    hipError_t err = fspt::launch_wf(kind, p, t->count, cus, st);
    ev_end(t, e, st);
    if (err != hipSuccess) { fspt_set_error("stream launch %d failed: %s", kind, hipGetErrorString(err)); return FSPT_E_HIP; }
    return FSPT_OK;
  };

  uint32_t done = 0;
  while (done < n_ticks) {
    // a run covers batch_ticks ticks (what fspt_target_prepare sized the pool for), at most WF_MAX_BATCH
    const uint32_t run_max = t->batch_ticks && t->batch_ticks < (uint32_t)fspt::WF_MAX_BATCH ? t->batch_ticks : (uint32_t)fspt::WF_MAX_BATCH;
    uint32_t nbt = n_ticks - done < run_max ? n_ticks - done : run_max;
    // the samples of a run are numbered in 32 bits (kernels: g = first + i): a frame beyond 2^32 / 128 pixels runs fewer ticks at a time
    while (nbt > 1 && (uint64_t)work_total * nbt > 0xFFFFFFFFull) nbt /= 2;
    if ((uint64_t)work_total * nbt > 0xFFFFFFFFull) { fspt_set_error("frame too large for the stream scheduler (more than 2^32 pixels per shard)"); return FSPT_E_INVALID; }
    fspt::WfP p = base;
    StPlan pl;
    uint32_t iters = 0;
    int res_done = -1; // the iteration whose cursor position marks what has been folded into the accumulator
    bool susp_run = susp_on;
    {
      const uint32_t units = units_total;
      if ((rc = st_plan(t, units, nbt, nb, pl))) return rc;
      const uint64_t budget = t->mem_limit ? t->mem_limit : ~0ull;
      if ((rc = st_ensure(t, ln, pl.cap, pl.ring_slots, budget))) return rc;
      for (int k = 0; k < 2; ++k) {
        fspt::WfSet &ws = p.set[k];
        ws.A = (float4 *)ln.mem[6 * k + 0]; ws.B = (float4 *)ln.mem[6 * k + 1]; ws.C = (float4 *)ln.mem[6 * k + 2];
        ws.E = (float4 *)ln.mem[6 * k + 3]; ws.D = (float4 *)ln.mem[6 * k + 4]; ws.P = (float4 *)ln.mem[6 * k + 5];
      }
      p.hit = (float2 *)ln.mem[12]; p.shadow_hit = (int *)ln.mem[13]; p.fin = (float *)ln.mem[14];
      p.counts = ln.counts; p.heads = ln.heads; p.ctl = ln.ctl;
      p.work_total = units * 64u; p.n_batch = nbt; p.first_tick = first_tick + done;
      p.ring_slots = pl.ring_slots; p.cap = pl.cap; p.take_max = pl.take_max;
      p.wide |= wide_bit(t, fspt::WF_K_TRACE, (double)pl.cap); // (every trace launch of a run is about pool-sized)
      if (susp_run && (rc = susp_ensure(t, ln, pl.cap, &susp_run))) return rc;
      p.susp[0] = susp_run ? ln.susp[0] : nullptr; p.susp[1] = susp_run ? ln.susp[1] : nullptr; p.susp_stride = ln.susp_stride; p.susp_budget = susp_run ? t->susp_budget : 0u;
      p.serial = overlap ? 0u : 1u;
      for (uint32_t j = 0; j < nbt; ++j) { p.rb_cam[j] = rb_cam ? rb_cam[done + j] : 0.0f; p.rb_trace[j] = rb_trace[done + j]; }
      // how many iterations hand out all units: what the last such run needed, else from the pool's equilibrium
      // (about 0.45 of the pool is new samples per iteration at 30 % survival per step)
      st_collect(ln);
      const uint64_t key = ((uint64_t)units << 32) ^ ((uint64_t)nbt << 24) ^ ((uint64_t)nb << 16) ^ (uint64_t)pl.cap * 0x9E3779B97F4A7C15ull;
      uint32_t take_eq = (uint32_t)((overlap ? 0.45 : 0.75) * pl.cap / pl.unit_slots);
      if (take_eq > pl.take_max) take_eq = pl.take_max;
      if (take_eq < 1) take_eq = 1;
      uint32_t gen = (units + take_eq - 1) / take_eq + (units > take_eq ? 1u : 0u);
      if (ln.stat_key == key && ln.stat_gen_iters) gen = ln.stat_gen_iters;
      const uint32_t drain = t->stream_drain >= 0 ? (uint32_t)t->stream_drain : (gen > 1 ? 2u : 0u);
      iters = gen + drain;
      if (t->stream_iter_cap && iters > t->stream_iter_cap) iters = t->stream_iter_cap;
      if (iters < 1) iters = 1;
      ln.ctl_key = key;
      ln.ctl_units = units;
      // a fresh run: cursor 0, no history, counters and pool heads zero
      if (!ln.zeroed) {
        HIP_TRY(hipMemsetAsync(ln.ctl, 0, ST_CTL_BYTES, ln.stream));
        HIP_TRY(hipMemsetAsync(ln.counts, 0, ST_COUNTS_BYTES, ln.stream));
        HIP_TRY(hipMemsetAsync(ln.heads, 0, WF_HEADS_BYTES, ln.stream));
      }
      ln.zeroed = false;
      HIP_TRY(hipEventRecord(ln.ev_run, ln.stream));
      HIP_TRY(hipStreamWaitEvent(ln.stream_b, ln.ev_run, 0));
    }
    for (uint32_t it = 0; it < iters; ++it) {
      {
        hipStream_t A = ln.stream, B = overlap ? ln.stream_b : ln.stream;
        p.round = it; p.cnt_in = (it + R - 1) % R; p.cnt_out = it % R; p.set_in = (it + 1) & 1u; p.set_out = it & 1u;
        if (!overlap) {
          // ---- one stream: logic(it) first, so that plan(it) sees what really survived and fills the pool to the brim
          if (it >= 1) {
            p.carry_blocks = susp_run ? FSPT_CARRY_BLOCKS : 0u;
            if (susp_run && !FSPT_CARRY_BLOCKS && (rc = launch(fspt::WF_K_CARRY, p, A))) return rc;
            if ((rc = launch(fspt::WF_K_LOGIC, p, A))) return rc;
            p.carry_blocks = 0u;
          }
          if ((rc = launch(fspt::WF_K_PLAN, p, A))) return rc;
          if ((rc = launch(fspt::WF_K_PRIMARY, p, A))) return rc;
          const int to = (int)it - (int)pl.horizon; // after logic(it) every path generated up to iteration `to` has ended
          if (to >= 0 && to > res_done) {
            p.res_from = res_done; p.res_to = to;
            if ((rc = launch(fspt::WF_K_RESOLVE, p, A))) return rc;
            res_done = to;
          }
          { const uint32_t keep = p.susp_budget;
            if (it + 1 == iters) p.susp_budget = 0;
            if ((rc = launch(fspt::WF_K_TRACE, p, A))) return rc;
            p.susp_budget = keep; }
          continue;
        }
        // ---- B: plan + primary of iteration `it` (beside trace(it - 1)), then the resolve that logic(it - 1) made possible
        if (it >= 2) HIP_TRY(hipStreamWaitEvent(B, ln.ev_logic[(it - 1) % R], 0)); // logic(it-1) read the set primary(it) writes
        if ((rc = launch(fspt::WF_K_PLAN, p, B))) return rc;
        if ((rc = launch(fspt::WF_K_PRIMARY, p, B))) return rc;
        HIP_TRY(hipEventRecord(ln.ev_b[it % R], B));
        const int to = (int)it - 1 - (int)pl.horizon; // after logic(it-1) every path generated up to iteration `to` has ended
        if (to >= 0 && to > res_done) {
          p.res_from = res_done; p.res_to = to;
          if ((rc = launch(fspt::WF_K_RESOLVE, p, B))) return rc;
          res_done = to;
        }
        // ---- A: logic(it) on the results of trace(it - 1), then trace(it) once primary(it) has added its survivors
        if (it >= 1) {
          p.carry_blocks = susp_run ? FSPT_CARRY_BLOCKS : 0u;
          if (susp_run && !FSPT_CARRY_BLOCKS && (rc = launch(fspt::WF_K_CARRY, p, A))) return rc;
          if ((rc = launch(fspt::WF_K_LOGIC, p, A))) return rc;
          p.carry_blocks = 0u;
          HIP_TRY(hipEventRecord(ln.ev_logic[it % R], A));
        }
        HIP_TRY(hipStreamWaitEvent(A, ln.ev_b[it % R], 0));
        { // (the run's last trace launch lets its long rays finish: the tail kernel would trace them again from the start)
          const uint32_t keep = p.susp_budget;
          if (it + 1 == iters) p.susp_budget = 0;
          if ((rc = launch(fspt::WF_K_TRACE, p, A))) return rc;
          p.susp_budget = keep;
        }
      }
    }
    // ---- the end of the run: logic on the last trace's results, then the tail kernel runs whatever is alive to
    // completion and generates whatever the cursor has not handed out; then the rest is folded into the accumulator
    {
      hipStream_t A = ln.stream, B = overlap ? ln.stream_b : ln.stream;
      const uint32_t it = iters;
      p.round = it; p.cnt_in = (it + R - 1) % R; p.cnt_out = it % R; p.set_in = (it + 1) & 1u; p.set_out = it & 1u;
      // (no carry launch: the run's last trace launch does not suspend)
      if ((rc = launch(fspt::WF_K_LOGIC, p, A))) return rc;
      p.finish = 1;
      if ((rc = launch(fspt::WF_K_TAIL, p, A))) return rc;
      HIP_TRY(hipEventRecord(ln.ev_b_last, B));
      HIP_TRY(hipStreamWaitEvent(A, ln.ev_b_last, 0)); // the resolves so far ran on B
      // everything the iterations handed out (up to the last plan's cursor): the units the finishing launch generated
      // itself went straight into the accumulator
      p.res_from = res_done; p.res_to = (int)it - 1;
      if ((rc = launch(fspt::WF_K_RESOLVE, p, A))) return rc;
      HIP_TRY(hipMemcpyAsync(ln.ctl_host, ln.ctl, ST_CTL_BYTES, hipMemcpyDeviceToHost, A));
      HIP_TRY(hipEventRecord(ln.ctl_ready, A));
      ln.ctl_pending = true;
      HIP_TRY(hipEventRecord(ln.resolved, A));
      // cleared for the next run behind this one, not in front of the next one's first kernel
      HIP_TRY(hipMemsetAsync(ln.ctl, 0, ST_CTL_BYTES, A));
      HIP_TRY(hipMemsetAsync(ln.counts, 0, ST_COUNTS_BYTES, A));
      HIP_TRY(hipMemsetAsync(ln.heads, 0, WF_HEADS_BYTES, A));
      ln.zeroed = true;
      HIP_TRY(hipStreamWaitEvent(B, ln.resolved, 0)); // the next run's B work comes after this run
    }
    done += nbt;
  }
  HIP_TRY(hipStreamWaitEvent(t->stream, ln.resolved, 0));
  return FSPT_OK;
}

// Every path ends after MAX_PATH_ITERS loop iterations (the cap on tracer.fs:488's `i--`), and `i` never exceeds the
// iteration count: a larger NUM_BOUNCES cannot change any sample.  Clamping keeps the per-round tables
// (WfCounts[WF_ROUNDS_MAX + 2], the 8-bit bounce field of the path flags) in range for any caller value.
static uint32_t clamp_bounces(uint32_t nb) { return nb > (uint32_t)FSPT_MAX_BOUNCES ? (uint32_t)FSPT_MAX_BOUNCES : nb; }

// n_ticks ticks with ray generation in the path kernels and explicit per-tick randBase values, on either pipeline
static int render_ticks(fspt_target *t, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks,
                        const float *rbc, const float *rbt) {
  t->ev_used = 0; t->ev_overflow = false;
  if (t->pipeline == 1) {
    HIP_TRY(hipEventRecord(t->ev0, t->stream));
    int rc = t->sched == 1 ? render_stream(t, cam, first_tick, n_ticks, rbc, rbt, false)
                           : render_wavefront(t, cam, first_tick, n_ticks, rbc, rbt, false);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(t->ev1, t->stream));
    t->timed = true; t->last_launches = n_ticks;
    return FSPT_OK;
  }
  fspt::TraceP p{};
  fill_trace_params(t, p);
  std::memcpy(p.cam.P, cam->P, 12); std::memcpy(p.cam.I, cam->I, 12);
  p.cam.fov_scale = cam->fov_scale; p.cam.lens[0] = cam->lens[0]; p.cam.lens[1] = cam->lens[1];
  p.env_theta = cam->env_theta; p.num_bounces = cam->num_bounces;
  bool first = true;
  uint32_t done = 0;
  while (done < n_ticks) {
    uint32_t batch = n_ticks - done < WORK_RING ? n_ticks - done : WORK_RING;
    HIP_TRY(hipMemsetAsync(t->work_counters, 0, (size_t)batch * 4, t->stream));
    if (first) { HIP_TRY(hipEventRecord(t->ev0, t->stream)); first = false; }
    for (uint32_t k = 0; k < batch; ++k) {
      p.rand_base_cam = rbc[done + k];
      p.rand_base = rbt[done + k];
      p.tick = first_tick + done + k;
      p.work_counter = t->work_counters + k;
      HIP_TRY(fspt::launch_trace(p, true, t->count != 0, t->scene->num_cus, t->stream));
    }
    done += batch;
  }
  HIP_TRY(hipEventRecord(t->ev1, t->stream));
  t->timed = true; t->last_launches = n_ticks;
  return FSPT_OK;
}

// Execute the recorded two-call ticks: runs of consecutive ticks with the same camera / envTheta / NUM_BOUNCES go
// through the batched path (ray generation in the kernel from the recorded randBase values - the same arithmetic as
// k_camera followed by a trace of the ray buffers, tests/test_parity_gpu.py).  Called by everything that observes or
// changes state the ticks depend on.
static bool same_view(const fspt_camera_params &a, const fspt_camera_params &b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }
static int flush_pending(fspt_target *t) {
  if (t->pending.empty()) return FSPT_OK;
  std::vector<fspt_target::Deferred> q;
  q.swap(t->pending); // (a failing batch drops the rest: the error is reported once)
  HIP_TRY(hipSetDevice(t->scene->device));
  std::vector<float> rbc, rbt;
  size_t i = 0;
  while (i < q.size()) {
    size_t j = i + 1;
    while (j < q.size() && same_view(q[j].cam, q[i].cam) && q[j].tick == q[j - 1].tick + 1) ++j;
    rbc.clear(); rbt.clear();
    for (size_t k = i; k < j; ++k) { rbc.push_back(q[k].rb_cam); rbt.push_back(q[k].rb_trace); }
    int rc = render_ticks(t, &q[i].cam, q[i].tick, (uint32_t)(j - i), rbc.data(), rbt.data());
    if (rc) return rc;
    i = j;
  }
  return FSPT_OK;
}

// the ray buffers as the most recent fspt_camera call left them (drawCamera's two render targets)
static int materialise_rays(fspt_target *t) {
  if (!t->cam_recorded) return FSPT_OK;
  fspt::CameraP c;
  std::memcpy(c.P, t->last_cam.P, 12); std::memcpy(c.I, t->last_cam.I, 12);
  c.fov_scale = t->last_cam.fov_scale; c.lens[0] = t->last_cam.lens[0]; c.lens[1] = t->last_cam.lens[1];
  HIP_TRY(fspt::launch_camera(t->W, t->H, t->vw, t->vh, c, t->last_rb_cam, t->ray_pos, t->ray_dir, t->stream));
  t->cam_recorded = false;
  return FSPT_OK;
}

int fspt_trace(fspt_target *t, uint32_t tick, float rand_base, float env_theta, uint32_t num_bounces) {
  if (!t) { fspt_set_error("fspt_trace: NULL target"); return FSPT_E_INVALID; }
  if (!t->rays_valid) { fspt_set_error("fspt_trace: call fspt_camera or fspt_set_rays first"); return FSPT_E_STATE; }
  HIP_TRY(hipSetDevice(t->scene->device));
  num_bounces = clamp_bounces(num_bounces);
  if (!t->rays_injected) {
    // rays come from fspt_camera: record the tick; it runs with its neighbours in one batch at the next flush point
    fspt_target::Deferred d;
    d.cam = t->last_cam; d.cam.env_theta = env_theta; d.cam.num_bounces = num_bounces;
    d.rb_cam = t->last_rb_cam; d.tick = tick; d.rb_trace = rand_base;
    t->pending.push_back(d);
    if (!t->defer || t->pending.size() >= (size_t)t->batch_ticks) return flush_pending(t);
    return FSPT_OK;
  }
  FLUSH_OR_RETURN(t);
  if (t->pipeline == 1) {
    fspt_camera_params cp{};
    cp.env_theta = env_theta; cp.num_bounces = num_bounces;
    t->ev_used = 0; t->ev_overflow = false;
    HIP_TRY(hipEventRecord(t->ev0, t->stream));
    int rc = t->sched == 1 ? render_stream(t, &cp, tick, 1, nullptr, &rand_base, true)
                           : render_wavefront(t, &cp, tick, 1, nullptr, &rand_base, true);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(t->ev1, t->stream));
    t->timed = true; t->last_launches = 1;
    return FSPT_OK;
  }
  t->ev_used = 0;
  fspt::TraceP p{};
  fill_trace_params(t, p);
  p.tick = tick; p.rand_base = rand_base; p.rand_base_cam = 0.0f; p.env_theta = env_theta; p.num_bounces = num_bounces;
  HIP_TRY(hipMemsetAsync(t->work_counters, 0, 4, t->stream));
  p.work_counter = t->work_counters;
  HIP_TRY(hipEventRecord(t->ev0, t->stream));
  HIP_TRY(fspt::launch_trace(p, false, t->count != 0, t->scene->num_cus, t->stream));
  HIP_TRY(hipEventRecord(t->ev1, t->stream));
  t->timed = true; t->last_launches = 1;
  return FSPT_OK;
}

int fspt_trace_test(fspt_target *t, uint32_t tick) {
  if (!t) { fspt_set_error("fspt_trace_test: NULL target"); return FSPT_E_INVALID; }
  if (!t->rays_valid) { fspt_set_error("fspt_trace_test: call fspt_camera or fspt_set_rays first"); return FSPT_E_STATE; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  int rcm = materialise_rays(t);
  if (rcm) return rcm;
  fspt::TraceP p{};
  fill_trace_params(t, p);
  p.tick = tick;
  t->ev_used = 0;
  HIP_TRY(hipEventRecord(t->ev0, t->stream));
  HIP_TRY(fspt::launch_bvh_test(p, t->stream));
  HIP_TRY(hipEventRecord(t->ev1, t->stream));
  t->timed = true; t->last_launches = 1;
  return FSPT_OK;
}

int fspt_render(fspt_target *t, const fspt_camera_params *cam_in, uint32_t first_tick, uint32_t n_ticks, uint64_t seed) {
  if (!t || !cam_in) { fspt_set_error("fspt_render: NULL argument"); return FSPT_E_INVALID; }
  if (n_ticks == 0) return FSPT_OK;
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  fspt_camera_params cam_c = *cam_in;
  cam_c.num_bounces = clamp_bounces(cam_c.num_bounces);
  std::vector<float> rbc(n_ticks), rbt(n_ticks);
  uint64_t st0 = seed;
  for (uint32_t k = 0; k < n_ticks; ++k) { rbc[k] = fspt_rand_base_next(&st0); rbt[k] = fspt_rand_base_next(&st0); }
  return render_ticks(t, &cam_c, first_tick, n_ticks, rbc.data(), rbt.data());
}

int fspt_target_set_deferred(fspt_target *t, int enable) {
  if (!t) { fspt_set_error("fspt_target_set_deferred: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  t->defer = enable != 0;
  return FSPT_OK;
}

int fspt_clear(fspt_target *t) {
  if (!t) { fspt_set_error("fspt_clear: NULL target"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipMemsetAsync(t->accum, 0, (size_t)t->W * t->H * 16, t->stream));
  return FSPT_OK;
}

int fspt_sync(fspt_target *t) {
  if (!t) { fspt_set_error("fspt_sync: NULL target"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipStreamSynchronize(t->stream));
  return FSPT_OK;
}

int fspt_read_radiance(fspt_target *t, float *out) {
  if (!t || !out) { fspt_set_error("fspt_read_radiance: NULL argument"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipMemcpyAsync(out, t->accum, (size_t)t->W * t->H * 16, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  return FSPT_OK;
}

int fspt_draw(fspt_target *t, float exposure, float saturation, int denoise, float max_sigma, uint8_t *out_rgba8) {
  return fspt_draw_scaled(t, exposure, saturation, denoise, max_sigma, 1.0f, out_rgba8);
}

int fspt_draw_scaled(fspt_target *t, float exposure, float saturation, int denoise, float max_sigma, float scale,
                     uint8_t *out_rgba8) {
  if (!t || !out_rgba8) { fspt_set_error("fspt_draw: NULL argument"); return FSPT_E_INVALID; }
  if (!(scale > 0.0f && scale <= 1.0f)) { fspt_set_error("fspt_draw: scale must be in (0, 1]"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  FLUSH_OR_RETURN(t);
  size_t n = (size_t)t->W * t->H;
  uint32_t *d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, n * 4));
  hipError_t e = fspt::launch_draw(t->accum, t->W, t->H, exposure, saturation, denoise, max_sigma, scale, d, t->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out_rgba8, d, n * 4, hipMemcpyDeviceToHost, t->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d);
  if (e != hipSuccess) { fspt_set_error("fspt_draw: %s", hipGetErrorString(e)); return FSPT_E_HIP; }
  return FSPT_OK;
}

int fspt_last_kernel_ms(fspt_target *t, float *ms, uint32_t *launches) {
  if (!t || !ms) { fspt_set_error("fspt_last_kernel_ms: NULL argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (!t->timed) { fspt_set_error("fspt_last_kernel_ms: nothing traced yet"); return FSPT_E_STATE; }
  HIP_TRY(hipSetDevice(t->scene->device));
  HIP_TRY(hipEventSynchronize(t->ev1));
  HIP_TRY(hipEventElapsedTime(ms, t->ev0, t->ev1));
  if (launches) *launches = t->last_launches;
  return FSPT_OK;
}

int fspt_target_set_viewport(fspt_target *t, uint32_t w, uint32_t h) {
  if (!t) { fspt_set_error("fspt_target_set_viewport: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (w > t->W || h > t->H) { fspt_set_error("fspt_target_set_viewport: %ux%u exceeds the target %ux%u", w, h, t->W, t->H); return FSPT_E_INVALID; }
  const uint32_t nw = w ? w : t->W, nh = h ? h : t->H;
  if (nw != t->vw || nh != t->vh) prim_reset(t);
  t->vw = nw;
  t->vh = nh;
  return FSPT_OK;
}

int fspt_target_set_pipeline(fspt_target *t, int pipeline, uint32_t batch_ticks) {
  if (!t) { fspt_set_error("fspt_target_set_pipeline: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (pipeline < 0 || pipeline > 2) { fspt_set_error("pipeline must be 0 (megakernel), 1 (wavefront, batches) or 2 (wavefront, stream)"); return FSPT_E_INVALID; }
  if (batch_ticks > (uint32_t)fspt::WF_MAX_BATCH) {
    fspt_set_error("batch_ticks must be <= %d", fspt::WF_MAX_BATCH);
    return FSPT_E_INVALID;
  }
  const int sched = pipeline == 2 ? 1 : 0;
  if (pipeline != 0 && sched != t->sched) {
    // the two schedulers size the path state differently: give it back (the next render allocates what it needs)
    HIP_TRY(hipSetDevice(t->scene->device));
    wf_release(t->wf);
  }
  t->pipeline = pipeline == 0 ? 0 : 1;
  if (pipeline != 0) t->sched = sched;
  if (batch_ticks) t->batch_ticks = batch_ticks;
  return FSPT_OK;
}

int fspt_target_set_pool(fspt_target *t, uint32_t paths, int drain_iterations, uint32_t max_iterations, int overlap) {
  if (!t) { fspt_set_error("fspt_target_set_pool: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (drain_iterations < -1 || drain_iterations > FSPT_MAX_BOUNCES + 1) { fspt_set_error("fspt_target_set_pool: drain_iterations must be -1 (default) or 0..%d", FSPT_MAX_BOUNCES + 1); return FSPT_E_INVALID; }
  t->pool_paths = paths;
  t->stream_drain = drain_iterations;
  t->stream_iter_cap = max_iterations;
  t->stream_overlap = overlap < 0 ? -1 : (overlap ? 1 : 0);
  return FSPT_OK;
}

int fspt_target_set_trace_budget(fspt_target *t, uint32_t steps) {
  if (!t) { fspt_set_error("fspt_target_set_trace_budget: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  t->susp_budget = steps;
  return FSPT_OK;
}

int fspt_target_set_primary_form(fspt_target *t, int form) {
  if (!t) { fspt_set_error("fspt_target_set_primary_form: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (form < 0 || form > 2) { fspt_set_error("fspt_target_set_primary_form: form must be 0 (measure and choose), 1 or 2"); return FSPT_E_INVALID; }
  t->primary_form = form;
  return FSPT_OK;
}

int fspt_target_get_primary_form(fspt_target *t, uint32_t batch_ticks, int *form, double ms_per_sample[2]) {
  if (!t || !form) { fspt_set_error("fspt_target_get_primary_form: NULL argument"); return FSPT_E_INVALID; }
  HIP_TRY(hipSetDevice(t->scene->device));
  prim_collect(t, true);
  double m1 = -1.0, m2 = -1.0;
  const auto it = t->prim_ms.find(batch_ticks);
  if (it != t->prim_ms.end()) { m1 = it->second.best[1]; m2 = it->second.best[2]; }
  *form = (t->primary_form == 1 || t->primary_form == 2) ? t->primary_form : (int)prim_choose(t, batch_ticks);
  if (ms_per_sample) { ms_per_sample[0] = m1; ms_per_sample[1] = m2; }
  return FSPT_OK;
}

int fspt_target_set_node_form(fspt_target *t, int primary, int trace, int tail, int64_t trace_below) {
  if (!t) { fspt_set_error("fspt_target_set_node_form: NULL target"); return FSPT_E_INVALID; }
  const int v[3] = {primary, trace, tail};
  for (int x : v) if (x < -1 || x > 1) { fspt_set_error("fspt_target_set_node_form: form %d (want -1, 0 or 1)", x); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  for (int k = 0; k < 3; ++k) t->node_form[k] = v[k];
  if (trace_below >= 0) t->wide_trace_below = trace_below > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)trace_below;
  prim_reset(t); // the primary-form measurements were taken with the other node form
  return FSPT_OK;
}

int fspt_target_set_tail(fspt_target *t, int round) {
  if (!t) { fspt_set_error("fspt_target_set_tail: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  if (round < -1 || round > FSPT_MAX_BOUNCES + 1) { fspt_set_error("fspt_target_set_tail: round must be -1 (adaptive), 0 (never) or 1..%d", FSPT_MAX_BOUNCES + 1); return FSPT_E_INVALID; }
  t->tail_round = round;
  return FSPT_OK;
}

int fspt_target_live_paths(fspt_target *t, double *frac, uint32_t n_rounds) {
  if (!t || !frac) { fspt_set_error("fspt_target_live_paths: NULL argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  if (t->wf.counts_pending) {
    HIP_TRY(hipEventSynchronize(t->wf.counts_ready));
    wf_collect_counts(t, t->wf);
  }
  for (uint32_t r = 0; r < n_rounds; ++r) frac[r] = (t->live_known && r < 80) ? (double)t->live_frac[r] : 0.0;
  return t->live_known ? FSPT_OK : FSPT_E_STATE;
}

int fspt_target_set_memory_limit(fspt_target *t, uint64_t bytes) {
  if (!t) { fspt_set_error("fspt_target_set_memory_limit: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  t->mem_limit = bytes;
  return FSPT_OK;
}

int fspt_target_path_state_bytes(fspt_target *t, uint64_t *bytes, uint32_t *batch_ticks) {
  if (!t || !bytes) { fspt_set_error("fspt_target_path_state_bytes: NULL argument"); return FSPT_E_INVALID; }
  uint64_t b = 0;
  b = t->wf.bytes + t->wf.susp_bytes;
  *bytes = b;
  if (batch_ticks) *batch_ticks = t->batch_ticks;
  return FSPT_OK;
}

int fspt_target_prepare(fspt_target *t) {
  if (!t) { fspt_set_error("fspt_target_prepare: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  if (t->pipeline != 1) return FSPT_OK;
  fspt::TraceP tp{};
  fill_trace_params(t, tp);
  const uint64_t work_total = (uint64_t)tp.n_owned_tiles * tp.tile * tp.tile;
  if (work_total == 0) return FSPT_OK;
  if (t->sched == 1) {
    // the pool of the configured steady state: runs of batch_ticks ticks (at most WF_MAX_BATCH per run)
    const uint32_t units_total = (uint32_t)(work_total >> 6);
    const uint32_t nbt = t->batch_ticks < (uint32_t)fspt::WF_MAX_BATCH ? (t->batch_ticks ? t->batch_ticks : 1u) : (uint32_t)fspt::WF_MAX_BATCH;
    StPlan pl;
    int rc = st_plan(t, units_total, nbt, clamp_bounces(t->last_cam.num_bounces ? t->last_cam.num_bounces : 8u), pl);
    if (rc == FSPT_OK) rc = st_ensure(t, t->wf, pl.cap, pl.ring_slots, t->mem_limit ? t->mem_limit : ~0ull);
    return rc;
  }
  uint32_t batch;
  return wf_plan_and_ensure(t, work_total, 0, batch);
}

int fspt_last_stage_ms(fspt_target *t, float ms[5], uint32_t launches[5]) {
  if (!t || !ms || !launches) { fspt_set_error("fspt_last_stage_ms: NULL argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  HIP_TRY(hipStreamSynchronize(t->stream));
  for (int k = 0; k < fspt::WF_K_KINDS; ++k) { ms[k] = 0.0f; launches[k] = 0; }
  for (uint32_t i = 0; i < t->ev_used; ++i) {
    float e = 0.0f;
    HIP_TRY(hipEventElapsedTime(&e, t->ev_pool[2 * i], t->ev_pool[2 * i + 1]));
    int k = t->ev_kind[i];
    if (k >= 0 && k < fspt::WF_K_KINDS) { ms[k] += e; launches[k]++; }
  }
  if (t->ev_overflow) { fspt_set_error("stage timing: more than %u launches, timing truncated", EV_PAIRS); return FSPT_E_STATE; }
  return FSPT_OK;
}

int fspt_enable_counters(fspt_target *t, int enable) {
  if (!t) { fspt_set_error("fspt_enable_counters: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  t->count = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
  return FSPT_OK;
}

int fspt_counters_reset(fspt_target *t) {
  if (!t) { fspt_set_error("fspt_counters_reset: NULL target"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  HIP_TRY(hipMemsetAsync(t->counters, 0, 64, t->stream));
  return FSPT_OK;
}

int fspt_get_counters(fspt_target *t, fspt_counters *out) {
  if (!t || !out) { fspt_set_error("fspt_get_counters: NULL argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  unsigned long long v[6];
  HIP_TRY(hipMemcpyAsync(v, t->counters, 48, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  out->samples = v[0]; out->rays = v[1]; out->steps = v[2]; out->leaves = v[3]; out->shades = v[4];
  out->env_lookups = v[5];
  return FSPT_OK;
}

int fspt_get_trace_lds_steps(fspt_target *t, uint64_t *steps) {
  if (!t || !steps) { fspt_set_error("fspt_get_trace_lds_steps: NULL argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  unsigned long long v = 0;
  HIP_TRY(hipMemcpyAsync(&v, t->counters + 6, 8, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  *steps = v;
  return FSPT_OK;
}

// ---------------------------------------------------------------------------
// one frame over several devices (include/fspt.h: fspt_multi_*)
// ---------------------------------------------------------------------------
struct fspt_multi {
  uint32_t W = 0, H = 0;
  std::vector<int> devices;
  std::vector<fspt_scene *> scenes;
  std::vector<fspt_target *> targets;
  std::vector<float4 *> packed;   // per target: its own pixels in work-index order (on its device)
  std::vector<float4 *> staging;  // per target: the same, on devices[0]
  std::vector<hipEvent_t> arrived;
  std::vector<int> peer_direct;   // per target: bit 0 = its device can write devices[0]'s memory directly, bit 1 = the reverse
  uint64_t gather_bytes = 0;
};

static void multi_pack_params(fspt_target *t, fspt::TilePackP &q) {
  fspt::TraceP tp{};
  fill_trace_params(t, tp);
  q.W = t->W; q.H = t->H; q.vw = t->W; q.vh = t->H; // the read-out moves whole tiles, whatever the viewport
  q.shard = tp.shard; q.n_shards = tp.n_shards; q.tile = tp.tile; q.tiles_x = tp.tiles_x; q.tiles_y = tp.tiles_y;
  q.n_owned_tiles = tp.n_owned_tiles;
}

int fspt_multi_destroy(fspt_multi *m) {
  if (!m) return FSPT_OK;
  for (size_t i = 0; i < m->targets.size(); ++i) {
    if (m->targets[i]) { hipSetDevice(m->devices[i]); hipStreamSynchronize(m->targets[i]->stream); }
  }
  for (size_t i = 0; i < m->devices.size(); ++i) {
    if (i < m->packed.size() && m->packed[i]) { hipSetDevice(m->devices[i]); hipFree(m->packed[i]); }
    if (i < m->staging.size() && m->staging[i]) { hipSetDevice(m->devices[0]); hipFree(m->staging[i]); }
    if (i < m->arrived.size() && m->arrived[i]) { hipSetDevice(m->devices[i]); hipEventDestroy(m->arrived[i]); }
  }
  for (fspt_target *t : m->targets) fspt_target_destroy(t);
  for (fspt_scene *s : m->scenes) fspt_scene_destroy(s);
  delete m;
  return FSPT_OK;
}

int fspt_multi_create(const fspt_scene_desc *desc, const int *devices, uint32_t n_devices, uint32_t W, uint32_t H, fspt_multi **out) {
  if (!desc || !devices || !out || n_devices == 0 || n_devices > 64) { fspt_set_error("fspt_multi_create: bad argument (1..64 devices)"); return FSPT_E_INVALID; }
  *out = nullptr;
  fspt_multi *m = new fspt_multi();
  m->W = W; m->H = H;
  m->devices.assign(devices, devices + n_devices);
  m->packed.assign(n_devices, nullptr); m->staging.assign(n_devices, nullptr); m->arrived.assign(n_devices, nullptr); m->peer_direct.assign(n_devices, 3);
  int rc = FSPT_OK;
  for (uint32_t i = 0; i < n_devices && rc == FSPT_OK; ++i) {
    // one scene copy per DISTINCT device (a device listed twice shares it)
    fspt_scene *s = nullptr;
    for (uint32_t j = 0; j < i; ++j) if (devices[j] == devices[i]) { s = m->targets[j]->scene; break; }
    if (!s) { rc = fspt_scene_create(desc, devices[i], &s); if (rc == FSPT_OK) m->scenes.push_back(s); }
    fspt_target *t = nullptr;
    if (rc == FSPT_OK) rc = fspt_target_create(s, W, H, &t);
    if (rc == FSPT_OK) { m->targets.push_back(t); rc = fspt_target_set_shard(t, i, n_devices, 32); }
    if (rc == FSPT_OK && i > 0) {
      fspt::TilePackP q{};
      multi_pack_params(t, q);
      const size_t bytes = (size_t)q.n_owned_tiles * q.tile * q.tile * sizeof(float4);
      hipError_t e = hipSetDevice(devices[i]);
      if (e == hipSuccess) e = hipMalloc((void **)&m->packed[i], bytes ? bytes : 16);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&m->arrived[i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipSetDevice(devices[0]);
      if (e == hipSuccess) e = hipMalloc((void **)&m->staging[i], bytes ? bytes : 16);
      if (e == hipSuccess && devices[i] != devices[0]) {
        // The gather copy is issued on the SENDING device's stream and writes devices[0]'s memory (multi_gather), so
        // the mapping that matters is devices[i] -> devices[0]; the reverse one is enabled too (hipMemcpyPeerAsync may
        // pick either end's copy engine).  Without peer access the copy still works, staged through the host.
        int can_out = 0, can_in = 0;
        (void)hipDeviceCanAccessPeer(&can_out, devices[i], devices[0]);
        (void)hipDeviceCanAccessPeer(&can_in, devices[0], devices[i]);
        if (can_out && hipSetDevice(devices[i]) == hipSuccess && hipDeviceEnablePeerAccess(devices[0], 0) != hipSuccess) (void)hipGetLastError(); // already enabled
        if (can_in && hipSetDevice(devices[0]) == hipSuccess && hipDeviceEnablePeerAccess(devices[i], 0) != hipSuccess) (void)hipGetLastError();
        m->peer_direct[i] = (can_out ? 1 : 0) | (can_in ? 2 : 0);
      } else if (e == hipSuccess) {
        m->peer_direct[i] = 3; // the same device
      }
      if (e != hipSuccess) { fspt_set_error("fspt_multi_create: %s", hipGetErrorString(e)); rc = FSPT_E_HIP; }
    }
  }
  if (rc != FSPT_OK) { fspt_multi_destroy(m); return rc; }
  *out = m;
  return FSPT_OK;
}

int fspt_multi_target(fspt_multi *m, uint32_t i, fspt_target **out) {
  if (!m || !out || i >= m->targets.size()) { fspt_set_error("fspt_multi_target: bad argument"); return FSPT_E_INVALID; }
  *out = m->targets[i];
  return FSPT_OK;
}

#define MULTI_EACH(call)                                                     \
  do {                                                                       \
    if (!m) { fspt_set_error("fspt_multi: NULL handle"); return FSPT_E_INVALID; } \
    for (fspt_target *t : m->targets) { int rc_ = (call); if (rc_) return rc_; }  \
    return FSPT_OK;                                                          \
  } while (0)

int fspt_multi_camera(fspt_multi *m, const float P[3], const float I[3], float fov_scale, const float lens[2], float rand_base) {
  MULTI_EACH(fspt_camera(t, P, I, fov_scale, lens, rand_base));
}
int fspt_multi_trace(fspt_multi *m, uint32_t tick, float rand_base, float env_theta, uint32_t num_bounces) {
  MULTI_EACH(fspt_trace(t, tick, rand_base, env_theta, num_bounces));
}
int fspt_multi_render(fspt_multi *m, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks, uint64_t seed) {
  MULTI_EACH(fspt_render(t, cam, first_tick, n_ticks, seed));
}
int fspt_multi_clear(fspt_multi *m) { MULTI_EACH(fspt_clear(t)); }
int fspt_multi_sync(fspt_multi *m) { MULTI_EACH(fspt_sync(t)); }

// every device packs its own tiles and sends them to devices[0] on its own stream; devices[0] scatters them
static int multi_gather(fspt_multi *m) {
  fspt_target *t0 = m->targets[0];
  m->gather_bytes = 0;
  for (fspt_target *t : m->targets) FLUSH_OR_RETURN(t); // recorded two-call ticks of every device run before its tiles are packed
  for (size_t i = 1; i < m->targets.size(); ++i) {
    fspt_target *t = m->targets[i];
    fspt::TilePackP q{};
    multi_pack_params(t, q);
    const size_t bytes = (size_t)q.n_owned_tiles * q.tile * q.tile * sizeof(float4);
    if (!bytes) continue;
    HIP_TRY(hipSetDevice(m->devices[i]));
    q.accum = t->accum; q.packed = m->packed[i];
    HIP_TRY(fspt::launch_tile_pack(q, false, t->stream));
    HIP_TRY(hipMemcpyPeerAsync(m->staging[i], m->devices[0], m->packed[i], m->devices[i], bytes, t->stream));
    HIP_TRY(hipEventRecord(m->arrived[i], t->stream));
    m->gather_bytes += bytes;
  }
  HIP_TRY(hipSetDevice(m->devices[0]));
  for (size_t i = 1; i < m->targets.size(); ++i) {
    fspt::TilePackP q{};
    multi_pack_params(m->targets[i], q);
    if (!q.n_owned_tiles) continue;
    HIP_TRY(hipStreamWaitEvent(t0->stream, m->arrived[i], 0));
    q.accum = t0->accum; q.packed = m->staging[i];
    HIP_TRY(fspt::launch_tile_pack(q, true, t0->stream));
  }
  return FSPT_OK;
}

int fspt_multi_read_radiance(fspt_multi *m, float *out) {
  if (!m || !out) { fspt_set_error("fspt_multi_read_radiance: NULL argument"); return FSPT_E_INVALID; }
  int rc = multi_gather(m);
  if (rc) return rc;
  return fspt_read_radiance(m->targets[0], out);
}

int fspt_multi_draw(fspt_multi *m, float exposure, float saturation, int denoise, float max_sigma, uint8_t *out_rgba8) {
  if (!m || !out_rgba8) { fspt_set_error("fspt_multi_draw: NULL argument"); return FSPT_E_INVALID; }
  int rc = multi_gather(m);
  if (rc) return rc;
  return fspt_draw(m->targets[0], exposure, saturation, denoise, max_sigma, out_rgba8);
}

int fspt_multi_size(fspt_multi *m, uint32_t *W, uint32_t *H) {
  if (!m || !W || !H) { fspt_set_error("fspt_multi_size: NULL argument"); return FSPT_E_INVALID; }
  *W = m->W; *H = m->H;
  return FSPT_OK;
}

int fspt_multi_peer_access(fspt_multi *m, uint32_t i, int *mask) {
  if (!m || !mask || i >= m->targets.size()) { fspt_set_error("fspt_multi_peer_access: bad argument"); return FSPT_E_INVALID; }
  *mask = m->peer_direct[i];
  return FSPT_OK;
}

int fspt_multi_last_gather_bytes(fspt_multi *m, uint64_t *bytes) {
  if (!m || !bytes) { fspt_set_error("fspt_multi_last_gather_bytes: NULL argument"); return FSPT_E_INVALID; }
  *bytes = m->gather_bytes;
  return FSPT_OK;
}

// ---------------------------------------------------------------------------
// stand-alone intersect + math probes
// ---------------------------------------------------------------------------
int fspt_intersect(fspt_scene *s, const float *rays, uint32_t n, float *t_out, int32_t *index_out, uint32_t *steps_out,
                   uint32_t *leaves_out) {
  return fspt_intersect_form(s, 0, rays, n, t_out, index_out, steps_out, leaves_out);
}

int fspt_scene_two_level_nodes(const fspt_scene *s, int *present, uint64_t *bytes) {
  if (!s) { fspt_set_error("fspt_scene_two_level_nodes: NULL argument"); return FSPT_E_INVALID; }
  if (present) *present = s->quads != nullptr;
  if (bytes) *bytes = s->quads ? (uint64_t)s->n_interior * fspt::QUAD_F4 * 16u : 0u;
  return FSPT_OK;
}

int fspt_intersect_form(fspt_scene *s, int two_level, const float *rays, uint32_t n, float *t_out, int32_t *index_out, uint32_t *steps_out,
                        uint32_t *leaves_out) {
  if (!s || (n && (!rays || !t_out || !index_out))) { fspt_set_error("fspt_intersect: NULL argument"); return FSPT_E_INVALID; }
  if (two_level && !s->quads) { fspt_set_error("fspt_intersect_form: the scene has no two-level nodes (its boxes are not the unions of their children's)"); return FSPT_E_INVALID; }
  if (n == 0) return FSPT_OK;
  HIP_TRY(hipSetDevice(s->device));
  float *d_rays = nullptr, *d_t = nullptr;
  int *d_i = nullptr;
  uint32_t *d_s = nullptr, *d_l = nullptr;
  int rc = FSPT_OK;
  hipError_t e = hipMalloc((void **)&d_rays, (size_t)n * 24);
  if (e == hipSuccess) e = hipMalloc((void **)&d_t, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&d_i, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&d_s, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&d_l, (size_t)n * 4);
  if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)n * 24, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    fspt::IntersectP p{};
    p.scene = s->d; p.rays = d_rays; p.n = n; p.t_out = d_t; p.index_out = d_i; p.steps_out = d_s; p.leaves_out = d_l;
    p.wide = two_level ? 1u : 0u;
    e = fspt::launch_intersect(p, nullptr);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(t_out, d_t, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(index_out, d_i, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && steps_out) e = hipMemcpy(steps_out, d_s, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && leaves_out) e = hipMemcpy(leaves_out, d_l, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { fspt_set_error("fspt_intersect: %s", hipGetErrorString(e)); rc = FSPT_E_HIP; }
  hipFree(d_rays); hipFree(d_t); hipFree(d_i); hipFree(d_s); hipFree(d_l);
  return rc;
}

int fspt_math_eval(int device, int op, const float *a, const float *b, uint32_t n, float *out) {
  if (!a || !out) { fspt_set_error("fspt_math_eval: NULL argument"); return FSPT_E_INVALID; }
  int rc = check_device(device);
  if (rc) return rc;
  if (n == 0) return FSPT_OK;
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  hipError_t e = hipMalloc((void **)&da, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&dout, (size_t)n * 4);
  if (e == hipSuccess && b) e = hipMalloc((void **)&db, (size_t)n * 4);
  if (e == hipSuccess) e = hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && b) e = hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = fspt::launch_math(op, da, db, n, dout, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(dout);
  if (e != hipSuccess) { fspt_set_error("fspt_math_eval: %s", hipGetErrorString(e)); return FSPT_E_HIP; }
  return FSPT_OK;
}

} // extern "C"
