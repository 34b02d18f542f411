// fspt_internal.hpp - what the translation units of libfspt's host side share: the scene and target objects behind the
// opaque handles of include/fspt.h, the tuning defaults, and the helpers that cross file boundaries.
//   fspt_api.cpp           scene and target objects, every entry point that is not a scheduler (include/fspt.h order)
//   fspt_sched_batch.cpp   the batch scheduler of the wavefront pipeline (render_wavefront) and its path state
//   fspt_sched_stream.cpp  the stream scheduler (render_stream): a fixed pool of live paths
//   fspt_multi.cpp         one frame over several devices (fspt_multi_*), tile pack / unpack, the optional RCCL exchange
//   scene_build.cpp        the native scene builder (OBJ / MTL / SAH BVH; no GPU)
#pragma once
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

void fspt_set_error(const char *fmt, ...);

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
#ifndef FSPT_TAIL_SLICE_LARGE
#define FSPT_TAIL_SLICE_LARGE 16u // scenes of >= 2^18 triangles (fspt_sched_batch.cpp wf_tail_slice)
#endif
#ifndef FSPT_TAIL_SLICE_SMALL
#define FSPT_TAIL_SLICE_SMALL 32u
#endif
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
// measured in the tail kernel), and a two-level fetch issues 8 requests where the walk needs 4 or 8.  The ADAPTIVE tail
// (two-level nodes only once a wave's list is used up - the launch's end phase, a few lanes walking dependent chains on a
// mostly idle chip) loses as well: 0.037-0.039 -> 0.038-0.040 / 0.80-0.82 -> 0.83-0.88 / 0.120-0.128 -> 0.136-0.145
// (ab_tail_adaptive_*.log) - the second node array is touched by that phase alone, so its lines come from the Infinity Cache
// or HBM where the 64-byte nodes, which every trace launch keeps warm, hit the L2: half as many fetches at twice the
// latency.  So: everything off; the form stays selectable (fspt_target_set_node_form) and tested.
#ifndef FSPT_WIDE_PRIMARY
#define FSPT_WIDE_PRIMARY 0
#endif
#ifndef FSPT_WIDE_TAIL
#define FSPT_WIDE_TAIL 0
#endif
#ifndef FSPT_CARRY_BLOCKS
#define FSPT_CARRY_BLOCKS 0u // trailing blocks of a logic launch that do k_wf_carry's work; 0: a launch of its own per round (rounds 3-4).
// Measured (profiles/r05/ab_fixed_costs_*.log): 4 blocks x 512 threads are a straggler - a few thousand records, one
// memory-side atomic each - the logic launch waits for: logic 0.124 -> 0.152 ms per tick on C2, 0.125 -> 0.178 on the 1 M-triangle scene
#endif
#ifndef FSPT_BATCH_ON_TARGET_STREAM
#define FSPT_BATCH_ON_TARGET_STREAM 1 // the batch scheduler's launches go to the target's stream (0: a stream of their own behind events, rounds 1-4)
#endif
#ifndef FSPT_MIN_BATCH
#define FSPT_MIN_BATCH 8 // ticks a batch of the batch scheduler holds at least; a frame whose path state allows fewer runs on the stream scheduler
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
  bool stream_fallback = false; // sched 0, but the path state of FSPT_MIN_BATCH ticks did not fit: calls run on the stream scheduler (cleared by every setter that changes what fits)
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
  bool stage_events = true;   // fspt_target_set_stage_timing: a HIP event pair around every launch (fspt_last_stage_ms)
};

static const uint32_t WORK_RING = 4096;
static const uint32_t WF_ROUNDS_MAX = fspt::MAX_PATH_ITERS + 4;
static const uint32_t EV_PAIRS = 4096;
static const size_t WF_HEADS_BYTES = (size_t)(WF_ROUNDS_MAX + 2) * fspt::WF_HEADS * fspt::WF_HEAD_STRIDE * sizeof(uint32_t);
static const uint64_t WF_SLOT_BUDGET = 448ull << 20; // path slots, 216 B each (up to 101 GB of the 288 GB HBM: a 4K frame x 56 ticks)
static_assert(WF_SLOT_BUDGET < (1ull << 29), "k_wf_trace keeps a path's state index in 29 bits");


// bytes per path slot of every path-state array (fspt_device.hpp: WfP)
// two state sets of A B C E D P (float4) | hit (float2) | shadow_hit (int) | fin (3 floats)   = 216 bytes per slot
static const size_t WF_ARRAY_BYTES[WF_ARRAYS] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 8, 4, 12};

// Geometry of a stream run: n_batch ticks of a lane's share of the frame.
struct StPlan {
  uint32_t cap, unit_slots, take_max, horizon, ring_slots, units;
};

#define FLUSH_OR_RETURN(t) do { int rc_f = flush_pending(t); if (rc_f) return rc_f; } while (0)

// ---- fspt_api.cpp
int check_device(int device);
int flush_pending(fspt_target *t);     // execute the recorded two-call ticks
int materialise_rays(fspt_target *t);  // the ray buffers as the most recent fspt_camera call left them
uint32_t clamp_bounces(uint32_t nb);
// ---- fspt_sched_batch.cpp
void fill_trace_params(fspt_target *t, fspt::TraceP &p);
uint64_t susp_need(const fspt_target *t, uint64_t max_paths, uint32_t *stride_out, size_t *recs_out);
int susp_ensure(fspt_target *t, fspt_target::WfLane &ln, uint64_t max_paths, bool *on);
size_t wf_slot_bytes();
void wf_release(fspt_target::WfLane &ln);
int wf_plan_and_ensure(fspt_target *t, uint64_t work_total, uint32_t n_ticks, uint32_t &batch);
void prim_collect(fspt_target *t, bool wait);
void prim_reset(fspt_target *t);
uint32_t prim_choose(const fspt_target *t, uint32_t ticks);
int ev_begin(fspt_target *t, int kind, hipStream_t stream);
void ev_end(fspt_target *t, int i, hipStream_t stream);
void wf_collect_counts(fspt_target *t, fspt_target::WfLane &ln);
uint32_t wide_bit(const fspt_target *t, int kind, double paths);
uint32_t wf_tail_slice(const fspt_target *t); // fspt_sched_batch.cpp: the tail kernel's slice length for this target's scene
int render_wavefront(fspt_target *t, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks,
                     const float *rb_cam, const float *rb_trace, bool rays_from_buffers);
// ---- fspt_sched_stream.cpp
int st_ensure(fspt_target *t, fspt_target::WfLane &ln, uint32_t cap, uint32_t fin_slots, uint64_t budget_bytes);
int st_plan(const fspt_target *t, uint32_t units, uint32_t nbt, uint32_t nb, StPlan &pl);
int render_stream(fspt_target *t, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks,
                  const float *rb_cam, const float *rb_trace, bool rays_from_buffers);
