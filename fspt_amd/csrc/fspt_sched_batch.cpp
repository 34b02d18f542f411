// fspt_sched_batch.cpp - the batch scheduler of the wavefront pipeline (include/fspt_tuning.h: pipeline 1): every (pixel, tick)
// of up to 128 ticks at once - primary -> [trace <-> logic] x rounds [-> tail] -> resolve - its path state, the
// suspension records, the primary-form tuner and the tail hand-over rule.  Replaces the draw pair of the reference's tick()
// (main.js:842-843) for runs of recorded ticks; the kernels are fspt_kernels.hip's.
#include "fspt_internal.hpp"

void fill_trace_params(fspt_target *t, fspt::TraceP &p) {
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
uint64_t susp_need(const fspt_target *t, uint64_t max_paths, uint32_t *stride_out, size_t *recs_out) {
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
int susp_ensure(fspt_target *t, fspt_target::WfLane &ln, uint64_t max_paths, bool *on) {
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

size_t wf_slot_bytes() {
  size_t b = 0;
  for (size_t x : WF_ARRAY_BYTES) b += x;
  return b;
}

void wf_release(fspt_target::WfLane &ln) {
  if (ln.stream) hipStreamSynchronize(ln.stream);
  if (ln.stream_b) hipStreamSynchronize(ln.stream_b);
  if (FSPT_BATCH_ON_TARGET_STREAM) (void)hipDeviceSynchronize(); // (the batch scheduler's launches are on the target's stream)
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
int wf_plan_and_ensure(fspt_target *t, uint64_t work_total, uint32_t n_ticks, uint32_t &batch) {
  if (n_ticks > t->ticks_seen) t->ticks_seen = n_ticks;
  // A batch that does not fit the device (or the target's memory limit) is halved - but not below FSPT_MIN_BATCH ticks
  // (or what the call wants, if that is less): every batch pays the same dozen launches at their latency floors, and a
  // frame that cannot hold 8 ticks of path state is what the stream scheduler's FIXED pool is for.  FSPT_E_NOMEM then
  // tells the caller (render_ticks, fspt_target_prepare) to take that one; the configured batch size is left alone.
  const uint32_t configured = t->batch_ticks, want0 = wf_plan(t, work_total, n_ticks);
  const uint32_t floor_ticks = want0 < (uint32_t)FSPT_MIN_BATCH ? want0 : (uint32_t)FSPT_MIN_BATCH;
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
    if (batch / 2 < floor_ticks || batch <= 1) { t->batch_ticks = configured; return rc; } // (message set by wf_ensure)
    t->batch_ticks = batch / 2;
  }
}

// Primary-form tuner (fspt_target::prim_ms): fold a finished measurement in (wait = block until it has finished) ...
void prim_collect(fspt_target *t, bool wait) {
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
void prim_reset(fspt_target *t) {
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
uint32_t prim_choose(const fspt_target *t, uint32_t ticks) {
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

int ev_begin(fspt_target *t, int kind, hipStream_t stream) {
  // FSPT_STAGE_EVENTS=0 (environment, read once): no event pair around the launches - fspt_last_stage_ms then reports
  // zeros; a measurement hook for what the events themselves cost (profiles/r05/ab_stream_events.log)
  static const bool events_on = !(getenv("FSPT_STAGE_EVENTS") && getenv("FSPT_STAGE_EVENTS")[0] == '0');
  if (!events_on || !t->stage_events) return -1;
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
void ev_end(fspt_target *t, int i, hipStream_t stream) { if (i >= 0) hipEventRecord(t->ev_pool[2 * i + 1], stream); }

// Live-path statistics of the most recent finished batch (copied to pinned memory behind the batch, never waited for).
void wf_collect_counts(fspt_target *t, fspt_target::WfLane &ln) {
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

// Traversal steps per T phase of the tail kernel (k_wf_tail slices its rays so that finished pairs are shaded while long
// rays go on).  Short slices pay where rays are long and scatter - the 1 M-triangle scene: tail launch 0.112-0.129 ->
// 0.096-0.098 ms per tick at 16 steps (0.090 at 8), whole job +3.9 ... +5 % - and cost a single tick of the 70 k scene a few
// per cent (round 4: +7 %; round 6: inside the launch's +-0.1 ms): chosen by scene size, like the primary launch's first
// form (profiles/r06/scan_constants2_*.log).
uint32_t wf_tail_slice(const fspt_target *t) { return t->scene->n_tris >= (1u << 18) ? FSPT_TAIL_SLICE_LARGE : FSPT_TAIL_SLICE_SMALL; }

// Which node form a launch of kernel class `kind` over (an expected) `paths` paths walks: WfP::wide's bit for it.
uint32_t wide_bit(const fspt_target *t, int kind, double paths) {
  if (!t->scene->quads || t->count) return 0u;
  const int slot = kind == fspt::WF_K_PRIMARY ? 0 : kind == fspt::WF_K_TRACE ? 1 : kind == fspt::WF_K_TAIL ? 2 : -1;
  if (slot < 0) return 0u;
  bool on;
  if (t->node_form[slot] >= 0) on = t->node_form[slot] != 0; // (tail: 1 always two-level, 2 adaptive - WfP::tail_adaptive)
  else if (slot == 0) on = FSPT_WIDE_PRIMARY != 0;
  else if (slot == 2) on = FSPT_WIDE_TAIL != 0;
  else on = paths >= 0.0 && paths < (double)t->wide_trace_below;
  return on ? 1u << kind : 0u;
}

// n_ticks ticks through the wavefront pipeline.  rays_from_buffers: two-call form (n_ticks == 1).
int render_wavefront(fspt_target *t, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks,
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

  // The batch scheduler runs on the target's own stream: everything already queued there (clear, ray upload, earlier
  // renders) comes first by stream order.  (Rounds 1-4 ran it on a stream of its own behind an event of the target's
  // stream and made the target's stream wait for an event behind the last resolve: two hops between hardware queues per
  // call - ~0.12 ms of launch latency in front of every 20-tick region, profiles/r05/launch_list_c2.txt.)
  fspt_target::WfLane &ln = t->wf;
  hipStream_t st = FSPT_BATCH_ON_TARGET_STREAM ? t->stream : ln.stream;
  if (!FSPT_BATCH_ON_TARGET_STREAM) {
    HIP_TRY(hipEventRecord(t->ev_start, t->stream));
    HIP_TRY(hipStreamWaitEvent(st, t->ev_start, 0));
  }

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
    p.tail_adaptive = (t->node_form[2] < 0 ? FSPT_WIDE_TAIL : t->node_form[2]) == 2 ? 1u : 0u;
    p.tail_slice = wf_tail_slice(t);
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
  if (!FSPT_BATCH_ON_TARGET_STREAM) {
    HIP_TRY(hipEventRecord(ln.resolved, st));
    HIP_TRY(hipStreamWaitEvent(t->stream, ln.resolved, 0));
  }
  return FSPT_OK;
}

