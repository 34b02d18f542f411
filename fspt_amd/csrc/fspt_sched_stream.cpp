// fspt_sched_stream.cpp - the stream scheduler of the wavefront pipeline (include/fspt_tuning.h: pipeline 2): the same
// kernels over a FIXED pool of live paths that k_wf_plan keeps full (fspt_device.hpp: WfStreamCtl).
#include "fspt_internal.hpp"

// ---------------------------------------------------------------------------
// Stream scheduler (fspt_device.hpp: WfStreamCtl; fspt_target_set_pipeline code 2)
// ---------------------------------------------------------------------------
static const uint32_t ST_DEFAULT_POOL = 16u << 20; // paths per state set (3.4 GB; profiles/r03/sweep_stream_pool.log)
static const bool ST_DEFAULT_OVERLAP = true; // profiles/r03/ab_stream_overlap.log: 8 Mi pool, 20 / 128 steps: 3 733 / 4 095 Msamples/s against 3 702 / 3 975 on one stream
static const size_t ST_CTL_BYTES = sizeof(fspt::WfStreamCtl);
static const size_t ST_COUNTS_BYTES = sizeof(fspt::WfCounts) * (WF_ROUNDS_MAX + 2);

// Path state of one lane for the stream scheduler: two state sets + ray results for `cap` paths, `fin_slots` finished colours.
int st_ensure(fspt_target *t, fspt_target::WfLane &ln, uint32_t cap, uint32_t fin_slots, uint64_t budget_bytes) {
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

int st_plan(const fspt_target *t, uint32_t units, uint32_t nbt, uint32_t nb, StPlan &pl) {
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
int render_stream(fspt_target *t, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks,
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
  base.tail_adaptive = (t->node_form[2] < 0 ? FSPT_WIDE_TAIL : t->node_form[2]) == 2 ? 1u : 0u;
  base.tail_slice = wf_tail_slice(t);

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
      // a memory limit too tight for a pool of two units of this run's length: shorter runs (a unit is 64 pixels x the run's ticks)
      while ((rc = st_plan(t, units, nbt, nb, pl)) == FSPT_E_NOMEM && nbt > 1) nbt /= 2;
      if (rc) return rc;
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

