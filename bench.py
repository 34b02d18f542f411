#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-trace hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one tick of the reference's render loop (main.js:838-857): one
sample for every pixel of the frame (camera ray -> full path, tracer.fs:436-518)
accumulated into the running-mean radiance buffer.  N=1 workload = BASELINE
configs[1]: synthetic 'bunny' scene (69 316 triangles), 1920x1080, depth 8.
For N>1 (launched by torch.distributed.run, one rank per GPU) the frame grows
with N (weak scaling: 1920*a x 1080*b, a*b = N), is cut into 32x32 tiles dealt
round-robin to the ranks (no data-path collective), and ONE RCCL exchange of the
RGBA32F radiance buffer to rank 0 closes the timed region: a gather of
each rank's own tiles (default) or a sum-reduce of the full frame (--exchange reduce).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(arrays, W, H, cam, lens, bounces, budget_s=15.0):
    """The oracle (kind 'port': plain-C restatement, OpenMP over rows) timed on
    this host's cores on a bounded, uniformly tile-sampled part of the SAME
    frame: shard 0 of S round-robin 32x32-tile shards, S chosen from a probe."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    cores = os.cpu_count() or 1
    acc = np.zeros((H, W, 4), np.float32)
    n_tiles = ((W + 31) // 32) * ((H + 31) // 32)

    def run(n_shards, n_ticks):
        c = O.OCounters()
        acc[:] = 0
        t0 = time.perf_counter()
        O.render(arrays, W, H, cam["P"], cam["I"], cam["fov_scale"], lens, cam["env_theta"], bounces, 0, n_ticks, 1,
                 acc, counters=c, shard=0, n_shards=n_shards, tile=32)
        return time.perf_counter() - t0, c.as_dict()["samples"]

    probe_shards = max(1, n_tiles // 32)
    t, s = run(probe_shards, 1)
    rate = s / max(t, 1e-9)
    want = rate * budget_s
    n_shards = max(1, int(round(W * H / max(want, 1.0))))
    n_ticks = 1 if n_shards > 1 else max(1, min(256, int(want / (W * H))))
    t, s = run(n_shards, n_ticks)
    if t < 0.6 * budget_s and n_shards == 1:
        # the probe (a 1/32 tile sample) under-estimates how well the full frame scales over the cores: re-run with
        # the tick count that fills the budget
        n_ticks = max(n_ticks + 1, min(512, int(n_ticks * 0.8 * budget_s / max(t, 1e-3))))
        t, s = run(n_shards, n_ticks)
    return {"value": round(s / t / 1e6, 5), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"oracle/liboracle.so (C restatement, OpenMP x{cores}), {n_ticks} tick(s) over every "
                      f"{n_shards}-th 32x32 tile of the same {W}x{H} depth-{bounces} frame: {s} samples in {t:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--mesh-n", type=int, default=76, help="cube-sphere resolution: 12*n^2 triangles (289 -> 1M)")
    ap.add_argument("--aperture", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", default="wavefront", choices=["wavefront", "megakernel", "wavefront2"])
    ap.add_argument("--exchange", default="gather", choices=["gather", "reduce"],
                    help="multi-GPU read-out: gather each rank's own tiles to rank 0 (default) or sum-reduce the full frame")
    ap.add_argument("--batch", type=int, default=128, help="ticks per wavefront batch")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_gpus = world

    import numpy as np
    import torch
    import fspt_amd
    from fspt_amd import scene as S

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libfspt has no CPU path)")
    torch.cuda.set_device(local_rank)
    from fspt_amd import distributed as D
    dist = D.init_process_group(backend="nccl", device=torch.device("cuda", local_rank)) if n_gpus > 1 else None

    t0 = time.perf_counter()
    arrays = S.bunny_scene(n=args.mesh_n)
    build_s = time.perf_counter() - t0
    W, H = D.weak_frame(n_gpus, args.width, args.height)
    cam = dict(S.BUNNY_CAMERA)
    if args.aperture is not None:
        cam["aperture"] = args.aperture
    lens = S.lens_features(cam["focal_depth"], cam["aperture"])

    pt = fspt_amd.PathTracer(arrays, W, H, device=local_rank, num_bounces=args.bounces)
    pt.set_camera(**cam)
    pt.set_shard(rank, n_gpus, D.TILE)
    pt.set_pipeline(args.pipeline, args.batch)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    pt.bind_accumulator(accum.data_ptr(), keep=accum)
    pt.seed(1)
    exch = D.TileGather(rank, n_gpus, W, H, accum.device) if (n_gpus > 1 and args.exchange == "gather") else None
    pt.prepare()  # path-state allocation happens here, never inside the timed region (even with --warmup 0)

    def barrier():
        pt.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup (untimed) ----
    if args.warmup > 0:
        pt.render(args.warmup)
    barrier()
    # ---- timed: exactly K steps ----
    t_start = time.perf_counter()
    pt.render(args.steps)
    pt.sync()
    # the one exchange step of the path (RCCL over xGMI): rank 0 ends up with the whole frame
    if exch is not None:
        exch.exchange(accum)
    else:
        D.reduce_radiance(accum, dst=0)
    barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms, launches = pt.last_kernel_ms()
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_samples = float(W) * H * args.steps
    value = total_samples / elapsed / 1e6

    # ---- stage timing of the TIMED region (HIP events recorded on the target's stream around
    #      every kernel launch) + algorithmic work (counting variant, outside the timed region) ----
    if rank == 0:
        stages = pt.last_stage_ms() if args.pipeline.startswith("wavefront") else None
        pt.enable_counters(True)
        L = fspt_amd._lib
        L.check(L.lib().fspt_counters_reset(pt._t))
        pt.render(1)
        cnt = pt.counters()
        pt.enable_counters(False)
        bps = fspt_amd.bytes_per_sample(cnt)
        spt = cnt["samples"]  # samples per tick on this rank
        per_sample = {k: round(v / max(1, spt), 4) for k, v in cnt.items() if k != "samples"}
        if stages is not None:
            # The same counters for the camera rays alone (depth 0: no shading, no secondary rays) split the algorithmic
            # bytes of SURVEY 8d (60 S + 144 L + 280 H + 16 E + 64 per sample, reference layout) over the kernels:
            #   primary launch (ray generation + camera ray traversal + its shading): 60 S0 + 144 L0 + 280 H1 + 16 E0 + 32
            #   k_wf_trace  (rounds >= 1: extension and shadow rays)               : 60 (S - S0) + 144 (L - L0)
            #   k_wf_logic  (rounds >= 2)                                          : 280 (H - H1) + 16 (E - E0)
            #   k_wf_resolve (running mean)                                        : 32
            nb_keep = pt.num_bounces
            pt.num_bounces = 0
            pt.enable_counters(True)
            L.check(L.lib().fspt_counters_reset(pt._t))
            pt.render(1)
            c0 = pt.counters()
            pt.enable_counters(False)
            pt.num_bounces = nb_keep
            h1 = (c0["samples"] - c0["env_lookups"]) if args.bounces >= 1 else 0  # camera rays that hit and get shaded
            alg = {"primary": 60.0 * c0["steps"] + 144.0 * c0["leaves"] + 280.0 * h1 + 16.0 * c0["env_lookups"] + 32.0 * spt,
                   "trace": 60.0 * (cnt["steps"] - c0["steps"]) + 144.0 * (cnt["leaves"] - c0["leaves"]),
                   "logic": 280.0 * (cnt["shades"] - h1) + 16.0 * (cnt["env_lookups"] - c0["env_lookups"]),
                   "resolve": 32.0 * spt}
            names = {"primary": "fspt::k_wf_logic<false, true, true>", "trace": "fspt::k_wf_trace<false>",
                     "logic": "fspt::k_wf_logic<false, false, true>", "resolve": "fspt::k_wf_resolve"}
            kernels = {}
            for k, (ms, n) in stages.items():
                kernels[k] = {"kernel": names[k], "launches": n, "ms_per_step": round(ms / args.steps, 4),
                              "avg_launch_ms": round(ms / max(1, n), 4), "alg_bytes_per_step": round(alg[k]),
                              "achieved_GBps": round(alg[k] * args.steps / (ms / 1e3) / 1e9, 1) if ms > 0 else None}
            # the roofline block is about the kernel class that takes the largest share of the timed region
            dom = max(("primary", "trace", "logic"), key=lambda k: stages[k][0])
            d_ms, d_n = stages[dom]
            dom_bytes = alg[dom] * args.steps
            achieved = dom_bytes / (d_ms / 1e3) / 1e9
            avg_launch_ms = d_ms / max(1, d_n)
            kernel = names[dom]
            extra = {"launches": d_n, "bytes_per_launch": round(dom_bytes / max(1, d_n)), "kernels": kernels,
                     "stage_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in stages.items()},
                     "stage_launches": {k: v[1] for k, v in stages.items()},
                     "pipeline_GBps": round(bps * spt * args.steps / (kernel_ms / 1e3) / 1e9, 2)}
        else:
            dom = None
            avg_launch_ms = kernel_ms / max(1, launches)
            achieved = bps * spt / (avg_launch_ms / 1e3) / 1e9
            kernel = "fspt::k_trace<true,false>"
            extra = {"launches": launches}
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "kernel": kernel,
                    "avg_launch_ms": round(avg_launch_ms, 4), "bytes_per_sample": round(bps, 1),
                    "per_sample": per_sample}
        roofline.update(extra)
        # HBM traffic of the same kernels from the committed rocprofv3 PMC passes (bench.py cannot run the
        # profiler on itself): only reported for the exact workload those passes measured
        tpath = os.path.join(ROOT, "profiles", "r01", "final_hbm_traffic.json")
        key = {76: "c2_70k", 289: "c3_1M"}.get(args.mesh_n)
        if (stages is not None and key and os.path.exists(tpath) and n_gpus == 1 and args.steps == args.batch == 128
                and (args.width, args.height, args.bounces) == (1920, 1080, 8)):
            tall = json.load(open(tpath)).get(key, {})
            for k in kernels:
                tj = tall.get(names[k].replace("fspt::", ""))
                if tj:
                    kernels[k]["hbm_traffic_per_launch"] = round(tj["hbm_bytes_per_launch_corrected"])
            if "hbm_traffic_per_launch" in kernels[dom]:
                roofline["traffic"] = kernels[dom]["hbm_traffic_per_launch"]
                roofline["traffic_source"] = "profiles/r01/final_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, per launch)"
            # what the kernels queue on besides HBM: the CU's vector-memory pipeline (TA address / TD data-return
            # units of the L1), DESIGN.md 7
            lpath = os.path.join(ROOT, "profiles", "r01", "l1_pipe.json")
            if key == "c2_70k" and os.path.exists(lpath):
                lj = json.load(open(lpath))
                for k in kernels:
                    kj = lj["kernels"].get(names[k].replace("fspt::", ""))
                    if kj:
                        kernels[k]["vmem_pipe_busy"] = {"TA": kj["TA_busy"], "TD": kj["TD_busy"]}
                if "vmem_pipe_busy" in kernels[dom]:
                    roofline["vmem_pipe"] = {"TA_busy": kernels[dom]["vmem_pipe_busy"]["TA"], "TD_busy": kernels[dom]["vmem_pipe_busy"]["TD"],
                                             "l1_gather_peak_GBps": lj["l1_gather_GBps"]["divergent_64B_records"],
                                             "source": "profiles/r01/l1_pipe.json (rocprofv3 --pmc TA_TA_BUSY / TD_TD_BUSY; tools/microbench/gather2)"}
        out = {
            "metric": "Msamples/s at 1920x1080 depth 8 (bunny, 70k tri)",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"bunny-synthetic {arrays.n_tris} tri, {W}x{H}, depth {args.bounces}, "
                                   f"1 spp/step, aperture {cam['aperture']}",
                       "scene_bytes": arrays.nbytes(), "bvh_nodes": arrays.n_nodes, "bvh_depth": arrays.depth,
                       "env_bins": int(arrays.bins.size // 4), "sharding": f"32x32 tiles round-robin over {n_gpus}", "exchange": (args.exchange if n_gpus > 1 else "none"), "pipeline": args.pipeline, "batch_ticks": args.batch,
                       "scene_build_s": round(build_s, 2)},
            "roofline": roofline,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(arrays, W, H, cam, lens, args.bounces)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    pt.close()


if __name__ == "__main__":
    main()
