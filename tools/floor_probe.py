"""Per-launch latency floors of the wavefront kernels at small batch sizes (profiling aid).
    python tools/floor_probe.py W H [--tail R] [--reps N] K [K ...]
Prints, per K: wall ms of one fspt_render(K) (best of reps) and the per-class stage times."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fspt_amd
from fspt_amd import scene as S
ap = argparse.ArgumentParser()
ap.add_argument("W", type=int); ap.add_argument("H", type=int)
ap.add_argument("--tail", type=int, default=-1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mesh-n", type=int, default=76)
ap.add_argument("--two-call", action="store_true", help="fspt_camera + fspt_trace per tick instead of fspt_render")
ap.add_argument("K", type=int, nargs="+")
a = ap.parse_args()
arrays = S.bunny_scene(n=a.mesh_n)
for K in a.K:
    pt = fspt_amd.PathTracer(arrays, a.W, a.H, num_bounces=8)
    pt.set_camera(**S.BUNNY_CAMERA)
    pt.seed(1)
    pt.set_pipeline("wavefront", max(1, min(128, K)))
    pt.set_tail(a.tail)
    pt.prepare()
    best = None
    for _ in range(a.reps + 1):  # first = warm-up (and the history the adaptive tail uses)
        t0 = time.perf_counter()
        if a.two_call:
            for _k in range(K):
                pt.tick()
        else:
            pt.render(K)
        pt.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None or _ == 1 else min(best, dt)
    st = pt.last_stage_ms()
    print(a.W, a.H, "K", K, "tail", a.tail, "two_call" if a.two_call else "fused", "wall ms", round(best * 1e3, 3), "Ms/s", round(a.W * a.H * K / best / 1e6, 1),
          {k: (round(v[0], 3), v[1]) for k, v in st.items()}, "live", [round(x, 4) for x in pt.live_paths(10)[1:]], flush=True)
    pt.close()
