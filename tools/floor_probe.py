"""Per-launch latency floors of the wavefront kernels at small batch sizes (profiling aid).
    python tools/floor_probe.py W H K [K ...]"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import fspt_amd
from fspt_amd import scene as S
arrays = S.bunny_scene(n=76)
W, H = int(sys.argv[1]), int(sys.argv[2])
for K in [int(x) for x in sys.argv[3:]]:
    pt = fspt_amd.PathTracer(arrays, W, H, num_bounces=8)
    pt.set_camera(**S.BUNNY_CAMERA)
    pt.seed(1)
    pt.set_pipeline("wavefront", max(1, min(64, K)))
    pt.prepare()
    pt.render(K); pt.sync()
    t0 = time.perf_counter(); pt.render(K); pt.sync(); dt = time.perf_counter() - t0
    st = pt.last_stage_ms()
    print(W, H, K, "wall ms", round(dt * 1e3, 3), {k: (round(v[0], 3), v[1], round(v[0] / max(1, v[1]), 4)) for k, v in st.items()}, flush=True)
    pt.close()
