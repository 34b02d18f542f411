"""What ONE rank of an N-GPU run does, measured on one GPU (the 8-GPU run itself is the driver's): the rank's share of
the weak-scaled frame (bench.py --scaling weak: the same picture at sqrt(N) x the linear resolution) or of the fixed 3840x2160 frame (--config c4),
32x32 tiles dealt round-robin, 128-tick batches.  Prints the rank's own Msamples/s - the per-GPU rate the N-GPU run can
at best sum up to before its read-out exchange.
    python tools/shard_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fspt_amd
from fspt_amd import scene as S, distributed as D
arrays = S.bunny_scene(76)
K = 128
for label, n, rank, strong in (("N=1", 1, 0, False), ("weak N=2 rank 1", 2, 1, False), ("weak N=4 rank 2", 4, 2, False),
                               ("weak N=8 rank 5", 8, 5, False), ("c4 strong N=8 rank 0", 8, 0, True),
                               ("c4 strong N=8 rank 7", 8, 7, True), ("c4 strong N=4 rank 3", 4, 3, True)):
    W, H = (3840, 2160) if strong else D.weak_frame(n, 1920, 1080)
    pt = fspt_amd.PathTracer(arrays, W, H, num_bounces=8)
    pt.set_camera(**S.BUNNY_CAMERA)
    pt.set_shard(rank, n, D.TILE)
    pt.set_pipeline("wavefront", K)
    pt.prepare()
    pt.seed(1)
    pt.render(K); pt.sync()
    best = None
    for _ in range(3):
        t0 = time.perf_counter(); pt.render(K); pt.sync(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    own = int(D.owner_mask(rank, n, W, H).sum())
    print(f"{label:24s} frame {W}x{H}  own pixels {own:9d}  {K} ticks in {best * 1e3:7.2f} ms  -> {own * K / best / 1e6:7.1f} Msamples/s on this rank", flush=True)
    pt.close()
