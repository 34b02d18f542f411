#!/usr/bin/env python3
"""the reference's own tracer.fs (unmodified apart from the documented quad-replication work-around,
SURVEY App. B.4) on the SwiftShader software rasteriser, timed in the BUILD container (8 vCPU, no GPU) at BASELINE
config 1 - 256x256, depth 4, 16 spp - on the FINAL synthetic scene bench.py uses (69 316 triangles, 2048x1024
environment, 86 importance bins).  Reads /root/reference at run time; nothing of it is stored.  Prints one JSON line.
    python tools/glsl_baseline.py [--size 256] [--spp 16] [--bounces 4]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools", "glsl_oracle"))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--bounces", type=int, default=4)
args = ap.parse_args()
import glsl_ref as G
import oracle as O
from fspt_amd import scene as S
if not G.available():
    raise SystemExit("needs /root/reference and the kaleido SwiftShader (build container only)")
arrays = S.bunny_scene(n=76)
cam = dict(S.BUNNY_CAMERA)
lens = S.lens_features(cam["focal_depth"], cam["aperture"])
W = H = args.size
g = G.GlslRef()
g.scene(arrays)
g.target(W, H, replicate=True)
g.tracer(num_bounces=args.bounces)
rbs = O.rand_base_stream(1, 2 * args.spp)
t0 = time.perf_counter()
first = None
for k in range(args.spp):
    g.draw_camera(cam["P"], cam["I"], cam["fov_scale"], lens, rbs[2 * k])
    g.draw_tracer(k, rbs[2 * k + 1], cam["env_theta"])
    if k == 0:
        img, mm = g.read_screen(0)  # forces the first tick (shader JIT included) to finish
        first = time.perf_counter() - t0
img, mm = g.read_screen((args.spp - 1) % 2)
total = time.perf_counter() - t0
# the oracle on the same frame, same randBase values, same container (8 vCPU)
acc = np.zeros((H, W, 4), np.float32)
t1 = time.perf_counter()
O.render(arrays, W, H, cam["P"], cam["I"], cam["fov_scale"], lens, cam["env_theta"], args.bounces, 0, args.spp, 1, acc)
t_or = time.perf_counter() - t1
n = W * H * args.spp
print(json.dumps({"config": f"{W}x{H}, depth {args.bounces}, {args.spp} spp, bunny-synthetic {arrays.n_tris} tri, env {arrays.env_w}x{arrays.env_h}, {arrays.bins.size // 4} bins",
                  "renderer": g.renderer, "cores": os.cpu_count(), "samples": n, "seconds_total": round(total, 2),
                  "seconds_first_tick_incl_jit": round(first, 2), "Msamples_per_s_total": round(n / total / 1e6, 4),
                  "Msamples_per_s_after_first_tick": round((n - W * H) / max(total - first, 1e-9) / 1e6, 4),
                  "lane_mask_mismatches": int(mm), "image_mean": [float(x) for x in img[..., :3].mean((0, 1))],
                  "oracle_same_frame": {"seconds": round(t_or, 2), "Msamples_per_s": round(n / t_or / 1e6, 3), "image_mean": [float(x) for x in acc[..., :3].mean((0, 1))]}}))
