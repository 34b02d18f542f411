#!/usr/bin/env python3
"""What does ONE traversal step cost a lone wave?  (the tail launch's end phase.)

`fspt_intersect_form` (k_intersect: one ray per lane, intersectScene with step / leaf counters) on the bench scene:
  1. step and leaf-visit counts of a few thousand rays (camera rays and rays between random points of the scene's box);
  2. a handful of those rays, chosen across the range of step counts, each traced ALONE (n = 1: one lane of one wave on
     an idle chip) and as 64 copies of itself (one full wave, no divergence), min of --reps wall-clock calls;
  3. a least-squares line  time = a + b * steps + c * leaves  over the chosen rays: b and c are the dependent cost of a
     node step and of a leaf visit; a is the call's fixed cost (copies, launch, sync).
usage (GPU box):  python3 tools/step_latency.py [--mesh-n 76] [--reps 40] [--two-level]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh-n", type=int, default=76)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--picks", type=int, default=32)
    ap.add_argument("--two-level", action="store_true")
    args = ap.parse_args()
    import fspt_amd
    from fspt_amd import scene as S
    arrays = S.bunny_scene(n=args.mesh_n)
    sc = fspt_amd.Scene(arrays, 0)
    rng = np.random.default_rng(5)
    eye = np.array(S.BUNNY_CAMERA["P"], np.float32)
    n = 8192
    # camera-like rays towards the unit box around the origin, and rays between random points of a slightly larger box
    tgt = rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32)
    d0 = tgt - eye
    a = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    b = rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32)
    rays = np.concatenate([np.concatenate([np.broadcast_to(eye, (n, 3)), d0], 1), np.concatenate([a, b - a], 1)], 0)
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    t, idx, steps, leaves = sc.intersect(rays, two_level=args.two_level)
    order = np.argsort(steps)
    picks = order[np.linspace(0, len(order) - 1, args.picks).astype(int)]

    def timed(r):
        best = 1e9
        for _ in range(args.reps):
            t0 = time.perf_counter()
            sc.intersect(r, two_level=args.two_level)
            best = min(best, time.perf_counter() - t0)
        return best * 1e6

    rows = []
    for i in picks:
        one = rays[i:i + 1]
        rows.append((int(steps[i]), int(leaves[i]), timed(one), timed(np.repeat(one, 64, 0))))
    A = np.array([[1.0, s - l, l] for s, l, _, _ in rows])
    out = {"scene_triangles": int(12 * args.mesh_n ** 2), "two_level": bool(args.two_level), "rays": len(rays),
           "steps_mean": float(steps.mean()), "steps_max": int(steps.max()), "leaf_share_of_steps": float(leaves.sum() / steps.sum()),
           "note": "steps counts the reference's loop iterations: interior-node steps + leaf visits"}
    for name, col in (("one_lane", 2), ("one_full_wave", 3)):
        y = np.array([r[col] for r in rows])
        coef, res, _, _ = np.linalg.lstsq(A, y, rcond=None)
        out[name] = {"fixed_us": round(float(coef[0]), 2), "us_per_node_step": round(float(coef[1]), 4), "us_per_leaf_visit": round(float(coef[2]), 4),
                     "rms_us": round(float(np.sqrt(np.mean((A @ coef - y) ** 2))), 2)}
    print(json.dumps(out))
    for r in rows:
        print("steps %4d leaves %3d  one lane %8.1f us   64 copies %8.1f us" % r)


if __name__ == "__main__":
    main()
