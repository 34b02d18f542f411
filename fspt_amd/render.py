"""Render the synthetic BASELINE scene to a PNG (a browser-free way to look at the result).

    python -m fspt_amd.render --out bunny.png --width 960 --height 540 --spp 256 --bounces 8

Path tracing runs in the HIP kernels (fspt_render), tone mapping in the draw.fs kernel (fspt_draw).
"""
import argparse
import time

import numpy as np

from . import PathTracer, scene as S


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--out", default="fspt.png")
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--mesh-n", type=int, default=76)
    ap.add_argument("--exposure", type=float, default=1.0)
    ap.add_argument("--saturation", type=float, default=1.0)
    ap.add_argument("--denoise", action="store_true")
    ap.add_argument("--hdr", default=None, help="also save the RGBA32F radiance buffer as .npy")
    args = ap.parse_args()
    arrays = S.bunny_scene(n=args.mesh_n)
    pt = PathTracer(arrays, args.width, args.height, num_bounces=args.bounces)
    pt.set_camera(**S.BUNNY_CAMERA)
    t0 = time.perf_counter()
    pt.render(args.spp)
    pt.sync()
    dt = time.perf_counter() - t0
    rgba = pt.draw(args.exposure, args.saturation, args.denoise)
    print(f"{args.width}x{args.height} x {args.spp} spp in {dt:.3f} s = {args.width * args.height * args.spp / dt / 1e6:.0f} Msamples/s")
    if args.hdr:
        np.save(args.hdr, pt.readRadiance())
    from PIL import Image
    Image.fromarray(rgba[::-1, :, :3]).save(args.out)  # radiance rows are bottom-up (GL origin)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
