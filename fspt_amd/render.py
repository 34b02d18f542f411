"""Render to PNG without a browser: the synthetic BASELINE scene, or a scene JSON / a frame sequence of the
reference's format (scene/<name>.json with props / static_props / animated_props, main.js:869-975).

    python -m fspt_amd.render --out bunny.png --width 960 --height 540 --spp 256 --bounces 8
    python -m fspt_amd.render --scene web/scene/bunny.json --out bunny.png
    python -m fspt_amd.render --scene 'web/scene/anim_{frame}.json' --frames 0:24 --out 'out/{frame}.png'

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
    ap.add_argument("--scene", default=None, help="scene JSON ({frame} is replaced per frame with --frames)")
    ap.add_argument("--assets", default=None, help="web root the JSON's paths are relative to (default: parent of the scene folder)")
    ap.add_argument("--frames", default=None, help="A:B = frames A..B-1 (the reference's ?frame=N loop, main.js:851-866)")
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    if args.scene:
        from . import scene_file as F
        spp = args.spp if "--spp" in " ".join(__import__("sys").argv) else None  # default: the scene's `samples`
        kw = dict(samples=spp, bounces=args.bounces, seed=args.seed, saturation=args.saturation, denoise=args.denoise)
        if args.frames:
            a, b = (int(x) for x in args.frames.split(":"))
            t0 = time.perf_counter()
            out = F.render_sequence(args.scene, range(a, b), args.out, args.width, args.height, args.assets, **kw)
            print(f"{len(out)} frames in {time.perf_counter() - t0:.2f} s:", *out)
        else:
            arrays, settings = F.load_scene_file(args.scene, args.assets)
            settings["exposure"] = settings["exposure"] * args.exposure
            rgba, rad = F.render_frame(arrays, settings, args.width, args.height, **kw)
            if args.hdr:
                np.save(args.hdr, rad)
            from PIL import Image
            Image.fromarray(rgba[:, :, :3]).save(args.out)
            print("wrote", args.out, f"({arrays.n_tris} triangles, {spp or settings['samples']} spp)")
        return
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
