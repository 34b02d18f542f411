"""Full-size cross-check on the GPU: the wavefront pipeline (adaptive tail, no tail, tail after round 1, two lanes) and
the megakernel - two independent schedulers of the same arithmetic - give bit-identical 1920x1080 frames.
    python tools/fullsize_crosscheck.py [c2|c3|textured ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fspt_amd
from fspt_amd import scene as S
W, H, TICKS = 1920, 1080, 6
for name in (sys.argv[1:] or ["c2"]):
    arrays = {"c2": lambda: S.bunny_scene(76), "c3": lambda: S.bunny_scene(289),
              "c5": lambda: S.bunny_scene(76, sun_deg=0.5, sun_gain=2000.0),
              "textured": lambda: S.bunny_scene_textured(76)}[name]()
    ref = None
    for label, pipe, batch, tail in (("megakernel", "megakernel", 0, 0), ("wavefront tail -1", "wavefront", 4, -1),
                                     ("wavefront tail 0", "wavefront", 3, 0), ("wavefront tail 1", "wavefront", 6, 1),
                                     ("wavefront2", "wavefront2", 4, -1)):
        pt = fspt_amd.PathTracer(arrays, W, H, num_bounces=8)
        cam = dict(S.BUNNY_CAMERA, aperture=0.1 if name == "c5" else 0.02)
        pt.set_camera(**cam)
        pt.set_pipeline(pipe, batch)
        if pipe != "megakernel":
            pt.set_tail(tail)
        pt.seed(7)
        pt.render(TICKS)
        img = pt.readRadiance()
        pt.close()
        if ref is None:
            ref = img
            print(name, label, "mean radiance", float(img[..., :3].mean()), flush=True)
        else:
            same = np.array_equal(img, ref, equal_nan=True)
            print(name, label, "== megakernel:", same, "" if same else f"{int((img != ref).any(-1).sum())} pixels differ", flush=True)
            assert same
print("crosscheck ok")
