#!/usr/bin/env python3
"""bench.py's synthetic 'bunny' workload as a web root on disk, the form the reference loads scenes from (scene JSON ->
OBJ files -> environment image; main.js:915-950), for the Node host's bench.js:

    python tools/write_bench_scene.py <dir> [--mesh-n 76] [--sun-deg 1.5] [--sun-gain 60]
    node fspt_amd/js/bench.js --scene <dir>/scene/bench.json --focal-depth 2 --aperture 0.02

Same OBJ text, props, environment and camera as fspt_amd.scene.bunny_scene / BUNNY_CAMERA (bench.py --config c2)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fspt_amd import scene as S  # noqa: E402


def write(root, mesh_n=76, sun_deg=1.5, sun_gain=60.0, env_size=(2048, 1024)):
    from PIL import Image
    os.makedirs(os.path.join(root, "scene"), exist_ok=True)
    os.makedirs(os.path.join(root, "synthetic"), exist_ok=True)
    os.makedirs(os.path.join(root, "environment"), exist_ok=True)
    open(os.path.join(root, "synthetic", "cube_sphere.obj"), "w").write(S.cube_sphere_obj(mesh_n))
    open(os.path.join(root, "synthetic", "quad.obj"), "w").write(S.QUAD_OBJ)
    env, w, h = S.synthetic_env(env_size[0], env_size[1], sun_deg=sun_deg, sun_gain=sun_gain)
    Image.fromarray(np.asarray(env, np.uint8).reshape(h, w, 4), "RGBA").save(os.path.join(root, "environment", "sky.RGBE.PNG"))
    cam = S.BUNNY_CAMERA
    scene = {"props": S.bunny_props(), "environment": "environment/sky.RGBE.PNG", "environmentTheta": cam["env_theta"],
             "cameraPos": cam["P"], "cameraDir": cam["I"], "fovScale": cam["fov_scale"], "samples": 20}
    path = os.path.join(root, "scene", "bench.json")
    json.dump(scene, open(path, "w"), indent=1)
    return path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--mesh-n", type=int, default=76)
    ap.add_argument("--sun-deg", type=float, default=1.5)
    ap.add_argument("--sun-gain", type=float, default=60.0)
    a = ap.parse_args()
    print(write(a.dir, a.mesh_n, a.sun_deg, a.sun_gain))
