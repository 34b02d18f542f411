"""Runs the reference's JS scene pipeline under Node (build-container tooling)."""
import base64
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(REF) and shutil.which("node") is not None


def run(job, max_old_space_mb=6000):
    with tempfile.TemporaryDirectory() as td:
        for f in ("vector.js", "bvh.js", "obj_loader.js", "mtl_loader.js", "utility.js", "env_sampler.js"):
            shutil.copy(os.path.join(REF, f), td)
        shutil.copy(os.path.join(HERE, "driver.js"), td)
        with open(os.path.join(td, "package.json"), "w") as fh:
            fh.write('{"type":"module"}')
        with open(os.path.join(td, "job.json"), "w") as fh:
            json.dump(job, fh)
        subprocess.check_call(["node", f"--max-old-space-size={max_old_space_mb}", "--experimental-modules",
                               "driver.js", "job.json", "out.json"], cwd=td, stderr=subprocess.DEVNULL)
        out = json.load(open(os.path.join(td, "out.json")))
    for k in ("bvh", "tri", "mat", "norm", "uv"):
        if k in out:
            out[k] = np.frombuffer(base64.b64decode(out[k]), dtype=np.float32).copy()
    if "bins" in out:
        out["bins"] = np.array(out["bins"], dtype=np.uint32)
    return out


def scene_job(props, obj_texts, materials, leaf_size=4):
    """props: scene-JSON prop dicts; materials: getMaterial results (one per prop)."""
    jp = []
    for p, m in zip(props, materials):
        q = dict(p)
        q["material"] = m
        jp.append(q)
    return {"props": jp, "objs": obj_texts, "leaf_size": leaf_size}


def env_job(rgba, w, h):
    return {"env": {"rgba_b64": base64.b64encode(np.ascontiguousarray(rgba, np.uint8).tobytes()).decode(),
                    "width": w, "height": h}}
