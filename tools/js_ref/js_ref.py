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


def _cut_function(src, name):
    """The source text of `function name(...) {...}` inside main.js (brace matching)."""
    a = src.index("function " + name + "(")
    i = src.index("{", a)
    depth = 0
    while True:
        c = src[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return src[a:i + 1]
        i += 1


def _cut_block(src, head, start=0):
    """The source text from `head` (e.g. 'if (scene.normalize) {', 'for (...) {') to its matching closing brace."""
    a = src.index(head, start)
    i = src.index("{", a + len(head) - 1)
    depth = 0
    while True:
        c = src[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return src[a:i + 1], i + 1
        i += 1


def run(job, max_old_space_mb=6000, raw=False):
    """raw: the packed arrays come back through binary files instead of base64 in the JSON (large scenes)."""
    with tempfile.TemporaryDirectory() as td:
        if raw:
            job = dict(job, raw_dir=td)
        for f in ("vector.js", "bvh.js", "obj_loader.js", "mtl_loader.js", "utility.js", "env_sampler.js",
                  "texture_packer.js"):
            shutil.copy(os.path.join(REF, f), td)
        # main.js cannot be imported (DOM at import time): lift its pure functions, unmodified, into a module
        main_src = open(os.path.join(REF, "main.js")).read()
        with open(os.path.join(td, "ref_functions.js"), "w") as fh:
            fh.write("export " + _cut_function(main_src, "getMaterial") + "\n")
            fh.write("export " + _cut_function(main_src, "mergeSceneProps") + "\n")
            # shootAutoFocusRay closes over PathTracer's locals: give it the same names
            fh.write("export function autoFocus(Vec3, bvh, eye, dir) {\n  const maxT = 1e6;\n  let lensFeatures = [0, 0];\n"
                     "  let elements = { focalDepthElement: {} };\n" + _cut_function(main_src, "shootAutoFocusRay") +
                     "\n  shootAutoFocusRay();\n  return lensFeatures[0];\n}\n")
            # initBVH's scene.normalize block (main.js:337-348) and its packing loops (main.js:358-392) are statements
            # inside initBVH, not functions: their text is cut out the same way and given the names they use
            init = main_src[main_src.index("async function initBVH("):]
            norm, _ = _cut_block(init, "if (scene.normalize) {")
            fh.write("export function normalizeScene(Vec3, scene, bounds, geometry) {\n" + norm + "\n}\n")
            a = init.index("let bvhArray = bvh.serializeTree();")
            loop, end = _cut_block(init, "for (let i = 0; i < bvhArray.length; i++) {", a)
            fh.write("export function packScene(bvh) {\n  let time = 0;\n" + init[a:end] +
                     "\n  return { bvhBuffer, trianglesBuffer, materialBuffer, normalBuffer, uvBuffer };\n}\n")
            fh.write("export " + _cut_function(main_src, "maskBVHBuffer") + "\n")
        shutil.copy(os.path.join(HERE, "driver.js"), td)
        with open(os.path.join(td, "package.json"), "w") as fh:
            fh.write('{"type":"module"}')
        with open(os.path.join(td, "job.json"), "w") as fh:
            json.dump(job, fh)
        subprocess.check_call(["node", f"--max-old-space-size={max_old_space_mb}", "--experimental-modules",
                               "driver.js", "job.json", "out.json"], cwd=td,
                              stderr=None if os.environ.get("JS_REF_DEBUG") else subprocess.DEVNULL)
        out = json.load(open(os.path.join(td, "out.json")))
        for k in ("bvh", "tri", "mat", "norm", "uv"):
            if isinstance(out.get(k), dict):
                out[k] = np.fromfile(os.path.join(td, out[k]["file"]), dtype=np.float32)
    for k in ("bvh", "tri", "mat", "norm", "uv"):
        if k in out and not isinstance(out[k], np.ndarray):
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


def full_scene_job(scene, obj_texts, files=None, images=None, leaf_size=4, autofocus=None):
    """scene: the scene JSON (props / static_props / animated_props, worldTransforms, normalize, atlasRes);
    files: {url: text} served to obj_loader.js's mtllib fetches; images: {url: {"height": h}};
    autofocus: [[eye, dir], ...] -> out["autofocus"] = lensFeatures[0] of shootAutoFocusRay for each."""
    return {"scene": scene, "objs": obj_texts, "files": files or {}, "images": images or {}, "leaf_size": leaf_size,
            "autofocus": autofocus or []}


def env_job(rgba, w, h):
    return {"env": {"rgba_b64": base64.b64encode(np.ascontiguousarray(rgba, np.uint8).tobytes()).decode(),
                    "width": w, "height": h}}
