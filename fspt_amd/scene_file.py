"""Scene files and frame sequences: the host steps either side of the path tracer that the reference does in
`PathTracer()` / `start()` / `tick()` (main.js:17-75, 284-445, 838-866, 869-975) when it is pointed at
`scene/<name>.json?frame=N`:

  * read the scene JSON, merge props + static_props + animated_props (main.js:869-871);
  * fetch every asset it names: OBJ texts, the MTL libraries their `mtllib` lines name, texture images
    (prop-level `diffuse` / `metallicRoughness` / `normal` / `emission` strings and MTL `map_*` urls), the RGBE
    environment image (main.js:926-946, obj_loader.js:185-190);
  * build the scene (initBVH) and shoot the auto-focus ray (main.js:903);
  * per frame: render `samples` ticks, tone-map with draw.fs, write the image, go to frame + 1 (main.js:851-866).

Image files are decoded with PIL to straight-alpha RGBA8, row 0 = top - what a browser hands to texImage2D.
Paths in the JSON are relative to the web root (`asset_root`, default: the parent of the scene file's folder)."""
import json
import os

import numpy as np

from . import scene as S


def _read_text(root, rel):
    with open(os.path.join(root, rel), "r", encoding="utf-8", errors="replace") as fh:
        return fh.read()


def _read_image(root, rel):
    from PIL import Image
    with Image.open(os.path.join(root, rel)) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))


def mtllib_urls(obj_text, base_path):
    """The urls obj_loader.js:185-187 fetches while parsing: base_path + '/' + the rest of each `mtllib` line."""
    urls = []
    for line in obj_text.split("\n"):
        tok = S._js_split_spaces(line)
        if tok[0] == "mtllib":
            urls.append(base_path + "/" + " ".join(tok[1:]))
    return urls


def load_scene_file(scene_path, asset_root=None, leaf_size=4):
    """Returns (SceneArrays, settings).  settings = camera and display values with the reference's defaults
    (initGlobals, main.js:50-75): eye, dir, fov_scale, env_theta, exposure, samples, focus (lensFeatures[0]
    after shootAutoFocusRay), aperture (index.html default 0.02)."""
    with open(scene_path, "r", encoding="utf-8") as fh:
        scene = json.load(fh)
    root = asset_root or os.path.dirname(os.path.dirname(os.path.abspath(scene_path)))
    props = S.merge_scene_props(scene)
    obj_texts, mtl_texts, images = {}, {}, {}

    def want_image(url):
        if url not in images:
            images[url] = _read_image(root, url)

    for p in props:
        if p["path"] not in obj_texts:
            obj_texts[p["path"]] = _read_text(root, p["path"])
        base = "/".join(p["path"].split("/")[:-1])
        for url in mtllib_urls(obj_texts[p["path"]], base):
            if url not in mtl_texts:
                mtl_texts[url] = _read_text(root, url)
            for tex_url in S.parse_materials(mtl_texts[url], base)[1]:
                want_image(tex_url)
        for key in ("diffuse", "metallicRoughness"):
            if isinstance(p.get(key), str):
                want_image(p[key])
        for key in ("normal", "emission"):
            if p.get(key) and isinstance(p[key], str):
                want_image(p[key])
    env, env_w, env_h = None, 0, 0
    e = scene.get("environment")
    if isinstance(e, str):
        img = _read_image(root, e)
        env_h, env_w = img.shape[:2]
        env = img.reshape(-1)
    elif e:
        # array-of-stops skies (main.js:182-204) are broken in the reference itself
        raise ValueError("array-of-stops environments are not supported; use an RGBE image or none")
    eye = [float(x) for x in (scene.get("cameraPos") or [0, 0, 2])]
    d = [float(x) for x in (scene.get("cameraDir") or [0, 0, -1])]
    arrays = S.build_scene_json(scene, obj_texts, mtl_texts, images, env=env, env_w=env_w, env_h=env_h,
                                leaf_size=leaf_size, focus_rays=[(eye, d)])
    settings = dict(eye=eye, dir=d, fov_scale=float(scene.get("fovScale") or 0.5),
                    env_theta=float(scene.get("environmentTheta") or 0), exposure=float(scene.get("exposure") or 1.0),
                    samples=int(scene.get("samples") or 2000), focus=arrays.meta["focus"][0], aperture=0.02)
    return arrays, settings


def render_frame(arrays, settings, width, height, samples=None, bounces=4, seed=1, saturation=1.0, denoise=False,
                 max_sigma=3.0, device=0):
    """One frame as the reference produces it in frame mode: `samples` ticks from a cleared accumulator
    (main.js:838-857; its very first, discarded tick is not reproduced), then drawQuad.  Returns
    (rgba8 [H, W, 4] top row first - what canvas.toBlob encodes -, radiance [H, W, 4] bottom row first)."""
    from .tracer import PathTracer
    pt = PathTracer(arrays, width, height, device=device, num_bounces=bounces)
    try:
        pt.eye, pt.dir = list(settings["eye"]), list(settings["dir"])
        pt.fovScale, pt.envTheta = settings["fov_scale"], settings["env_theta"]
        pt.lensFeatures = [settings["focus"], settings["aperture"]]
        pt.seed(seed)
        pt.render(int(samples if samples is not None else settings["samples"]))
        rgba = pt.draw(settings["exposure"], saturation, denoise, max_sigma)
        rad = pt.readRadiance()
    finally:
        pt.close()
        pt.scene.close()
    return rgba[::-1].copy(), rad


def render_sequence(scene_pattern, frames, out_pattern, width, height, asset_root=None, **kw):
    """frame=N sequencing (main.js:851-866, 966-969): for every N in `frames` load `scene_pattern.format(frame=N)`
    (the per-frame scene JSON the reference's server hands out for `?frame=N`), render it, write
    `out_pattern.format(frame=N)` (the reference POSTs the canvas PNG to /upload/<scene>/<N>), go on to N + 1."""
    from PIL import Image
    written = []
    for n in frames:
        arrays, settings = load_scene_file(scene_pattern.format(frame=n), asset_root)
        rgba, _ = render_frame(arrays, settings, width, height, **kw)
        out = out_pattern.format(frame=n)
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        Image.fromarray(rgba[:, :, :3]).save(out)
        written.append(out)
    return written
