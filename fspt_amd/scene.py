"""Scene assembly on the host: the Python mirror of the reference's scene
pipeline entry points, driving the native builder in libfspt.

  TexturePacker      texture_packer.js:5-63 (layer indices, flat colours)
  get_material       main.js:206-270 (getMaterial) for colour-valued props
  build_scene        main.js:284-445 (initBVH): parse props, BVH, pack, env bins
  synthetic inputs   SURVEY.md 8d: the real bunny/HDRi blobs are absent from the
                     reference checkout (.MISSING_LARGE_BLOBS), so the benchmark
                     scene is generated: displaced cube-sphere + two quads + a
                     procedural RGBE environment with a sun.
"""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L


@dataclass
class SceneArrays:
    """Exactly what main.js hands to gl.texImage2D (un-padded), plus uniforms."""
    bvh: np.ndarray      # float32 [n_nodes*9] (first 3 words per node are int bits)
    tri: np.ndarray      # float32 [n_tris*9]
    mat: np.ndarray      # float32 [n_tris*12]
    norm: np.ndarray     # float32 [n_tris*27]
    uv: np.ndarray       # float32 [n_tris*6]
    atlas: np.ndarray    # uint8 [layers*res*res*4]
    atlas_res: int
    atlas_layers: int
    env: np.ndarray      # uint8 [h*w*4] RGBE or None
    env_w: int
    env_h: int
    bins: np.ndarray     # uint32 [n_bins*4]
    leaf_size: int = 4
    depth: int = 0
    meta: dict = field(default_factory=dict)

    @property
    def n_nodes(self):
        return self.bvh.size // 9

    @property
    def n_tris(self):
        return self.tri.size // 9

    def desc(self):
        """ctypes fspt_scene_desc viewing these arrays (keep self alive while in use)."""
        d = L.SceneDesc()
        d.bvh = L.fptr(self.bvh); d.n_nodes = self.n_nodes
        d.tri = L.fptr(self.tri); d.n_tris = self.n_tris
        d.mat = L.fptr(self.mat); d.norm = L.fptr(self.norm); d.uv = L.fptr(self.uv)
        d.atlas = L.u8ptr(self.atlas); d.atlas_res = self.atlas_res; d.atlas_layers = self.atlas_layers
        if self.env is not None:
            d.env = L.u8ptr(self.env); d.env_w = self.env_w; d.env_h = self.env_h
        else:
            d.env = None; d.env_w = 0; d.env_h = 0
        d.bins = L.u32ptr(self.bins); d.n_bins = self.bins.size // 4
        d.leaf_size = self.leaf_size
        return d

    def nbytes(self):
        n = self.bvh.nbytes + self.tri.nbytes + self.mat.nbytes + self.norm.nbytes + self.uv.nbytes
        n += self.atlas.nbytes + (self.env.nbytes if self.env is not None else 0) + self.bins.nbytes
        return n


def _js_num(v):
    """Number -> string as Array.prototype.join does (integers without '.0')."""
    f = float(v)
    if f == int(f) and abs(f) < 1e21:
        return str(int(f))
    return repr(f)


class TexturePacker:
    """texture_packer.js:5-63.  Only flat colours are produced natively (images
    need a GL context + decoders in the reference: texture_packer.js:66-185);
    pre-resampled RGBA8 images can be added with add_pixels()."""

    def __init__(self, atlas_res=2048):
        self.res = atlas_res
        self.image_set = []
        self.image_keys = {}
        self.images = {}
        self.max_res = 1

    def _add(self, key, item):
        # `if (this.imageKeys[key])`: index 0 is falsy, so layer 0 is never de-duplicated
        if self.image_keys.get(key):
            return self.image_keys[key]
        self.image_set.append(item)
        self.image_keys[key] = len(self.image_set) - 1
        return self.image_keys[key]

    def add_color(self, color):
        return self._add(" ".join(_js_num(c) for c in color), ("color", [float(c) for c in color]))

    def add_pixels(self, key, rgba):
        """rgba: uint8 [h, w, 4] already in atlas orientation/resolution."""
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        self.max_res = max(self.max_res, rgba.shape[0])
        return self._add(key, ("pixels", rgba))

    _KEEP = object()

    def add_texture(self, key, rgba, corrected=False, swizzle=_KEEP):
        """addTexture (texture_packer.js:13-24): a decoded image, uint8 [h, w, 4] with row 0 = top as an
        HTMLImageElement uploads; key plays the role of image.currentSrc.  corrected = sRGB-decode (diffuse
        maps, main.js:214-219), fixed by the FIRST add of the image.  swizzle = mrSwizzle / pmr_swizzle: the
        reference stores it on the shared image object BEFORE de-duplicating (main.js:226-236) and reads it
        when the atlas is written, so the LAST metallic-roughness use of an image decides (None = identity);
        other uses leave it alone."""
        ent = self.images.get(key)
        if ent is None:
            ent = self.images[key] = {"rgba": np.ascontiguousarray(rgba, dtype=np.uint8), "corrected": None, "swizzle": None}
        if swizzle is not TexturePacker._KEEP:
            ent["swizzle"] = None if swizzle is None else tuple(int(x) for x in swizzle)
        if self.image_keys.get(key):
            return self.image_keys[key]
        self.max_res = max(self.max_res, ent["rgba"].shape[0])
        ent["corrected"] = bool(corrected)  # `image.corrected = corrected` on every real insertion (layer 0 re-adds)
        return self._add(key, ("image", ent))

    def describe(self):
        """Layer list in the form tools/js_ref dumps the reference's imageSet."""
        out = []
        for kind, item in self.image_set:
            if kind == "color":
                out.append({"color": list(item)})
            elif kind == "image":
                key = [k for k, v in self.images.items() if v is item][0]
                out.append({"src": key, "corrected": bool(item["corrected"]),
                            "swizzle": list(item["swizzle"]) if item["swizzle"] else None})
            else:
                out.append({"pixels": True})
        return out

    def get_resolution(self):
        if self.max_res < self.res:
            self.res = self.max_res
        return self.res

    def get_pixels(self):
        res = self.get_resolution()
        out = np.zeros((len(self.image_set), res, res, 4), dtype=np.uint8)
        for i, (kind, item) in enumerate(self.image_set):
            if kind == "color":
                # gl.clearColor(c) + readPixels RGBA8 (texture_packer.js:152-157)
                px = [int(math.floor(min(max(c, 0.0), 1.0) * 255.0 + 0.5)) for c in item[:3]] + [255]
                out[i, :, :, :] = np.array(px, dtype=np.uint8)
            elif kind == "image":
                out[i] = resample_image(item["rgba"], res, item["corrected"], item["swizzle"] or (0, 1, 2, 3))
            else:
                if item.shape[0] != res or item.shape[1] != res:
                    raise ValueError("pre-resampled image must be res x res")
                out[i] = item
        return out.reshape(-1)


def resample_image(rgba, res, corrected=False, swizzle=(0, 1, 2, 3)):
    """WebGLTextureWriter.setAndDrawTexture + the writer shader (texture_packer.js:103-121,159-175):
    bilinear (S = REPEAT, T = CLAMP_TO_EDGE) resample to res x res at uv = (x+.5, res-(y+.5))/res,
    sRGB -> linear before filtering when `corrected` (SRGB8_ALPHA8 upload), channel swizzle,
    rgb premultiplied by alpha, alpha forced to 1, quantised to RGBA8.  Output row 0 = bottom (readPixels)."""
    src = np.asarray(rgba, dtype=np.float32) / np.float32(255.0)
    h, w = src.shape[:2]
    if corrected:  # sRGB EOTF on rgb, alpha linear
        c = src[..., :3]
        src = src.copy()
        src[..., :3] = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(np.float32)
    px = (np.arange(res, dtype=np.float32) + 0.5) / np.float32(res)
    u = px * w - 0.5
    v = (1.0 - px) * h - 0.5
    i0 = np.floor(u).astype(np.int64); a = (u - np.floor(u)).astype(np.float32)
    j0 = np.floor(v).astype(np.int64); b = (v - np.floor(v)).astype(np.float32)
    i1 = (i0 + 1) % w; i0 = i0 % w
    j1 = np.clip(j0 + 1, 0, h - 1); j0 = np.clip(j0, 0, h - 1)
    top = src[j0][:, i0] * (1 - a)[None, :, None] + src[j0][:, i1] * a[None, :, None]
    bot = src[j1][:, i0] * (1 - a)[None, :, None] + src[j1][:, i1] * a[None, :, None]
    c = top * (1 - b)[:, None, None] + bot * b[:, None, None]
    c = c[..., list(swizzle)]
    out = np.empty((res, res, 4), np.float32)
    out[..., :3] = c[..., :3] * c[..., 3:4]
    out[..., 3] = 1.0
    return np.clip(np.floor(out * 255.0 + 0.5), 0, 255).astype(np.uint8)


def _js_parse_float(tok):
    """parseFloat: longest numeric prefix, NaN when none."""
    import re
    m = re.match(r"\s*[+-]?(Infinity|(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?))", tok or "")
    if not m:
        return float("nan")
    try:
        return float(m.group(0))
    except ValueError:  # "1e" style prefixes: parseFloat backs off to the mantissa
        m2 = re.match(r"\s*[+-]?(\d+\.?\d*|\.\d+)", tok)
        return float(m2.group(0)) if m2 else float("nan")


def parse_materials(mtl_text, base_path):
    """ParseMaterials (mtl_loader.js:3-43): {name: {key: value}} and the set of texture urls.
    Values that are falsy in JS (0, NaN, empty) are dropped, as `if (value)` does."""
    materials, urls = {}, []
    scalar = {"ns", "ni", "d", "illum", "dielectric", "ior"}
    vector = {"ka", "kd", "kem", "ks", "ke", "pr", "pm", "pmr", "pmr_swizzle"}
    string = {"map_bump", "map_kd", "map_kem", "map_ks", "map_d", "map_ns", "map_pmr"}
    name = None
    for line in mtl_text.split("\n"):
        tokens = _js_split_spaces(line)
        key = tokens[0].lower()
        if key == "newmtl":
            name = tokens[1] if len(tokens) > 1 else None
            if name is not None:
                materials[name] = {}
        if name:
            value, is_url = None, False
            if key in scalar:
                value = _js_parse_float(tokens[1]) if len(tokens) > 1 else float("nan")
                if value != value or value == 0:
                    value = None
            elif key in vector:
                value = [_js_parse_float(t) for t in tokens[1:]]  # an array is always truthy
            elif key in string:
                value = tokens[1] if len(tokens) > 1 else None
                is_url = True
                if value == "":
                    value = None
            if value is not None:
                if is_url and (base_path + "/" + value) not in urls:
                    urls.append(base_path + "/" + value)
                materials[name][key] = value
    return materials, urls


def _js_split_spaces(line):
    """line.trim().split(/[ ]+/)"""
    import re
    return re.split(r"[ ]+", line.strip(" \t\r\n\f\v\ufeff\xa0"))


# The four atlas layers of a material, in the order their ids are handed out (the packer numbers layers by first use, so
# the order is part of the result).  Per layer the first source that exists wins: the group's MTL image map, the group's
# MTL colour, the prop's scene-JSON field (an image URL, or - where `prop_colour` - a colour), the default colour.
# Same table as fspt_amd/js/fspt.js MATERIAL_LAYERS; semantics: getMaterial, main.js:206-270.
MATERIAL_LAYERS = (
    dict(id="diffuseIndex", map="map_kd", colour="kd", prop="diffuse", prop_colour=True, default=[0.5, 0.5, 0.5], srgb=True),
    dict(id="roughnessIndex", map="map_pmr", colour="pmr", prop="metallicRoughness", prop_colour=True, default=[0.0, 0.3, 0],
         map_swizzle="pmr_swizzle", prop_swizzle="mrSwizzle"),
    dict(id="specularIndex", map="map_kem", colour="kem", prop="emission", prop_colour=False, default=[0, 0, 0]),
    dict(id="normalIndex", map="map_bump", colour=None, prop="normal", prop_colour=False, any_truthy_prop=True, default=[0.5, 0.5, 1]),
)


def get_material(prop, packer, images=None, group_material=None, base_path=""):
    """One OBJ group's material: atlas layer ids + ior / dielectric / emittance.  `images` = {path: uint8 [h, w, 4],
    row 0 = top}; image layers are resampled by the packer (diffuse maps sRGB-decoded, metallic-roughness maps through
    their swizzle).  An array-valued scene-JSON `emission` / `normal` is not a colour source (MATERIAL_LAYERS)."""
    images = images or {}
    mtl = group_material or {}

    def image(url, layer, swizzle):
        if url not in images:
            raise KeyError(f"texture {url!r} is not in `images`")
        kw = {"swizzle": swizzle} if "map_swizzle" in layer else {}
        return packer.add_texture(url, images[url], bool(layer.get("srgb")), **kw)

    out = {}
    for layer in MATERIAL_LAYERS:
        pv = prop.get(layer["prop"])
        if mtl.get(layer["map"]):
            out[layer["id"]] = image(base_path + "/" + mtl[layer["map"]], layer, mtl.get(layer.get("map_swizzle")))
        elif layer["colour"] and mtl.get(layer["colour"]):
            out[layer["id"]] = packer.add_color(mtl[layer["colour"]])
        elif isinstance(pv, str) or (layer.get("any_truthy_prop") and pv):
            out[layer["id"]] = image(pv, layer, prop.get(layer.get("prop_swizzle")))
        elif layer["prop_colour"] and isinstance(pv, (list, tuple, dict)):
            out[layer["id"]] = packer.add_color(pv)
        else:
            out[layer["id"]] = packer.add_color(layer["default"])
    out["ior"] = float(mtl.get("ior") or prop.get("ior") or 1.4)
    out["dielectric"] = float(mtl.get("dielectric") or prop.get("dielectric") or -1)
    out["emittance"] = prop.get("emittance", [0, 0, 0])
    return out


_NORMALS_MODE = {None: 0, "flat": 0, "smooth": 1, "mesh": 2}


def env_bins(env_rgbe, w, h):
    """ProcessEnvRadiance (env_sampler.js) through the native builder."""
    lib = L.lib()
    env_rgbe = np.ascontiguousarray(env_rgbe, dtype=np.uint8).reshape(-1)
    n = C.c_uint32(0)
    L.check(lib.fspt_env_bins(L.u8ptr(env_rgbe), w, h, None, 0, C.byref(n)))
    bins = np.zeros(n.value * 4, dtype=np.uint32)
    L.check(lib.fspt_env_bins(L.u8ptr(env_rgbe), w, h, L.u32ptr(bins), n.value, C.byref(n)))
    return bins


def _rot_array(rot):
    flat = []
    for r in rot:
        flat += [float(r["axis"][0]), float(r["axis"][1]), float(r["axis"][2]), float(r["angle"])]
    arr = (C.c_double * max(len(flat), 1))(*flat)
    return arr, len(rot)


def build_scene(props, obj_texts, env=None, env_w=0, env_h=0, leaf_size=4, atlas_res=2048, images=None,
                world_transforms=None, normalize=None, mtl_texts=None, focus_rays=None):
    """initBVH (main.js:284-445) for props = list of scene-JSON prop dicts and obj_texts = {path: OBJ text}.
    env = RGBE uint8 [h*w*4] or None; world_transforms = scene.worldTransforms; normalize = scene.normalize;
    mtl_texts = {url: MTL text} for `mtllib` lines (url = <dir of the OBJ>/<name>, obj_loader.js:186);
    images = {url: decoded RGBA8 [h, w, 4], row 0 = top} for texture maps; focus_rays = [(eye, dir), ...] ->
    meta["focus"] = shootAutoFocusRay's lensFeatures[0] = 1 - 1/dist for each (main.js:447-546)."""
    lib = L.lib()
    packer = TexturePacker(atlas_res)
    focus = []
    mtl_texts = mtl_texts or {}
    b = C.c_void_p()
    L.check(lib.fspt_builder_create(C.byref(b)))
    keep = []
    try:
        world = world_transforms or []
        wt = (L.WorldTransform * max(len(world), 1))()
        for i, t in enumerate(world):
            if "rotate" in t and t["rotate"] is not None:  # `if (transform.rotate)`: an empty list is truthy
                arr, n = _rot_array(t["rotate"])
                keep.append(arr)
                wt[i].rotate = C.cast(arr, C.POINTER(C.c_double)); wt[i].n_rotate = n; wt[i].has_rotate = 1
            elif t.get("translate"):
                wt[i].translate = (C.c_double * 3)(*[float(x) for x in t["translate"]]); wt[i].has_translate = 1
        for prop in props:
            pd = L.PropDesc()
            rot_arr, n_rot = _rot_array(prop.get("rotate", []))
            pd.rotate = C.cast(rot_arr, C.POINTER(C.c_double))
            pd.n_rotate = n_rot
            pd.scale = float(prop.get("scale", 1.0))
            tr = prop.get("translate", [0, 0, 0])
            pd.translate = (C.c_double * 3)(*[float(x) for x in tr])
            pd.normals_mode = _NORMALS_MODE[prop.get("normals")]
            text = obj_texts[prop["path"]].encode("utf-8")
            skips = [str(x).encode("utf-8") for x in (prop.get("skips") or [])]
            skip_arr = (C.c_char_p * max(len(skips), 1))(*skips)
            ng = C.c_uint32()
            L.check(lib.fspt_builder_parse_obj(b, text, len(text), C.byref(pd), wt, len(world), skip_arr, len(skips),
                                               C.byref(ng)))
            base_path = "/".join(prop["path"].split("/")[:-1])
            libs = {}
            mats = (L.GroupMaterial * max(ng.value, 1))()
            for g in range(ng.value):
                name, nt, mi = C.c_char_p(), C.c_uint32(), C.c_int32()
                L.check(lib.fspt_builder_group_info(b, g, C.byref(name), C.byref(nt), C.byref(mi)))
                gm = {}
                if mi.value >= 0:
                    if mi.value not in libs:
                        ln = C.c_char_p()
                        L.check(lib.fspt_builder_mtllib_name(b, mi.value, C.byref(ln)))
                        url = base_path + "/" + ln.value.decode("utf-8")
                        if url not in mtl_texts:
                            raise KeyError(f"mtllib {url!r} is not in `mtl_texts`")
                        libs[mi.value] = parse_materials(mtl_texts[url], base_path)[0]
                    gm = libs[mi.value].get(name.value.decode("utf-8"), {})
                m = get_material(prop, packer, images, gm, base_path)
                mats[g].diffuse_layer = m["diffuseIndex"]; mats[g].emissive_layer = m["specularIndex"]
                mats[g].normal_layer = m["normalIndex"]; mats[g].mr_layer = m["roughnessIndex"]
                mats[g].emittance = (C.c_double * 3)(*[float(x) for x in m["emittance"]])
                mats[g].ior = m["ior"]; mats[g].dielectric = m["dielectric"]
            L.check(lib.fspt_builder_commit_obj(b, mats, ng.value))
        if normalize:
            L.check(lib.fspt_builder_normalize(b, float(normalize)))
        L.check(lib.fspt_builder_build(b, leaf_size))
        nn, nt, dp = C.c_uint32(), C.c_uint32(), C.c_uint32()
        L.check(lib.fspt_builder_counts(b, C.byref(nn), C.byref(nt), C.byref(dp)))
        bvh = np.zeros(nn.value * 9, np.float32); tri = np.zeros(nt.value * 9, np.float32)
        mat = np.zeros(nt.value * 12, np.float32); norm = np.zeros(nt.value * 27, np.float32)
        uv = np.zeros(nt.value * 6, np.float32)
        L.check(lib.fspt_builder_get(b, L.fptr(bvh), L.fptr(tri), L.fptr(mat), L.fptr(norm), L.fptr(uv)))
        for eye, d in (focus_rays or []):
            dist = C.c_double()
            L.check(lib.fspt_builder_autofocus(b, (C.c_double * 3)(*[float(x) for x in eye]),
                                               (C.c_double * 3)(*[float(x) for x in d]), C.byref(dist)))
            focus.append(1 - 1 / dist.value)
    finally:
        lib.fspt_builder_destroy(b)
    atlas = packer.get_pixels()
    if env is not None:
        env = np.ascontiguousarray(env, dtype=np.uint8).reshape(-1)
        bins = env_bins(env, env_w, env_h)
    else:
        bins = np.array([0, 0, 1, 2048], dtype=np.uint32)  # main.js:292
    return SceneArrays(bvh=bvh, tri=tri, mat=mat, norm=norm, uv=uv, atlas=atlas, atlas_res=packer.res,
                       atlas_layers=len(packer.image_set), env=env, env_w=env_w, env_h=env_h, bins=bins,
                       leaf_size=leaf_size, depth=dp.value, meta={"layers": packer.describe(), "focus": focus})


def build_scene_json(scene, obj_texts, mtl_texts=None, images=None, env=None, env_w=0, env_h=0, leaf_size=4,
                     focus_rays=None):
    """build_scene for a whole scene JSON (props / static_props / animated_props, worldTransforms, normalize,
    atlasRes: main.js:284-445,869-871,944)."""
    return build_scene(merge_scene_props(scene), obj_texts, env=env, env_w=env_w, env_h=env_h, leaf_size=leaf_size,
                       atlas_res=scene.get("atlasRes") or 2048, images=images,
                       world_transforms=scene.get("worldTransforms"), normalize=scene.get("normalize"),
                       mtl_texts=mtl_texts, focus_rays=focus_rays)


def merge_scene_props(scene):
    """mergeSceneProps (main.js:869-871): props + static_props + the values of animated_props."""
    anim = scene.get("animated_props") or []
    anim = list(anim.values()) if isinstance(anim, dict) else list(anim)
    return list(scene.get("props") or []) + list(scene.get("static_props") or []) + anim


# ---------------------------------------------------------------------------
# synthetic inputs
# ---------------------------------------------------------------------------
QUAD_OBJ = "\n".join([
    "v 0.5 0.0 0.5", "v 0.5 0.0 -0.5", "v -0.5 0.0 -0.5", "v -0.5 0.0 0.5", "",
    "vt 0.0 0.0", "vt 0.0 1.0", "vt 1.0 1.0", "vt 1.0 0.0", "",
    "f 1/1 3/3 2/2", "f 3/3 1/1 4/4", ""])
"""A unit quad in the y=0 plane with uvs: the geometry of the reference's
asset_packs/misc/top_mono.obj (2 triangles)."""


def cube_sphere_obj(n, bump=0.05):
    """Displaced cube-sphere as OBJ text: 12*n^2 triangles, closed manifold, no
    poles (a UV-sphere's zero-area pole fans give NaN normals in
    obj_loader.js:40-44).  r = 1 + bump*sin(9*theta)*sin(7*phi)."""
    index = {}
    verts = []

    def vid(i, j, k):
        key = (i, j, k)
        if key not in index:
            x, y, z = (2.0 * i / n - 1.0), (2.0 * j / n - 1.0), (2.0 * k / n - 1.0)
            l = math.sqrt(x * x + y * y + z * z)
            x, y, z = x / l, y / l, z / l
            theta = math.atan2(z, x)
            phi = math.acos(max(-1.0, min(1.0, y)))
            r = 1.0 + bump * math.sin(9.0 * theta) * math.sin(7.0 * phi)
            verts.append((x * r, y * r, z * r))
            index[key] = len(verts)
        return index[key]

    faces = []
    for axis in range(3):
        for side in (0, n):
            for a in range(n):
                for b in range(n):
                    def p(u, v):
                        c = [0, 0, 0]
                        c[axis] = side
                        c[(axis + 1) % 3] = u
                        c[(axis + 2) % 3] = v
                        return vid(*c)
                    q = [p(a, b), p(a + 1, b), p(a + 1, b + 1), p(a, b + 1)]
                    if side == 0:
                        q = q[::-1]
                    faces.append((q[0], q[1], q[2]))
                    faces.append((q[0], q[2], q[3]))
    lines = ["v %r %r %r" % v for v in verts]
    lines += ["f %d %d %d" % f for f in faces]
    return "\n".join(lines) + "\n"


def synthetic_env(w=2048, h=1024, sun_deg=1.5, sun_gain=60.0, sun_dir=(0.35, 0.55)):
    """Procedural RGBE-in-RGBA8 equirect: sky gradient + ground + one sun.
    Encoding is the inverse of tracer.fs:412 / env_sampler.js:17-20:
    E = ceil(log2(max)) + 128, rgb = round(c / 2^(E-128) * 255)."""
    v = (np.arange(h, dtype=np.float64) + 0.5) / h
    u = (np.arange(w, dtype=np.float64) + 0.5) / w
    V, U = np.meshgrid(v, u, indexing="ij")
    # tracer.fs:417: v = asin(-dir.y)/pi + 0.5  -> row 0 is straight up
    elev = (0.5 - V) * math.pi
    up = np.clip(np.sin(elev), -1, 1)
    sky = np.stack([0.35 + 0.25 * (1 - up), 0.55 + 0.2 * (1 - up), 0.9 + 0.05 * (1 - up)], -1)
    ground = np.stack([0.22 + 0 * up, 0.2 + 0 * up, 0.17 + 0 * up], -1)
    img = np.where((up > 0)[..., None], sky, ground)
    su, sv = sun_dir
    du = np.minimum(np.abs(U - su), 1 - np.abs(U - su)) * 2 * math.pi * np.cos(elev)
    dv = (V - (0.5 - sv / 2)) * math.pi
    ang = np.sqrt(du * du + dv * dv)
    sun = (ang < math.radians(sun_deg))[..., None]
    img = np.where(sun, np.array([1.0, 0.93, 0.8]) * sun_gain, img)
    mx = np.maximum(img.max(-1), 1e-6)
    e = np.ceil(np.log2(mx))
    scale = np.exp2(e)
    rgb = np.clip(np.floor(img / scale[..., None] * 255.0 + 0.5), 0, 255)
    out = np.concatenate([rgb, (e + 128)[..., None]], -1).astype(np.uint8)
    return out.reshape(-1), w, h


def textured_test_scene(res=16):
    """Two image-mapped quads + a flat-colour sphere: exercises the bilinear RGBA8 atlas path (atlas res > 1),
    tangent-space normal mapping and emissive maps with procedurally generated images."""
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:res, 0:res]
    checker = (((xx // 2) + (yy // 2)) % 2).astype(np.uint8)
    diffuse = np.stack([40 + 180 * checker, 200 - 120 * checker, 90 + xx * 8, np.full_like(checker, 255)], -1).astype(np.uint8)
    mr = np.stack([(xx > res // 2) * 255, 30 + yy * 10, np.zeros_like(xx), np.full_like(xx, 255)], -1).astype(np.uint8)
    nrm = np.stack([128 + (rng.integers(-40, 41, (res, res))), 128 + (rng.integers(-40, 41, (res, res))),
                    np.full((res, res), 230), np.full((res, res), 255)], -1).astype(np.uint8)
    emis = np.zeros((res, res, 4), np.uint8); emis[..., 3] = 255; emis[res // 4: res // 2, res // 4: res // 2, :3] = (255, 160, 40)
    images = {"tex/diffuse.png": diffuse, "tex/mr.png": mr, "tex/normal.png": nrm, "tex/emissive.png": emis}
    props = [
        {"path": "synthetic/quad.obj", "scale": 4, "rotate": [], "translate": [0, -0.75, 0], "emittance": [0, 0, 0],
         "diffuse": "tex/diffuse.png", "metallicRoughness": "tex/mr.png", "normal": "tex/normal.png", "normals": "flat"},
        {"path": "synthetic/quad.obj", "scale": 4, "rotate": [{"angle": -1.57, "axis": [1, 0, 0]}], "translate": [0, 0.25, -1],
         "emittance": [0, 0, 0], "diffuse": "tex/diffuse.png", "emission": "tex/emissive.png",
         "metallicRoughness": [0, 0.4, 0], "normal": "tex/normal.png", "normals": "flat", "ior": 10},
        {"path": "synthetic/cube_sphere.obj", "scale": 0.35, "rotate": [], "translate": [0.1, -0.4, 0],
         "diffuse": [0.9, 0.9, 0.9], "emittance": [0, 0, 0], "metallicRoughness": [0, 0.2, 0], "normals": "smooth"},
    ]
    texts = {"synthetic/cube_sphere.obj": cube_sphere_obj(6), "synthetic/quad.obj": QUAD_OBJ}
    env, w, h = synthetic_env(64, 32)
    s = build_scene(props, texts, env=env, env_w=w, env_h=h, images=images)
    s.meta = dict(kind="textured-test")
    return s


def bunny_props():
    """scene/bunny.json:6-41 with the missing bunny mesh replaced by the
    cube-sphere and the missing image maps by flat colours."""
    return [
        {"path": "synthetic/cube_sphere.obj", "scale": 0.35, "rotate": [{"angle": 0, "axis": [0, 0, 1]}],
         "translate": [0.1, -0.7, 0], "diffuse": [1, 1, 1], "emittance": [0, 0, 0],
         "metallicRoughness": [0, 0.1, 0], "ior": 1.4, "normals": "smooth"},
        {"path": "synthetic/quad.obj", "scale": 4, "rotate": [{"angle": 3.1415, "axis": [0, 0, 1]}],
         "translate": [0, -0.75, 0], "emittance": [0, 0, 0], "diffuse": [0.5, 0.5, 0.5],
         "metallicRoughness": [0, 0.3, 0], "normals": "flat"},
        {"path": "synthetic/quad.obj", "scale": 4, "rotate": [{"angle": -1.57, "axis": [1, 0, 0]}],
         "translate": [0, 0.25, -1], "emittance": [0, 0, 0], "diffuse": [0.5, 0.5, 0.5],
         "metallicRoughness": [0, 0.3, 0], "normals": "flat", "ior": 10},
    ]


def procedural_maps(res=2048, seed=11):
    """Stand-ins for the reference's dungeon texture set (scene/bunny.json:24-36: 2048^2 baseColor / metallicRoughness /
    normal / emissive images; the normal maps are missing blobs in the checkout): flagstones with mortar lines, per-stone
    tint and roughness, a tangent-space normal map from the same height field, a few emissive runes.
    Returns {url: RGBA8 [res, res, 4]} for both image-mapped props."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:res, 0:res].astype(np.float32)
    cells = 16
    cx, cy = xx * cells / res, yy * cells / res
    ix, iy = np.floor(cx).astype(np.int32), np.floor(cy).astype(np.int32)
    fx, fy = cx - ix, cy - iy
    edge = np.minimum(np.minimum(fx, 1 - fx), np.minimum(fy, 1 - fy))      # distance to the mortar line, in cells
    mortar = np.clip(1.0 - edge / 0.06, 0.0, 1.0)
    tint = rng.uniform(0.55, 1.0, (cells, cells, 3)).astype(np.float32)[iy % cells, ix % cells]
    rough_stone = rng.uniform(0.25, 0.8, (cells, cells)).astype(np.float32)[iy % cells, ix % cells]
    grain = (np.sin(xx * 0.37 + np.sin(yy * 0.11) * 3.0) * np.sin(yy * 0.29 + xx * 0.05) * 0.5 + 0.5).astype(np.float32)
    height = (1.0 - mortar) * (0.8 + 0.2 * grain)
    gy, gx = np.gradient(height)
    nrm = np.stack([-gx * 6.0, -gy * 6.0, np.ones_like(gx)], -1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)

    def rgba(c):
        c = np.clip(c, 0.0, 1.0)
        a = np.full(c.shape[:2] + (1,), 1.0, np.float32)
        return np.round(np.concatenate([c, a], -1) * 255.0).astype(np.uint8)

    def maps(base_rgb, metal):
        base = (tint * base_rgb * (0.75 + 0.25 * grain[..., None])) * (1.0 - 0.6 * mortar[..., None])
        mr = np.stack([np.full_like(grain, metal) * (1.0 - mortar), np.clip(rough_stone + 0.3 * mortar, 0, 1), np.zeros_like(grain)], -1)
        return rgba(base), rgba(mr), rgba(nrm * [0.5, 0.5, 1.0] + [0.5, 0.5, 0.0])

    out = {}
    d, m, n = maps(np.array([0.62, 0.58, 0.52], np.float32), 0.0)
    out["asset_packs/dungeon/RootNode_baseColor.png"] = d
    out["asset_packs/dungeon/RootNode_metallicRoughness.png"] = m
    out["asset_packs/dungeon/RootNode_normal.png"] = n
    d, m, n = maps(np.array([0.45, 0.5, 0.6], np.float32), 0.35)
    out["asset_packs/dungeon/Scene_-_Root_baseColor.jpeg"] = d
    out["asset_packs/dungeon/Scene_-_Root_metallicRoughness.png"] = m
    out["asset_packs/dungeon/Scene_-_Root_normal.png"] = n
    rune = ((ix + 3 * iy) % 11 == 0) & (np.abs(fx - 0.5) < 0.12) & (np.abs(fy - 0.5) < 0.3)
    out["asset_packs/dungeon/Scene_-_Root_emissive.jpeg"] = rgba(rune[..., None] * np.array([1.0, 0.55, 0.15], np.float32))
    return out


def bunny_props_textured():
    """scene/bunny.json:6-41 with the floor / wall quads image-mapped exactly as the reference lists them (baseColor,
    metallicRoughness, normal; the wall also an emissive map); the bunny stays a flat-colour cube-sphere."""
    p = bunny_props()
    p[1].update(diffuse="asset_packs/dungeon/RootNode_baseColor.png",
                metallicRoughness="asset_packs/dungeon/RootNode_metallicRoughness.png",
                normal="asset_packs/dungeon/RootNode_normal.png")
    p[2].update(emission="asset_packs/dungeon/Scene_-_Root_emissive.jpeg",
                diffuse="asset_packs/dungeon/Scene_-_Root_baseColor.jpeg",
                metallicRoughness="asset_packs/dungeon/Scene_-_Root_metallicRoughness.png",
                normal="asset_packs/dungeon/Scene_-_Root_normal.png")
    return p


def bunny_scene_textured(n=76, env_size=(2048, 1024), sun_deg=1.5, sun_gain=60.0, res=2048):
    """The 'bunny' configs with the reference's real atlas size: 2048^2 image maps on both quads (7 image layers + the
    flat-colour layers, 16 MB each) - the 4 x 4-tap bilinear atlas gather of tracer.fs:453-456 on an atlas that does not
    fit any cache."""
    texts = {"synthetic/cube_sphere.obj": cube_sphere_obj(n), "synthetic/quad.obj": QUAD_OBJ}
    env, w, h = synthetic_env(env_size[0], env_size[1], sun_deg=sun_deg, sun_gain=sun_gain)
    s = build_scene(bunny_props_textured(), texts, env=env, env_w=w, env_h=h, images=procedural_maps(res), atlas_res=res)
    s.meta = dict(kind="bunny-synthetic-textured", n=n, res=res)
    return s


BUNNY_CAMERA = dict(P=[-0.751, 0.665, 1.820], I=[0.304, -0.489, -0.818], fov_scale=0.5, env_theta=1.66,
                    focal_depth=2.0, aperture=0.02)
"""scene/bunny.json:3-5, index.html:25,27, main.js:67-74."""


def lens_features(focal_depth, aperture):
    """main.js:74: lensFeatures = [1 - 1/focalDepth, apertureSize]."""
    return [1.0 - 1.0 / focal_depth, aperture]


def bunny_scene(n=76, env_size=(2048, 1024), sun_deg=1.5, sun_gain=60.0):
    """The BASELINE 'bunny' configs: n=76 -> 69 312 + 4 triangles (C1/C2/C5),
    n=289 -> 1 002 252 + 4 (C3)."""
    texts = {"synthetic/cube_sphere.obj": cube_sphere_obj(n), "synthetic/quad.obj": QUAD_OBJ}
    env, w, h = synthetic_env(env_size[0], env_size[1], sun_deg=sun_deg, sun_gain=sun_gain)
    s = build_scene(bunny_props(), texts, env=env, env_w=w, env_h=h)
    s.meta = dict(kind="bunny-synthetic", n=n)
    return s
