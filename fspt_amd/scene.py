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

    def add_texture(self, key, rgba, corrected=False, swizzle=None):
        """addTexture (texture_packer.js:13-24): a decoded image, uint8 [h, w, 4] with row 0 = top as an
        HTMLImageElement uploads; key plays the role of image.currentSrc.  corrected = sRGB-decode (diffuse
        maps, main.js:214-219); swizzle = mrSwizzle / pmr_swizzle (main.js:226-236)."""
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        if self.image_keys.get(key):
            return self.image_keys[key]
        self.max_res = max(self.max_res, rgba.shape[0])
        return self._add(key, ("image", (rgba, bool(corrected), tuple(swizzle) if swizzle else (0, 1, 2, 3))))

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
                out[i] = resample_image(item[0], res, item[1], item[2])
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


def get_material(prop, packer, images=None):
    """getMaterial (main.js:206-270).  Colour-valued maps become flat layers; string-valued maps name an
    entry of `images` ({path: uint8 [h, w, 4], row 0 = top}) and become resampled image layers (diffuse maps
    sRGB-decoded, main.js:214-219; metallicRoughness honours mrSwizzle, main.js:232-236)."""
    images = images or {}

    def layer(value, default, corrected=False, swizzle=None):
        if isinstance(value, str):
            return packer.add_texture(value, images[value], corrected, swizzle)
        if isinstance(value, (list, tuple)):
            return packer.add_color(value)
        return packer.add_color(default)
    diffuse = layer(prop.get("diffuse"), [0.5, 0.5, 0.5], corrected=True)
    rough = layer(prop.get("metallicRoughness"), [0.0, 0.3, 0], swizzle=prop.get("mrSwizzle"))
    em = prop.get("emission")
    spec = packer.add_texture(em, images[em]) if isinstance(em, str) else packer.add_color(
        em if isinstance(em, (list, tuple)) else [0, 0, 0])
    nm = prop.get("normal")
    normal = packer.add_texture(nm, images[nm]) if isinstance(nm, str) else packer.add_color([0.5, 0.5, 1])
    ior = prop.get("ior") or 1.4
    dielectric = prop.get("dielectric") or -1
    return dict(diffuseIndex=diffuse, roughnessIndex=rough, specularIndex=spec, normalIndex=normal,
                ior=float(ior), dielectric=float(dielectric), emittance=prop.get("emittance", [0, 0, 0]))


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


def build_scene(props, obj_texts, env=None, env_w=0, env_h=0, leaf_size=4, atlas_res=2048, images=None):
    """initBVH (main.js:284-445) for props = list of scene-JSON prop dicts and
    obj_texts = {path: OBJ text}.  env = RGBE uint8 [h*w*4] or None."""
    lib = L.lib()
    packer = TexturePacker(atlas_res)
    b = C.c_void_p()
    L.check(lib.fspt_builder_create(C.byref(b)))
    try:
        for prop in props:
            m = get_material(prop, packer, images)
            pd = L.PropDesc()
            rot = prop.get("rotate", [])
            flat = []
            for r in rot:
                flat += [float(r["axis"][0]), float(r["axis"][1]), float(r["axis"][2]), float(r["angle"])]
            rot_arr = (C.c_double * max(len(flat), 1))(*flat)
            pd.rotate = C.cast(rot_arr, C.POINTER(C.c_double))
            pd.n_rotate = len(rot)
            pd.scale = float(prop.get("scale", 1.0))
            tr = prop.get("translate", [0, 0, 0])
            pd.translate = (C.c_double * 3)(*[float(x) for x in tr])
            pd.normals_mode = _NORMALS_MODE[prop.get("normals")]
            pd.diffuse_layer = m["diffuseIndex"]; pd.emissive_layer = m["specularIndex"]
            pd.normal_layer = m["normalIndex"]; pd.mr_layer = m["roughnessIndex"]
            pd.emittance = (C.c_double * 3)(*[float(x) for x in m["emittance"]])
            pd.ior = m["ior"]; pd.dielectric = m["dielectric"]
            text = obj_texts[prop["path"]].encode("utf-8")
            L.check(lib.fspt_builder_add_obj(b, text, len(text), C.byref(pd)))
        L.check(lib.fspt_builder_build(b, leaf_size))
        nn, nt, dp = C.c_uint32(), C.c_uint32(), C.c_uint32()
        L.check(lib.fspt_builder_counts(b, C.byref(nn), C.byref(nt), C.byref(dp)))
        bvh = np.zeros(nn.value * 9, np.float32); tri = np.zeros(nt.value * 9, np.float32)
        mat = np.zeros(nt.value * 12, np.float32); norm = np.zeros(nt.value * 27, np.float32)
        uv = np.zeros(nt.value * 6, np.float32)
        L.check(lib.fspt_builder_get(b, L.fptr(bvh), L.fptr(tri), L.fptr(mat), L.fptr(norm), L.fptr(uv)))
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
                       leaf_size=leaf_size, depth=dp.value)


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
