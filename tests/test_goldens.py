"""Pins against the REFERENCE run in the build container (tests/golden/, made by
tools/make_goldens.py; nothing here reads /root/reference):

  D0  native scene pipeline == reference JS pipeline (bvh.js / obj_loader.js /
      env_sampler.js under Node), byte for byte
  D2  oracle primary hits   == reference GLSL (tracer.fs intersectScene on SwiftShader)
  D3  oracle first-hit shading inputs / BRDF helpers ~= reference GLSL
  D5  oracle converged mean ~= reference GLSL converged mean (statistical)
Tolerances: SURVEY.md App. D; the texture-unit rows are wider because
SwiftShader's RGBA8 sampler is only ~1e-4 accurate (and the RGBE decode
multiplies its alpha error by 255 in the exponent: ~1 % on environment radiance).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle as O
from fspt_amd import scene as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_js(name):
    z = np.load(os.path.join(GOLD, f"js_scene_{name}.npz"))
    return z, json.loads(str(z["scene"])), json.loads(str(z["texts"])), json.loads(str(z["files"]))


def stand_in_images(z):
    """Decoded-image stand-ins with the heights the golden was generated with (the reference's getMaterial /
    TexturePacker only look at currentSrc and height; tools/make_goldens.py stand_in_images)."""
    rng = np.random.default_rng(5)
    return {u: rng.integers(0, 256, size=(h, h, 4), dtype=np.uint8) for u, h in sorted(json.loads(str(z["heights"])).items())}


def native_build(name):
    z, scene, texts, files = load_js(name)
    return z, S.build_scene_json(scene, texts, files, stand_in_images(z), env=z["env"], env_w=int(z["env_w"]),
                                 env_h=int(z["env_h"]), focus_rays=z["focus_rays"].tolist())


def scene_from_golden(name):
    """SceneArrays made of the REFERENCE JS pipeline's arrays (+ the atlas of the same scene)."""
    z, nat = native_build(name)
    return S.SceneArrays(bvh=z["bvh"].copy(), tri=z["tri"].copy(), mat=z["mat"].copy(), norm=z["norm"].copy(),
                         uv=z["uv"].copy(), atlas=nat.atlas, atlas_res=nat.atlas_res, atlas_layers=nat.atlas_layers,
                         env=z["env"].copy(), env_w=int(z["env_w"]), env_h=int(z["env_h"]), bins=z["bins"].copy(),
                         leaf_size=4, depth=int(z["depth"]))


@pytest.mark.parametrize("name", ["small", "variant", "mtl"])
def test_d0_native_pipeline_matches_reference_js(name):
    """obj_loader.js + mtl_loader.js + getMaterial + TexturePacker ids + mergeSceneProps + scene.normalize +
    bvh.js + the packing loops, run UNMODIFIED under Node (tools/js_ref), against the native pipeline: every
    packed array byte for byte, and the atlas layer list (colour / image, sRGB flag, swizzle) entry for entry.
    'mtl': usemtl groups with MTL materials, array-index group names, skips, worldTransforms, normalize,
    static + animated props; 'variant': mesh normals, negative indices, quads, MTL emission."""
    z, a = native_build(name)
    for k in ("bvh", "tri", "mat", "norm", "uv"):
        got, want = getattr(a, k), z[k]
        assert got.size == want.size, k
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{k} differs from the reference JS output"
    assert np.array_equal(a.bins, z["bins"])
    assert a.depth == int(z["depth"])
    assert a.meta["layers"] == json.loads(str(z["image_set"]))
    assert a.atlas_layers == len(a.meta["layers"])
    # shootAutoFocusRay (main.js:447-546), float64 on the host tree: lensFeatures[0] bit for bit
    assert np.array_equal(np.array(a.meta["focus"], np.float64).view(np.uint64), z["focus"].view(np.uint64))


def test_mtl_parser_quirks():
    """ParseMaterials (mtl_loader.js): keys are case-folded, falsy scalars are dropped (`if (value)`), vectors are
    kept, url set in first-seen order, lines before the first newmtl ignored."""
    mats, urls = S.parse_materials("Kd 1 1 1\nnewmtl  a\nKD 0.5 0.25 1\nNs 0\nior 1.5\nmap_Kd  t/x.png\nmap_Bump t/x.png\n"
                                   "newmtl b\nPmr 0 0 0\ndielectric abc\nmap_kem\n", "base")
    assert mats == {"a": {"kd": [0.5, 0.25, 1.0], "ior": 1.5, "map_kd": "t/x.png", "map_bump": "t/x.png"},
                    "b": {"pmr": [0.0, 0.0, 0.0]}}
    assert urls == ["base/t/x.png"]


def test_d0_70k_scene_digests():
    """The BASELINE bunny stand-in (69 316 triangles): sha256 of every packed array equals the
    reference JS pipeline's (arrays themselves are 17 MB, so only digests are committed)."""
    dig = json.load(open(os.path.join(GOLD, "js_scene_70k_digest.json")))
    a = S.bunny_scene(n=76)
    assert (a.n_nodes, a.n_tris, a.depth) == (dig["n_nodes"], dig["n_tris"], dig["depth"])
    exc = dig.get("libm_exceptions", {})
    # everything that steers traversal is exact; tangents may differ in the last bit where glibc's
    # atan2/asin and V8's fdlibm round differently (obj_loader.js:64-71) - at most a handful of floats
    assert set(exc) <= {"norm"}
    for k in ("bvh", "tri", "mat", "norm", "uv", "bins"):
        arr = np.ascontiguousarray(getattr(a, k)).copy()
        if k in exc:
            assert len(exc[k]["index"]) <= 8 and exc[k]["max_ulp"] <= 1
            arr.view(np.uint32)[exc[k]["index"]] = np.array(exc[k]["js_bits"], dtype=np.uint32)
        assert hashlib.sha256(arr.tobytes()).hexdigest() == dig[k], k


def test_d0_1M_scene_digests():
    """BASELINE configs[2]'s input (bench.py --config c3: 1 002 256 triangles, 650 161 nodes, depth 22): the reference's OWN
    obj_loader.js + bvh.js built it under Node here (tools/make_goldens.py js1m: 142 s, of which 65.6 s in `new BVH`; the
    native builder: 8 s) and the native pipeline's packed arrays hash to the same sha256 - bvh, tri, mat, uv exactly; norm
    after patching the recorded 102 of its 27 M floats (tangents / bitangents only, <= 3 ulp: glibc's atan2 / asin against
    V8's fdlibm in the spherical-UV fallback of obj_loader.js:64-71, carried through the tangent's cross products)."""
    dig = json.load(open(os.path.join(GOLD, "js_scene_1M_digest.json")))
    a = S.bunny_scene(n=289)
    assert (a.n_nodes, a.n_tris, a.depth) == (dig["n_nodes"], dig["n_tris"], dig["depth"]) == (650161, 1002256, 22)
    exc = dig.get("libm_exceptions", {})
    assert set(exc) <= {"norm"}
    for k in ("bvh", "tri", "mat", "norm", "uv"):
        arr = np.ascontiguousarray(getattr(a, k)).copy()
        if k in exc:
            idx = np.array(exc[k]["index"])
            assert idx.size <= 128 and exc[k]["max_ulp"] <= 3
            assert (((idx % 27) % 9) // 3 >= 1).all()  # normTex record = n, t, bt per vertex: never a normal
            ulp = np.abs(arr.view(np.int32)[idx].astype(np.int64) - np.array(exc[k]["js_bits"], dtype=np.uint32).view(np.int32).astype(np.int64))
            assert ulp.max() <= 3
            arr.view(np.uint32)[idx] = np.array(exc[k]["js_bits"], dtype=np.uint32)
        assert hashlib.sha256(arr.tobytes()).hexdigest() == dig[k], k


@pytest.fixture(scope="module")
def stages():
    return np.load(os.path.join(GOLD, "glsl_stages_small.npz"))


@pytest.fixture(scope="module")
def first_hits(stages):
    a = scene_from_golden("small")
    W, H = int(stages["W"]), int(stages["H"])
    acc = np.zeros((H, W, 4), np.float32)
    fh = O.trace(a, W, H, stages["rays_pos"], stages["rays_dir"], 0, 1.0, float(stages["env_theta"]), 4, acc,
                 first_hits=True)
    return a, fh.reshape(H, W), acc


def test_d2_primary_hits_match_glsl(stages):
    a = scene_from_golden("small")
    rays = np.concatenate([stages["rays_pos"][..., :3], stages["rays_dir"][..., :3]], -1).reshape(-1, 6)
    t, idx, _, _ = O.intersect(a, rays)
    gi, gt = stages["hit_index"].reshape(-1), stages["hit_t"].reshape(-1)
    assert (gi == idx).mean() >= 0.9999
    m = (gi == idx) & (gi >= 0)
    assert m.sum() > 1000
    assert np.max(np.abs(gt[m] - t[m]) / t[m]) <= 1e-6
    assert np.all(gt[gi < 0] == np.float32(100000.0))


@pytest.mark.parametrize("field,key,cols,tol", [
    ("origin", "shade0", slice(0, 3), 1e-5), ("bary", "shade1", slice(0, 3), 5e-5), ("uv", "shade2", slice(0, 2), 1e-5),
    ("bary_normal", "shade6", slice(0, 3), 1e-5),
    # texture-unit rows (SwiftShader RGBA8 sampler precision)
    ("mr", "shade2", slice(2, 4), 5e-4), ("diffuse", "shade3", slice(0, 3), 5e-4),
    ("tex_normal", "shade4", slice(0, 3), 1e-3), ("macro_normal", "shade5", slice(0, 3), 1e-3),
    ("emissive", "shade7", slice(0, 3), 5e-4)])
def test_d3_first_hit_shading_inputs(stages, first_hits, field, key, cols, tol):
    _, fh, _ = first_hits
    hit = (stages["hit_index"] >= 0) & (fh["index"] == stages["hit_index"])
    assert hit.sum() > 1000
    d = np.abs(stages[key][..., cols] - fh[field])[hit]
    assert d.max() <= tol, f"{field}: max abs diff {d.max()}"


def test_d3_material_scalars(stages, first_hits):
    a, fh, _ = first_hits
    hit = stages["hit_index"] >= 0
    idx = stages["hit_index"][hit]
    assert np.array_equal(stages["shade3"][..., 3][hit], a.mat.reshape(-1, 12)[idx, 9])    # ior
    assert np.array_equal(stages["shade4"][..., 3][hit], a.mat.reshape(-1, 12)[idx, 10])   # dielectric


def test_d3_env_lookup_on_miss_pixels(stages, first_hits):
    _, fh, acc = first_hits
    miss = (stages["hit_index"] < 0) & (fh["index"] < 0)
    assert miss.sum() > 100
    g, o = stages["shade0"][..., :3][miss], acc[..., :3][miss]
    assert (np.abs(g - o) / np.maximum(o, 1e-3)).max() <= 0.03  # 255 x SwiftShader's alpha error in the exponent


@pytest.mark.parametrize("which,tol", [(0, 2e-4), (1, 2e-4), (2, 2e-5), (3, 0.03)])
def test_d3_brdf_helpers(stages, which, tol):
    a = scene_from_golden("small")
    inp = np.concatenate([stages["brdf_A"], stages["brdf_B"]], -1).reshape(-1, 8).copy()
    if which == 3:
        inp[:, 3] = float(stages["env_theta"])
    o = O.brdf_probe(a, which, inp)
    g = stages[f"brdf{which}"].reshape(-1, 4)
    fin = np.isfinite(g).all(1) & np.isfinite(o).all(1)
    assert fin.mean() > 0.99
    rel = np.abs(g - o)[fin] / np.maximum(np.abs(o[fin]), 1e-3)
    assert rel.max() <= tol


def test_d1_camera_rays_statistics(stages):
    """D1 is informational (ray jitter depends on the GLSL implementation's sin()): origins must
    stay inside the aperture disc and directions within the AA footprint of the oracle's."""
    W, H = int(stages["W"]), int(stages["H"])
    cam = S.BUNNY_CAMERA
    lens = S.lens_features(cam["focal_depth"], cam["aperture"])
    pos, d = O.camera(W, H, cam["P"], cam["I"], cam["fov_scale"], lens, float(stages["cam_rand_base"]))
    gp, gd = stages["cam_pos"], stages["cam_dir"]
    assert np.abs(np.linalg.norm(gd[..., :3], axis=-1) - 1).max() < 1e-5
    assert np.linalg.norm(gp[..., :3] - np.float32(cam["P"]), axis=-1).max() <= cam["aperture"] * 1.0001
    assert np.linalg.norm(pos[..., :3] - np.float32(cam["P"]), axis=-1).max() <= cam["aperture"] * 1.0001
    # pixel footprint ~ fov/H; AA disc radius 1.414 px + DoF parallax
    assert np.abs(gd[..., :3] - d[..., :3]).max() < 0.06
    assert abs(gd[..., :3].mean() - d[..., :3].mean()) < 2e-3


def test_d1_camera_rays_with_replayed_rnd_match_glsl():
    """camera.fs pinned deterministically: its main() run on SwiftShader with the four values its rnd() returned per pixel
    recorded next to the ray (tools/make_goldens.py camera_replay patches the reference file when the goldens are made);
    the oracle replays them (oracle_camera_probe) - the GLSL's sin() of a five-digit seed, which makes the two rnd() streams
    differ (the statistics test above: rays up to 0.04-0.09 apart), drops out, and everything behind it - getScreen, getAA,
    getDOF, the lens, the normalisation - agrees to float32 rounding: origins within 2 ulp, directions within 4 ulp of 1.
    Four cameras: the bench's, the same through configs[4]'s aperture 0.1, the variant scene's, one looking straight down."""
    z = np.load(os.path.join(GOLD, "glsl_camera_replay.npz"))
    W, H = int(z["W"]), int(z["H"])
    names = json.loads(str(z["names"]))
    assert len(names) == 4
    for n in names:
        P, I, lens = ([float(x) for x in z[f"{n}_{k}"]] for k in ("P", "I", "lens"))
        rec = z[f"{n}_rec"]
        assert rec.shape == (H, W, 4) and 0.45 < rec.mean() < 0.55 and rec.std() > 0.25  # four uniform numbers per pixel
        pos, d = O.camera_probe(W, H, P, I, float(z[f"{n}_fov"]), lens, rec)
        assert np.abs(pos[..., :3] - z[f"{n}_pos"]).max() <= 2 * np.spacing(np.float32(np.abs(P).max())), n
        assert np.abs(d[..., :3] - z[f"{n}_dir"]).max() <= 4 * np.spacing(np.float32(0.5)) * 2, n
        # the lens really moves the origin: on the aperture's disc around P, and the replay matters
        r = np.linalg.norm(pos[..., :3] - np.float32(P), axis=-1)
        assert r.max() <= lens[1] * 1.0001 and r.max() > 0.9 * lens[1]
        own, _ = O.camera(W, H, P, I, float(z[f"{n}_fov"]), lens, float(z[f"{n}_rand_base"]))
        assert np.abs(own[..., :3] - z[f"{n}_pos"]).max() > 10 * np.spacing(np.float32(3.0))


def _env_decode_bias(a, env_theta):
    """SwiftShader's RGBE decode relative to the exact one, on THIS environment map: its RGBA8 -> float conversion is
    ~6e-5 off, which envColor multiplies by 255 in the exponent (tracer.fs:412) - a deterministic factor (0.989 on the
    test maps) measured by the reference's own envSample on a fixed set of directions ('brdf3' of the stage goldens) and
    here against the oracle's lookup of the same directions.  Every photon of these scenes comes from the environment, so
    the GLSL's converged mean carries exactly this factor."""
    for name in ("small", "variant"):
        st = np.load(os.path.join(GOLD, f"glsl_stages_{name}.npz"))
        sc = scene_from_golden(name)
        if sc.env_w == a.env_w and sc.env_h == a.env_h and np.array_equal(sc.env, a.env):
            inp = np.concatenate([st["brdf_A"], st["brdf_B"]], -1).reshape(-1, 8).copy()
            inp[:, 3] = float(st["env_theta"])
            o = O.brdf_probe(sc, 3, inp)[:, :3].astype(np.float64)
            g = st["brdf3"].reshape(-1, 4)[:, :3].astype(np.float64)
            fin = np.isfinite(g).all(1) & np.isfinite(o).all(1)
            return float(g[fin].sum() / o[fin].sum())
    raise AssertionError("no stage golden with this environment map")


@pytest.mark.parametrize("name", ["small", "small_d4", "variant", "textured", "small_dof", "small_d1"])
def test_d5_converged_mean_matches_glsl(name):
    """Stage D5: 16 384 spp of the UNMODIFIED tracer.fs on SwiftShader (two randBase streams) against 16 384 spp of the
    oracle (its own stream): depth 8 (the BASELINE depth) on the flat-colour, the refractive / emissive and the
    image-mapped scene, depth 4 (tracer.fs:9 as shipped) on the first, and the first again through BASELINE configs[4]'s
    lens (aperture 0.1: camera.fs getDOF moves every sample's origin over a 5x wider disc), and depth 1 on the first
    scene (direct light only: one shading event's NEE sample and MIS-weighted extension ray, nothing averaged
    over later bounces).  After dividing out the deterministic RGBE-decode
    factor of SwiftShader (see _env_decode_bias) the whole-image mean agrees within 0.5 %, and the per-pixel difference is
    the Monte-Carlo noise of the two renders: rel-L2 within 1.3x the GLSL-vs-GLSL floor (0.015 / 0.026 / 0.02)."""
    z = np.load(os.path.join(GOLD, f"glsl_converged_{name}.npz"))
    scene_name = str(z["scene"])
    a = S.textured_test_scene() if scene_name == "textured" else scene_from_golden(scene_name)
    W, H, spp, bounces = int(z["W"]), int(z["H"]), int(z["spp"]), int(z["bounces"])
    assert spp >= 16384 and bounces == {"small_d4": 4, "small_d1": 1}.get(name, 8)
    ga, gb = z["a"][..., :3].astype(np.float64), z["b"][..., :3].astype(np.float64)

    def rel_l2(x, y):
        return float(np.linalg.norm(x - y) / np.linalg.norm(y))
    floor = rel_l2(ga, gb)  # GLSL-vs-GLSL, two randBase streams
    assert floor <= 0.03
    acc = np.zeros((H, W, 4), np.float32)
    O.render(a, W, H, [float(x) for x in z["P"]], [float(x) for x in z["I"]], float(z["fov_scale"]),
             [float(x) for x in z["lens"]], float(z["env_theta"]), bounces, 0, spp, 99, acc)
    bias = _env_decode_bias(a, float(z["env_theta"]))
    assert 0.985 <= bias <= 0.993, bias
    o = acc[..., :3].astype(np.float64)
    gm = 0.5 * (ga + gb)
    r = gm.mean() / o.mean()
    if scene_name == "small" and name != "small_d1":
        assert abs(r / bias - 1.0) <= 0.005, (r, bias)   # every photon comes from the environment
    else:
        # emitted light (tracer.fs:467: an MTL Kem / an emissive map) does not pass through the RGBE decode: the
        # GLSL's mean lies between the fully biased and the unbiased one.  (Depth 1: the decode factor depends on the
        # exponent BYTE - 0.986-0.992 over the probe directions, which see sky; 1.009 on a map of one other exponent
        #  - and at depth 1 half the light is the NEE's samples of the sun's texels, whose exponent the
        # probe set does not weigh: 0.994 measured against the probe average 0.989.)
        assert bias - 0.005 <= r <= 1.005, (r, bias)
    o = o * r
    assert rel_l2(o, ga) <= 1.3 * floor and rel_l2(o, gb) <= 1.3 * floor, (rel_l2(o, ga), rel_l2(o, gb), floor)


def test_d0_env_bins_odd_image():
    z = np.load(os.path.join(GOLD, "js_env_bins_odd.npz"))
    assert np.array_equal(S.env_bins(z["env"], int(z["env_w"]), int(z["env_h"])), z["bins"])
    assert z["bins"].size // 4 > 500  # the NaN path of biSplit really was exercised


@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_draw_fs_matches_glsl(i):
    """draw.fs (tonemap; variants 2,3 with the 5x5 firefly filter) run on SwiftShader vs oracle_draw.
    pow() precision differs by at most one 8-bit step.  Border pixels of the filtered variants are
    excluded: out-of-range texelFetch is undefined in GLES 3.0 (SwiftShader clamps) while WebGL 2 - the
    reference's platform - returns zero, which is what the oracle and the HIP kernel implement."""
    z = np.load(os.path.join(GOLD, "glsl_draw.npz"))
    e, s, d, g = [float(v) for v in z[f"params{i}"]]
    o = O.draw(z["hdr"], e, s, bool(d), g).astype(int)
    want = z[f"rgba{i}"].astype(int)
    if d:
        o, want = o[2:-2, 2:-2], want[2:-2, 2:-2]
    diff = np.abs(o - want)
    assert diff.max() <= 1
    assert (diff == 0).mean() >= 0.99
    assert (o[..., 3] == 255).all()


@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_atlas_writer_matches_reference_shader(i):
    """resample_image vs the reference's WebGLTextureWriter shader (texture_packer.js:103-121) on SwiftShader:
    bilinear resample, y-flip, swizzle, sRGB decode, premultiply -> within one 8-bit step."""
    z = np.load(os.path.join(GOLD, "glsl_atlas_writer.npz"))
    res, cor, *swz = [int(v) for v in z[f"params{i}"]]
    o = S.resample_image(z[f"src{i}"], res, bool(cor), tuple(swz)).astype(int)
    d = np.abs(o - z[f"dst{i}"].astype(int))
    assert d.max() <= 1 and (d == 0).mean() >= 0.9
    assert (o[..., 3] == 255).all()


def test_d3_textured_scene_bilinear_atlas():
    """Scene with 16x16 image maps: GLSL texture() (bilinear, REPEAT) vs the oracle's hand filter."""
    z = np.load(os.path.join(GOLD, "glsl_stages_textured.npz"))
    a = S.textured_test_scene()
    assert (a.atlas_res, a.atlas_layers) == (int(z["atlas_res"]), int(z["atlas_layers"])) and a.atlas_res == 16
    W, H = int(z["W"]), int(z["H"])
    acc = np.zeros((H, W, 4), np.float32)
    fh = O.trace(a, W, H, z["rays_pos"], z["rays_dir"], 0, 1.0, float(z["env_theta"]), 4, acc, first_hits=True).reshape(H, W)
    assert (fh["index"] == z["hit_index"]).mean() >= 0.9999
    hit = (z["hit_index"] >= 0) & (fh["index"] == z["hit_index"])
    assert hit.sum() > 1000
    for field, key, cols, tol in (("uv", "shade2", slice(0, 2), 1e-5), ("mr", "shade2", slice(2, 4), 1e-3),
                                  ("diffuse", "shade3", slice(0, 3), 1e-3), ("tex_normal", "shade4", slice(0, 3), 1e-3),
                                  ("macro_normal", "shade5", slice(0, 3), 1e-3), ("emissive", "shade7", slice(0, 3), 1e-3)):
        d = np.abs(z[key][..., cols] - fh[field])[hit]
        assert d.max() <= tol, (field, d.max())
    # the maps really vary across the image (not a flat-colour scene in disguise)
    assert fh["diffuse"][hit].std() > 0.1 and fh["tex_normal"][hit][:, 0].std() > 0.05


def write_asset_tree(root, z, scene, texts, files, frame_scenes=None):
    """Lay a golden scene out on disk the way the reference's web root holds it."""
    from PIL import Image
    for rel, text in list(texts.items()) + list(files.items()):
        os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
        with open(os.path.join(root, rel), "w") as fh:
            fh.write(text)
    for rel, img in stand_in_images(z).items():
        os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
        Image.fromarray(img).save(os.path.join(root, rel))  # PNG keeps straight-alpha RGBA bytes
    os.makedirs(os.path.join(root, "environment"), exist_ok=True)
    Image.fromarray(z["env"].reshape(int(z["env_h"]), int(z["env_w"]), 4)).save(os.path.join(root, "environment", "sky.RGBE.PNG"))
    os.makedirs(os.path.join(root, "scene"), exist_ok=True)
    for name, sc in (frame_scenes or {"test.json": scene}).items():
        sc = dict(sc, environment="environment/sky.RGBE.PNG")
        with open(os.path.join(root, "scene", name), "w") as fh:
            json.dump(sc, fh)


def test_scene_file_loader_matches_reference_arrays(tmp_path):
    """load_scene_file on an on-disk web root (JSON + OBJ + MTL + PNG maps + RGBE sky) gives the arrays the
    reference's JS pipeline gives for the same scene, and the reference's camera defaults (main.js:50-75)."""
    from fspt_amd import scene_file as F
    z, scene, texts, files = load_js("mtl")
    write_asset_tree(str(tmp_path), z, scene, texts, files)
    a, st = F.load_scene_file(os.path.join(str(tmp_path), "scene", "test.json"))
    for k in ("bvh", "tri", "mat", "norm", "uv"):
        assert np.array_equal(getattr(a, k).view(np.uint32), z[k].view(np.uint32)), k
    assert a.meta["layers"] == json.loads(str(z["image_set"]))
    assert np.array_equal(a.env, z["env"]) and np.array_equal(a.bins, z["bins"])
    assert st["eye"] == [0, 0, 2] and st["dir"] == [0, 0, -1] and st["fov_scale"] == 0.5 and st["samples"] == 2000
    assert st["focus"] == float(z["focus"][1])  # FOCUS_RAYS[1] is the default camera
    # the decoded maps really reach the atlas: layer 1 is mr.png through its last-assigned swizzle
    img = stand_in_images(z)["models/tex/mr.png"]
    want = S.resample_image(img, a.atlas_res, False, (2, 1, 0, 3))
    assert np.array_equal(a.atlas.reshape(a.atlas_layers, a.atlas_res, a.atlas_res, 4)[1], want)


@pytest.mark.parametrize("name,W,H", [("small", 64, 40), ("mtl", 48, 32)])
def test_bvh_test_mode_counts_equal_reference_glsl(name, W, H):
    """The reference's `mode=test` shader (bvh_test.fs, unmodified apart from the quad-replication fetch
    substitution) run on SwiftShader: per-pixel traversal-loop iteration counts x 0.001, two ticks of running
    mean.  Integer work counters -> exact: the oracle's traversal visits exactly the reference's node sequence
    lengths for every camera ray, and the accumulate arithmetic is bit-equal."""
    z = np.load(os.path.join(GOLD, "glsl_bvh_test.npz"))
    a = scene_from_golden(name)
    acc = np.zeros((H, W, 4), np.float32)
    for tick in range(2):
        O.trace_test(a, W, H, z[f"{name}_pos{tick}"], z[f"{name}_dir{tick}"], tick, acc)
        want = z[f"{name}_img{tick}"]
        assert np.array_equal(acc.view(np.uint32), want.view(np.uint32)), f"tick {tick}"
    # and they are the step counters fspt_intersect / oracle_intersect report
    rays = np.concatenate([z[f"{name}_pos1"][..., :3], z[f"{name}_dir1"][..., :3]], -1).reshape(-1, 6)
    _, _, steps, _ = O.intersect(a, rays)
    c0 = np.rint(z[f"{name}_img0"][..., 0] / np.float32(0.001))
    c1 = np.rint((z[f"{name}_img1"][..., 0].astype(np.float64) * 2 - z[f"{name}_img0"][..., 0]) / 0.001)
    assert np.array_equal(steps.reshape(H, W), c1) and c0.max() > 10


@pytest.mark.parametrize("i", [0, 1])
def test_draw_scale_uniform_matches_reference_glsl(i):
    """draw.fs with scale = 0.25 (what the reference draws while the camera moves): every output pixel fetches
    texel ivec2(gl_FragCoord * scale) of the accumulator.  Same tolerance as the unscaled cases; filtered
    variants exclude the pixels whose 5x5 window leaves the image (undefined fetch in GLES 3.0)."""
    z = np.load(os.path.join(GOLD, "glsl_draw.npz"))
    e, s, d, g, sc = [float(v) for v in z[f"scaled_params{i}"]]
    o = O.draw(z["hdr"], e, s, bool(d), g, sc).astype(int)
    want = z[f"scaled_rgba{i}"].astype(int)
    if d:  # texel (x*sc, y*sc) must be >= 2 from the border: drop the first 8+ output rows/columns
        o, want = o[12:, 12:], want[12:, 12:]
    diff = np.abs(o - want)
    assert diff.max() <= 1 and (diff == 0).mean() >= 0.99
    # a 4x4 block of output pixels shows one texel
    full = O.draw(z["hdr"], e, s, False, g, 1.0)
    quarter = O.draw(z["hdr"], e, s, False, g, 0.25)
    H, W = full.shape[:2]
    assert np.array_equal(quarter[:H - H % 4:4, :W - W % 4:4], full[:H // 4, :W // 4])


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 5, 8])
def test_bvh_test_counts_on_random_soups_equal_reference_glsl(seed):
    """bvh_test.fs (the reference's own traversal, LEAF_SIZE 1 / 2 / 4 / 5) on SwiftShader over random triangle
    soups with degenerate, duplicated, sliver and huge triangles: the oracle's per-pixel iteration counts are
    exactly the GLSL's on all 960 camera rays of every scene."""
    z = np.load(os.path.join(GOLD, "glsl_fuzz_bvh_test.npz"))
    tri = z[f"s{seed}_tri"]
    nt = tri.size // 9
    a = S.SceneArrays(bvh=z[f"s{seed}_bvh"], tri=tri, mat=np.zeros(nt * 12, np.float32), norm=np.zeros(nt * 27, np.float32),
                      uv=np.zeros(nt * 6, np.float32), atlas=np.zeros(4, np.uint8), atlas_res=1, atlas_layers=1, env=None,
                      env_w=0, env_h=0, bins=np.array([0, 0, 1, 2048], np.uint32), leaf_size=int(z[f"s{seed}_leaf"]))
    img = z[f"s{seed}_img"]
    H, W = img.shape[:2]
    acc = np.zeros((H, W, 4), np.float32)
    O.trace_test(a, W, H, z[f"s{seed}_pos"], z[f"s{seed}_dir"], 0, acc)
    assert np.array_equal(acc.view(np.uint32), img.view(np.uint32))
    assert img[..., 0].max() > 0.015
    # D2 on the same rays through tracer.fs's own intersectScene: the same triangle everywhere - incl. the equal-t
    # ties of duplicated triangles, where the first one visited wins - and t to float32 rounding (sliver triangles
    # amplify the GLSL compiler's freedom to contract a*b+c: 2.5e-5 on one scene, <= 8e-7 on the others)
    rays = np.concatenate([z[f"s{seed}_pos"][..., :3], z[f"s{seed}_dir"][..., :3]], -1).reshape(-1, 6)
    t, idx, _, _ = O.intersect(a, rays)
    assert np.array_equal(idx, z[f"s{seed}_hit_index"].reshape(-1))
    h = idx >= 0
    assert h.sum() > 100
    assert (np.abs(z[f"s{seed}_hit_t"].reshape(-1)[h] - t[h]) / np.maximum(t[h], 1e-6)).max() < 1e-4


def test_d2_d3_variant_scene_matches_glsl():
    """The same D2 / D3 probes on the 'variant' scene (mesh normals, quads, negative indices, a dielectric and an
    emissive MTL material, metallic-roughness colours): hit index and t, hit point, barycentrics, uv, interpolated and
    normal-mapped normals, the four material maps, ior and dielectric per hit triangle."""
    st = np.load(os.path.join(GOLD, "glsl_stages_variant.npz"))
    a = scene_from_golden("variant")
    W, H = int(st["W"]), int(st["H"])
    rays = np.concatenate([st["rays_pos"][..., :3], st["rays_dir"][..., :3]], -1).reshape(-1, 6)
    t, idx, _, _ = O.intersect(a, rays)
    gi, gt = st["hit_index"].reshape(-1), st["hit_t"].reshape(-1)
    assert np.array_equal(gi, idx)
    m = gi >= 0
    assert m.sum() > 1000 and np.max(np.abs(gt[m] - t[m]) / t[m]) <= 1e-6
    acc = np.zeros((H, W, 4), np.float32)
    fh = O.trace(a, W, H, st["rays_pos"], st["rays_dir"], 0, 1.0, float(st["env_theta"]), 4, acc, first_hits=True).reshape(H, W)
    hit = st["hit_index"] >= 0
    for field, key, cols, tol in [("origin", "shade0", slice(0, 3), 1e-5), ("bary", "shade1", slice(0, 3), 5e-5),
                                  ("uv", "shade2", slice(0, 2), 1e-5), ("bary_normal", "shade6", slice(0, 3), 1e-5),
                                  ("mr", "shade2", slice(2, 4), 5e-4), ("diffuse", "shade3", slice(0, 3), 5e-4),
                                  ("tex_normal", "shade4", slice(0, 3), 1e-3), ("macro_normal", "shade5", slice(0, 3), 1e-3),
                                  ("emissive", "shade7", slice(0, 3), 5e-4)]:
        d = np.abs(st[key][..., cols] - fh[field])[hit]
        assert d.max() <= tol, f"{field}: max abs diff {d.max()}"
    ti = st["hit_index"][hit]
    assert np.array_equal(st["shade3"][..., 3][hit], a.mat.reshape(-1, 12)[ti, 9])    # ior
    assert np.array_equal(st["shade4"][..., 3][hit], a.mat.reshape(-1, 12)[ti, 10])   # dielectric
    assert (a.mat.reshape(-1, 12)[ti, 10] > 0).any() and (st["shade7"][..., :3][hit] > 0).any()  # refractive, emissive hit


# ---- the stochastic half of tracer.fs, pinned deterministically ------------------------------------------------
# tools/make_goldens.py `samplers`: the reference's UNMODIFIED sampleMicrofacet / sampleLambert / sampleEnv and one
# iteration of main()'s bounce loop (tracer.fs:447-499, cut out of the reference at generation time) run on SwiftShader,
# with the values its rnd() returned recorded next to the outputs.  The oracle replays those values
# (oracle_sampler_probe / oracle_bounce_probe), so SwiftShader's sin() - 1e-5 off for seeds of a few thousand, which
# turns the sin-hash into an unrelated stream - drops out and everything downstream of rnd() is compared to float32
# rounding.  The bounce probe also replays the four texture() results (the GLSL sampler's 8-bit-ish filter precision is
# pinned by stage D3 above); `test_bounce_body_with_own_texture_fetch` keeps the oracle's own fetch in the loop.
#
# Tolerances: directions 1e-5 abs, pdfs / throughputs / MIS weights 1e-4 rel - plus, where tracer.fs itself evaluates
# an ill-conditioned expression, the float32 rounding of that expression (GLSL leaves both the rounding of every
# operation and the contraction of a*b+c to the implementation; the oracle's choice is DESIGN.md section 2):
#   cosTheta = sqrt((1 - r2) / (1 + (a^2 - 1) r2)), sinTheta = sqrt(1 - cosTheta^2)  (tracer.fs:263-264): the
#       denominator cancels to ~(1 - r2) and 1 - cosTheta^2 cancels again -> error 2^-23 / sinTheta * (1 + 1 / denominator)
#       in the half vector (a roughness-0.001 lobe is 1e-3 wide and float32 resolves 1 - cos^2 to 6e-8)
#   gtr2: t = 1 + (a^2 - 1) ndh^2    (tracer.fs:217)  relative error 2 * 2^-22 / t in D, t down to a^2 = 1e-6
SAMPLER_SCENES = ["small", "variant", "textured"]
EPS23 = 2.0 ** -23


def _sampler_golden(name):
    z = np.load(os.path.join(GOLD, f"glsl_samplers_{name}.npz"))
    a = S.textured_test_scene() if name == "textured" else scene_from_golden(name)
    return z, a


def _same_nonfinite(o, g):
    """NaN where the GLSL has NaN, the same infinity where it has one."""
    return np.array_equal(np.isnan(o), np.isnan(g)) and np.array_equal(np.where(np.isinf(o), o, 0), np.where(np.isinf(g), g, 0))


def _half_vector_slack(rough, r2, sin_theta):
    a = np.maximum(0.001, np.asarray(rough, np.float64))
    den = 1.0 + (a * a - 1.0) * np.asarray(r2, np.float64)
    return EPS23 / np.maximum(sin_theta, 1e-6) * (1.0 + 1.0 / np.maximum(den, 1e-7))


def _gtr2_t(a, ndh):
    a = np.maximum(0.001, a.astype(np.float64))
    return 1.0 + (a * a - 1.0) * ndh.astype(np.float64) ** 2


def test_rnd_replay_basis():
    """What the replay rests on: the GLSL advances `seed` exactly like the oracle (binary32 addition), its rnd() values
    are in [0, 1) - and they are NOT the oracle's (nor anybody's: SwiftShader's sin is ~1e-5 off at |x| ~ 1e4, and
    43758.5453 * 1e-5 is half a period), while the oracle's sin-hash is within 0.006 of the exact function."""
    z, _ = _sampler_golden("small")
    B = z["samp_B"].reshape(-1, 4)
    rec = np.concatenate([z["samp_rnd_a"], z["samp_rnd_b"]], -1).reshape(-1, 8)
    assert rec.min() >= 0.0 and rec.max() < 1.0 and 0.45 < rec.mean() < 0.55
    step = np.float32(0.211324865405187)
    s2 = ((B[:, 0] + step).astype(np.float32) + step).astype(np.float32)
    assert np.array_equal(z["samp_microfacet"].reshape(-1, 4)[:, 3].view(np.uint32), s2.view(np.uint32))
    assert np.array_equal(z["samp_lambert"].reshape(-1, 4)[:, 3].view(np.uint32), s2.view(np.uint32))
    own = O.rnd_sequence(B[:, 0], 8)
    sd, truth = B[:, 0].copy(), np.zeros(own.shape)
    for k in range(8):
        sd = (sd + step).astype(np.float32)
        v = np.sin(sd.astype(np.float64)) * 43758.5453123
        truth[:, k] = v - np.floor(v)

    def circ(x, y):
        d = np.abs(x - y)
        return np.minimum(d, 1 - d)
    assert circ(own, truth).max() < 0.008
    small = np.abs(B[:, 0]) < 40
    assert circ(rec, truth)[small].max() < 0.2 and circ(rec, truth)[~small].max() > 0.4


@pytest.mark.parametrize("name", SAMPLER_SCENES)
def test_sample_microfacet_and_lambert_match_glsl(name):
    """tracer.fs:256-280 on 2 560 (normal, roughness, seed) triples incl. the |n.z| >= 0.999 frame and roughness below
    the 0.001 clamp."""
    z, a = _sampler_golden(name)
    A = z["samp_A"].reshape(-1, 4)
    rec = np.concatenate([z["samp_rnd_a"], z["samp_rnd_b"]], -1).reshape(-1, 8)
    assert (np.abs(A[:, 2]) >= 0.999).sum() > 300 and (A[:, 3] < 0.001).sum() > 100
    o = O.sampler_probe(a, 0, A, rec)
    g = z["samp_microfacet"].reshape(-1, 4)
    assert (o[:, 3] == 2).all()
    d = np.abs(o[:, :3] - g[:, :3]).max(1)
    sin_theta = np.linalg.norm(np.cross(o[:, :3].astype(np.float64), A[:, :3].astype(np.float64)), axis=1)
    assert (d <= 1e-5 + _half_vector_slack(A[:, 3], rec[:, 1], sin_theta)).all(), d.max()
    well = (sin_theta > 0.05) & (rec[:, 1] < 0.9)
    assert well.mean() > 0.6 and d[well].max() <= 1e-5
    o = O.sampler_probe(a, 1, A, rec)
    g = z["samp_lambert"].reshape(-1, 4)
    assert (o[:, 3] == 2).all()
    assert np.abs(o[:, :3] - g[:, :3]).max() <= 1e-5
    assert np.abs(np.linalg.norm(o[:, :3], axis=1) - 1).max() < 1e-5
    assert ((o[:, :3] * A[:, :3]).sum(1) > -1e-6).all()  # in the normal's hemisphere


@pytest.mark.parametrize("name", SAMPLER_SCENES)
def test_sample_env_matches_glsl(name):
    """tracer.fs:421-434: bin choice, point in the bin, direction and pdf; bins the Uint16 truncation left empty
    (env_sampler.js:73) give pdf = inf on both sides."""
    z, a = _sampler_golden(name)
    rec = np.concatenate([z["samp_rnd_a"], z["samp_rnd_b"]], -1).reshape(-1, 8)
    o = O.sampler_probe(a, 2, z["samp_A"].reshape(-1, 4), rec, float(z["env_theta"]))
    g = z["samp_env"].reshape(-1, 4)
    assert np.abs(o[:, :3] - g[:, :3]).max() <= 1e-5
    assert _same_nonfinite(o[:, 3], g[:, 3])
    fin = np.isfinite(g[:, 3])
    assert fin.mean() > 0.9
    sin_phi = np.sqrt(np.maximum(1.0 - g[fin, 1].astype(np.float64) ** 2, 1e-12))
    rel = np.abs(o[fin, 3] - g[fin, 3]) / np.abs(g[fin, 3])
    assert (rel <= 1e-5 + 4 * EPS23 / sin_phi).all() and rel.max() <= 1e-4, rel.max()
    # every bin of the table was drawn from
    n_bins = a.bins.size // 4
    assert len(set(np.minimum((n_bins * rec[:, 0]).astype(int), n_bins - 1))) == n_bins


def _bounce(name, with_tex=True):
    z, a = _sampler_golden(name)
    rays = np.concatenate([z["rays_pos"][..., :3], z["rays_dir"][..., :3]], -1).reshape(-1, 6)
    t, idx = z["hit_t"].reshape(-1), z["hit_index"].reshape(-1)
    G = {k: z[f"bounce{k}"].reshape(-1, 4) for k in list(range(400, 414)) + [420, 421]}
    rec = np.concatenate([G[401], G[402]], -1)
    tex = np.zeros((rays.shape[0], 12), np.float32)
    tex[:, 0:3], tex[:, 3:6], tex[:, 6], tex[:, 7], tex[:, 8:11] = G[411][:, :3], G[412][:, :3], G[411][:, 3], G[412][:, 3], G[413][:, :3]
    o = O.bounce_probe(a, rays, t, idx, float(z["rand_base"]), float(z["env_theta"]), rec, tex if with_tex else None)
    return z, a, rays, t, idx, G, o


@pytest.mark.parametrize("name", SAMPLER_SCENES)
def test_bounce_body_matches_glsl(name):
    """One iteration of tracer.fs main()'s loop (447-499) on 2 560 rays: camera rays and rays that hit a triangle from
    behind.  Branch decisions equal on every ray; ray.origin / ray.dir, bsdfThroughput, envThroughput, the pdfs, the MIS
    weights, the micro / macro normals and the emitted colour to float32 rounding."""
    z, a, rays, t, idx, G, o = _bounce(name)
    _, io, _, _ = O.intersect(a, rays)
    assert np.array_equal(io, idx)                      # the hit the body starts from is the oracle's own, too
    h = idx >= 0
    assert h.sum() > 2000
    inside, specular, refracted = G[400][:, 1] > 0, G[400][:, 2] > 0, G[407][:, 3] < 0
    assert np.array_equal(o[h, 1] > 0, inside[h]) and np.array_equal(o[h, 2] > 0, specular[h])
    assert np.array_equal(o[h, 23] > 0, refracted[h]) and np.array_equal(o[h, 35], G[410][h, 3])
    lambert = h & ~specular & ~refracted
    for m in (inside, ~inside, specular, lambert):
        assert (m & h).sum() > 200
    if name == "variant":   # the dielectric: refraction out of and into the solid, total internal reflection
        ior = a.mat.reshape(-1, 12)[idx[h], 9]
        assert (refracted & inside & h).sum() > 10 and (refracted & ~inside & h).sum() > 100
        cos_i = np.abs((o[h, 28:31] * rays[h, 3:6]).sum(1))
        tir = inside[h] & (G[410][h, 3] >= 0) & (ior * ior * (1 - cos_i * cos_i) > 1.0001)
        assert tir.sum() > 20 and specular[h][tir].all()
    if name != "small":
        assert (G[408][h, :3] > 0).any()                # emissive hits (tracer.fs:467)
    # the rnd() calls consumed: 6 on the specular / refraction branches, 8 with sampleLambert
    assert np.array_equal(o[h, 27], np.where(lambert[h], 8, 6))
    step = np.float32(0.211324865405187)
    sd = G[400][h, 0].copy()
    for _ in range(8):
        sd = np.where(np.arange(8)[_] < o[h, 27], (sd + step).astype(np.float32), sd)
    assert np.array_equal(sd.view(np.uint32), G[408][h, 3].view(np.uint32))          # `seed` after the body
    assert (np.abs(o[h, 0] - G[400][h, 0]) <= 3e-4 * np.maximum(np.abs(G[400][h, 0]), 1.0)).all()  # tracer.fs:458

    def close(cols, key, gcols, tol_abs=0.0, tol_rel=0.0, extra=None, mask=h):
        x, y = o[mask][:, cols].astype(np.float64), G[key][mask][:, gcols].astype(np.float64)
        assert _same_nonfinite(x, y), key
        f = np.isfinite(y)
        lim = tol_abs + tol_rel * np.abs(y)
        if extra is not None:
            lim = lim + (extra[mask][:, None] if y.ndim > 1 else extra[mask])
        with np.errstate(invalid="ignore"):
            bad = f & ~(np.abs(np.where(f, x - y, 0)) <= lim)
        w = np.argwhere(bad)
        assert not bad.any(), (key, int(bad.sum()), np.flatnonzero(mask)[w[0][0]], x[tuple(w[0])], y[tuple(w[0])], lim[tuple(w[0])])
    # geometry
    close(slice(32, 35), 410, slice(0, 3), 1e-6)                                     # macroNormal
    close(slice(8, 11), 404, slice(0, 3), 2e-6)                                      # ray.origin
    close(slice(20, 23), 407, slice(0, 3), 1e-5)                                     # envDirPdf.xyz
    close(15, 405, 3, 1e-5)                                                          # cosEnv
    close(19, 406, 3, 0.0, 1e-4)                                                     # envDirPdf.a
    close(slice(24, 27), 408, slice(0, 3), 1e-6, 1e-6)                               # emitted colour
    mac, mic = o[:, 32:35].astype(np.float64), o[:, 28:31].astype(np.float64)
    rough = _tex_of(G)[:, 7].astype(np.float64) ** 2
    slack = _half_vector_slack(rough, G[401][:, 1], np.linalg.norm(np.cross(mic, mac), axis=1))
    close(slice(28, 31), 409, slice(0, 3), 1e-5, extra=slack)                        # microNormal
    close(slice(4, 7), 403, slice(0, 3), 1e-5, extra=np.where(specular | refracted, 2 * slack, 0.0))  # ray.dir
    # pdfs, throughputs, weights: 1e-4 rel + the rounding of gtr2's t on the specular branch
    inc = -rays[:, 3:6].astype(np.float64)
    hv = o[:, 4:7] + inc
    hv_len = np.maximum(np.linalg.norm(hv, axis=1), 1e-6)   # normalize(bsdfDir + incident) at grazing incidence (tracer.fs:230)
    hv /= np.maximum(np.linalg.norm(hv, axis=1, keepdims=True), 1e-30)
    t_b = np.abs(_gtr2_t(rough, np.abs((hv * mac).sum(1))))
    he = o[:, 20:23] + inc
    he_len = np.maximum(np.linalg.norm(he, axis=1), 1e-6)
    he /= np.maximum(np.linalg.norm(he, axis=1, keepdims=True), 1e-30)
    t_e = np.abs(_gtr2_t(rough, np.abs((he * mac).sum(1))))
    a_ = np.maximum(0.001, rough)
    den = np.maximum(1.0 + (a_ * a_ - 1.0) * G[401][:, 1].astype(np.float64), 1e-7)
    cond_b = np.where(specular, 4 * EPS23 * (2.0 + 1.0 / den + 4.0 / hv_len) / np.maximum(t_b, 1e-7), 0.0)  # the half vector's own error moves ndh
    cond_e = np.where(specular, 4 * EPS23 * (1.0 + 4.0 / he_len) / np.maximum(t_e, 1e-7), 0.0)
    # ... and its denominator 4 |bsdfDir . halfVec| (tracer.fs:233) = 2 |bsdfDir + incident| with a half vector that is
    # only known to 4 eps / |bsdfDir + incident| when the reflection grazes the microfacet
    cond_b = cond_b + np.where(specular, 8 * EPS23 / (hv_len * hv_len), 0.0)
    gp = np.abs(G[400][:, 3].astype(np.float64))
    # Lambert: dir.z = sqrt(1 - x^2 - y^2) (tracer.fs:211) cancels at the rim of the disc
    cl_ = np.maximum(np.abs((o[:, 4:7] * mac).sum(1)), 1e-6)
    rim = np.where(lambert, 4 * EPS23 / cl_, 0.0)
    close(3, 400, 3, 1e-6, 1e-4, extra=cond_b * gp + rim)                                               # bsdfPdf
    # w = a^2 / (a^2 + b^2): dw <= (rel. error of a + rel. error of b) / 2
    close(7, 403, 3, 5e-5, extra=0.5 * cond_b)                                                          # weights.x
    close(11, 404, 3, 5e-5, extra=0.5 * cond_b)                                                         # weights.y
    cl = np.maximum(np.abs((o[:, 4:7] * mac).sum(1)), 1e-7)
    gt = np.abs(G[405][:, :3].astype(np.float64)).max(1)
    close(slice(12, 15), 405, slice(0, 3), 1e-6, 1e-4, extra=np.where(specular, (4 * EPS23 / cl + cond_b) * gt, 0.0))   # bsdfThroughput
    ge = np.abs(G[406][:, :3].astype(np.float64)).max(1)
    ce = np.maximum(np.abs(o[:, 15].astype(np.float64)), 1e-7)                                          # clamp(cosEnv) at grazing light
    close(slice(16, 19), 406, slice(0, 3), 1e-6, 1e-4, extra=(cond_e + 4 * EPS23 / ce) * ge)            # envThroughput
    # how much of it needed no conditioning term at all: 99.5 % of the non-specular pdfs, 99 % of the directions and throughputs
    # (the specular pdf is the GTR2 lobe itself - at the scenes' roughness^2 of 0.01 its t cancels to 1e-4)
    plain = np.abs(o[h, 3] - G[400][h, 3]) <= 1e-6 + 1e-4 * np.abs(G[400][h, 3])
    assert plain[~specular[h]].mean() > 0.995
    assert (np.abs(o[h, 4:7] - G[403][h, :3]).max(1) <= 1e-5).mean() > 0.99
    assert (np.abs(o[h, 12:15] - G[405][h, :3]) <= 1e-6 + 1e-4 * np.abs(G[405][h, :3])).all(1).mean() > 0.995


@pytest.mark.parametrize("name", SAMPLER_SCENES)
def test_bounce_iteration_accumulation_matches_glsl(name):
    """The rest of the loop iteration (tracer.fs:467,500-512) after the replayed body: emission, the NEE shadow ray and
    its MIS-weighted contribution, the extension ray, accumulatedReflectance, the weighted environment term on a miss.
    Shadow / extension rays and environment lookups are the oracle's own here.  Which terms contribute is equal on every
    ray; the radiance differs by SwiftShader's RGBE decode alone - its alpha conversion error x 255 in the exponent
    (tracer.fs:412), the 0.989 the deterministic envSample probe shows (test_d3_env_lookup_on_miss_pixels, 'brdf3')."""
    z, a, rays, t, idx, G, o = _bounce(name)
    h = idx >= 0
    nxt = G[420][:, 3].astype(np.int32)
    assert (o[h, 39].astype(np.int32) == nxt[h]).mean() >= 0.998          # the extension ray's hit
    same = h & (o[:, 39].astype(np.int32) == nxt)
    assert 0.2 < (nxt[same] < 0).mean() < 0.8
    hit_again = same & (nxt >= 0)
    # grazing hits amplify the 1e-5 of ray.dir; a refracted ray re-hits its own surface at t ~ 1e-5 (origin - 2 eps n)
    dt = np.abs(o[hit_again, 43] - G[421][hit_again, 3]) / np.maximum(G[421][hit_again, 3], 1e-2)
    assert np.median(dt) <= 1e-6 and np.percentile(dt, 90) <= 1e-4 and np.percentile(dt, 99) <= 1e-2
    thr_o, thr_g = o[same, 40:43].astype(np.float64), G[421][same, :3].astype(np.float64)
    rel = np.abs(thr_o - thr_g) / np.maximum(np.abs(thr_g), 1e-6)
    assert rel.max() <= 1e-3 and np.percentile(rel, 99) <= 1e-5           # accumulatedReflectance *= bsdfThroughput
    c_o, c_g = o[same, 36:39].astype(np.float64), G[420][same, :3].astype(np.float64)
    fin = np.isfinite(c_g).all(1)
    assert np.array_equal(np.isfinite(c_o).all(1), fin) and fin.mean() > 0.9
    c_o, c_g = c_o[fin], c_g[fin]
    lit = c_g.max(1) > 1e-6
    assert np.array_equal(c_o.max(1) > 1e-6, lit) and 0.3 < lit.mean() < 0.9   # same shadow-ray and miss decisions
    ratio = c_g[lit].sum(1) / c_o[lit].sum(1)
    bias = np.median(ratio)
    assert 0.986 <= bias <= 0.992, bias
    # environment light carries the bias, emitted light (tracer.fs:467) does not: every ray lies between the two
    assert ratio.min() >= bias * (1 - 0.008) and ratio.max() <= 1.0 + 1e-3, (ratio.min(), ratio.max())
    env_only = o[same, 24:27][fin][lit].max(1) == 0
    assert env_only.mean() > 0.5 and np.abs(ratio[env_only] / bias - 1.0).max() <= 0.008


def _tex_of(G):
    tex = np.zeros((G[411].shape[0], 12), np.float32)
    tex[:, 0:3], tex[:, 3:6], tex[:, 6], tex[:, 7], tex[:, 8:11] = G[411][:, :3], G[412][:, :3], G[411][:, 3], G[412][:, 3], G[413][:, :3]
    return tex


@pytest.mark.parametrize("name", SAMPLER_SCENES)
def test_bounce_body_with_own_texture_fetch(name):
    """The same probe with the oracle's OWN atlas fetches (only rnd() replayed): the GLSL sampler's filter precision
    (<= 3e-4 per channel, stage D3) now reaches the normal map and the roughness - the 'texture-unit rows' at 1e-3."""
    z, a, rays, t, idx, G, o = _bounce(name, with_tex=False)
    h = idx >= 0
    same = h & ((o[:, 1] > 0) == (G[400][:, 1] > 0)) & ((o[:, 2] > 0) == (G[400][:, 2] > 0))
    assert same[h].mean() > 0.995                        # a Fresnel decision can flip on a 3e-4 roughness difference
    assert np.abs(o[same, 32:35] - G[410][same, :3]).max() <= 1e-3            # macroNormal
    assert np.abs(o[same, 8:11] - G[404][same, :3]).max() <= 1e-5             # ray.origin
    assert np.abs(o[same, 24:27] - G[408][same, :3]).max() <= 2e-2            # emitted colour (x30)
    d = np.abs(o[same, 4:7] - G[403][same, :3]).max(1)
    assert np.percentile(d, 99) <= 2e-3                                        # ray.dir


def _ragged(vals, cnt, k=1):
    """[n] counts + flat values -> zero-padded [n, max(count)(, k)]."""
    n, width = cnt.size, max(1, int(cnt.max()))
    out = np.zeros((n, width, k) if k > 1 else (n, width), np.float32)
    off = np.concatenate([[0], np.cumsum(cnt.astype(np.int64))])
    for i in range(n):
        v = vals[off[i] * k:off[i + 1] * k]
        out[i, :cnt[i]] = v.reshape(-1, k) if k > 1 else v
    return out


@pytest.mark.parametrize("name", ["small", "variant", "textured"])
def test_d6_whole_path_replay_matches_glsl(name):
    """Stage D6, the whole path: ONE tick of the reference's UNMODIFIED tracer.fs main() (436-518) at depth 8 on
    SwiftShader - rnd(), intersectScene(), envSample(), main() and (by #define) main()'s four texture() calls keep their
    bodies and are wrapped from outside by recorders (tools/make_goldens.py path_replay) - against the oracle's
    trace_path (the function trace_pixel runs) on the same injected camera rays, with what the GLSL's implementation
    -defined parts returned replayed in three steps:
      rnd            every value rnd() returned (SwiftShader's sin() of a five-digit seed is unrelated to anybody's,
                     SURVEY 0.2), nothing else: the radiance then differs by SwiftShader's RGBE decode (0.989, D5);
      rnd + env      + what envSample() returned per lookup: <= 1e-3 relative L2 (north_star's bar) on the flat-colour and
                     the refractive / emissive scene; the image-mapped scene keeps the sampler's 3e-4 in its roughness;
      rnd + env + tex  + the four texture() results of every iteration (its RGBA8 conversion is 6e-5 off even on flat
                     colours): what is left is the path logic and its float32 arithmetic - 2e-6 / 6e-6 / 1.4e-6.
    'Identical branch sequence' = the same number of rnd() calls (Lambert draws two more than the other branches, every
    iteration six or eight), the same number of intersectScene calls and the same hash over the hit indices they returned
    (primary, shadow and extension rays in call order), the same number of environment lookups: >= 99 % of the pixels."""
    z = np.load(os.path.join(GOLD, f"glsl_path_replay_{name}.npz"))
    a = S.textured_test_scene() if name == "textured" else scene_from_golden(name)
    W, H = int(z["W"]), int(z["H"])
    n = W * H
    assert int(z["bounces"]) == 8
    cnt, cap = z["rnd_count"].reshape(n), int(z["cap"])
    assert cnt.max() <= cap  # nothing was cut off
    rec = _ragged(z["rnd_values"], cnt)
    ec, ti = z["env_count"].reshape(n), z["tex_iters"].reshape(n)
    env, tex = _ragged(z["env_values"], ec, 3), _ragged(z["tex_values"], ti, 12)
    g = z["color"].reshape(n, 3).astype(np.float64)
    assert (cnt > 0).mean() > 0.4 and cnt.max() >= 48 and ti.max() >= 6  # paths of 6+ iterations are in the set

    def rel_l2(x, y):
        return float(np.linalg.norm(x - y) / np.linalg.norm(y))

    def replay(use_env, use_tex):
        col, used, sig, calls, eu = O.path_replay(a, z["rays_pos"], z["rays_dir"], rec, cnt, float(z["rand_base"]),
                                                  float(z["env_theta"]), 8, env if use_env else None, tex if use_tex else None)
        same = (used == cnt) & (sig == z["hit_sig"].reshape(n)) & (calls == z["hit_calls"].reshape(n)) & (eu == ec)
        return col.astype(np.float64), same

    # everything replayed: the path logic and its arithmetic
    o, same = replay(True, True)
    assert same.mean() >= 0.995, same.mean()
    full = rel_l2(g[same], o[same])
    assert full <= 2e-5, full
    assert np.percentile(np.abs(g[same] - o[same]).max(1) / np.maximum(np.abs(o[same]).max(1), 1e-3), 99) <= 1e-4
    # random numbers + environment lookups: north_star's 1e-3 on the HDR buffer
    o, same = replay(True, False)
    assert same.mean() >= 0.99, same.mean()
    env_only = rel_l2(g[same], o[same])
    assert env_only <= (1.2e-2 if name == "textured" else 1.0e-3), env_only
    # random numbers only: SwiftShader's RGBE decode is all that is left
    o, same = replay(False, False)
    assert same.mean() >= 0.99, same.mean()
    r = g[same].sum() / o[same].sum()
    assert 0.985 <= r <= 0.995, r
    if name == "small":  # every photon of this scene comes from the environment
        assert abs(r / _env_decode_bias(a, float(z["env_theta"])) - 1.0) <= 0.002
        assert rel_l2(g[same], o[same] * r) <= 2.5e-3
