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


@pytest.mark.parametrize("name", ["small", "variant"])
def test_d5_converged_mean_matches_glsl(name):
    path = os.path.join(GOLD, f"glsl_converged_{name}.npz")
    z = np.load(path)
    a = scene_from_golden(name)
    W, H, spp, bounces = int(z["W"]), int(z["H"]), int(z["spp"]), int(z["bounces"])
    ga, gb = z["a"][..., :3], z["b"][..., :3]

    def rel_l2(x, y):
        return float(np.linalg.norm(x - y) / np.linalg.norm(y))
    floor = rel_l2(ga, gb)  # GLSL-vs-GLSL, two randBase streams
    acc = np.zeros((H, W, 4), np.float32)
    O.render(a, W, H, [float(x) for x in z["P"]], [float(x) for x in z["I"]], float(z["fov_scale"]),
             [float(x) for x in z["lens"]], float(z["env_theta"]), bounces, 0, spp, 99, acc)
    o = acc[..., :3]
    gm = 0.5 * (ga + gb)
    # whole-image mean within 1 % (+ the ~1 % SwiftShader RGBE-exponent bias on environment light)
    assert abs(o.mean() / gm.mean() - 1.0) <= 0.025, (o.mean(), gm.mean())
    assert rel_l2(o, ga) <= 3.0 * floor and rel_l2(o, gb) <= 3.0 * floor, (rel_l2(o, ga), rel_l2(o, gb), floor)


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
