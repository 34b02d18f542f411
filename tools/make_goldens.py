#!/usr/bin/env python3
"""Generates tests/golden/* from the REFERENCE run in the build container.

  js_*    : the reference's own JS scene pipeline (bvh.js, obj_loader.js,
            env_sampler.js) under Node, via tools/js_ref         -> stage D0
  glsl_*  : the reference's own GLSL (camera.fs, tracer.fs) executed on
            SwiftShader via tools/glsl_oracle                     -> stages D1-D5
            

Needs /root/reference, node and the kaleido SwiftShader; none of them exist on
the GPU box, so only the committed vectors travel.  Re-run with
    python tools/make_goldens.py [js] [glsl] [converged]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "js_ref"))
sys.path.insert(0, os.path.join(ROOT, "tools", "glsl_oracle"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")

from fspt_amd import scene as S  # noqa: E402  (host-side helpers only: OBJ text, env image, packer)


# ---------------------------------------------------------------------------
# inputs shared by generator and tests (kept tiny so the fixtures stay small)
# ---------------------------------------------------------------------------
VARIANT_OBJ = """# quads, negative indices, vn, two groups
v -1 0 -1
v 1 0 -1
v 1 0 1
v -1 0 1
v 0 1.5 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 1 0
vn 0.6 0.8 0
vn -0.6 0.8 0
usemtl floor
f 1/1/1 2/2/1 3/3/1 4/4/1
usemtl roof
f 1/1/2 2/2/2 5/3/1
f -4/2/3 -3/3/3 -1/4/1
f 3/1/2 4/2/3 5/3/1
usemtl floor
f 4/1/1 1/2/1 5/3/1
"""


def small_inputs():
    scene = {"props": S.bunny_props()}
    texts = {"synthetic/cube_sphere.obj": S.cube_sphere_obj(8), "synthetic/quad.obj": S.QUAD_OBJ}
    env, w, h = S.synthetic_env(64, 32)
    return scene, texts, {}, {}, env, w, h


def variant_inputs():
    # the second prop's emissive colour comes from an MTL `Kem` (getMaterial ignores an array-valued
    # scene-JSON `emission`, main.js:249-253)
    props = [
        {"path": "variant.obj", "scale": 0.8, "rotate": [{"angle": 0.3, "axis": [0, 1, 0]}, {"angle": -0.2, "axis": [1, 0, 0]}],
         "translate": [0.1, -0.2, 0.3], "diffuse": [0.9, 0.2, 0.1], "emittance": [0, 0, 0],
         "metallicRoughness": [0, 0.25, 0], "normals": "mesh", "ior": 1.5, "dielectric": 0.5},
        {"path": "lit/variant_em.obj", "scale": 0.5, "rotate": [{"angle": 1.1, "axis": [0, 0, 1]}],
         "translate": [1.5, 0.4, -0.6], "diffuse": [0.2, 0.9, 0.3], "emittance": [0, 0, 0],
         "metallicRoughness": [0, 0.6, 0], "normals": "smooth"},
        {"path": "synthetic/quad.obj", "scale": 6, "rotate": [], "translate": [0, -1.0, 0], "emittance": [0, 0, 0],
         "normals": "flat"},
    ]
    texts = {"variant.obj": VARIANT_OBJ, "lit/variant_em.obj": "mtllib em.mtl\n" + VARIANT_OBJ, "synthetic/quad.obj": S.QUAD_OBJ}
    files = {"lit/em.mtl": "newmtl floor\nKem 0.3 0.3 0.1\nnewmtl roof\nKem 0.3 0.3 0.1\n"}
    # 60x30: midpoint splits go fractional (env_sampler.js:34-36) yet stay under SwiftShader's 261-vec4 uniform limit
    env, w, h = S.synthetic_env(60, 30, sun_deg=8.0)
    return {"props": props}, texts, files, {}, env, w, h


def mtl_inputs():
    """usemtl groups with their own MTL materials (colours and maps), array-index group names (JS iterates
    them first), prop.skips, scene.worldTransforms, scene.normalize, static / animated props, a string ior,
    one image used by several props with different swizzles."""
    lines = S.cube_sphere_obj(3).split("\n")
    faces = [l for l in lines if l.startswith("f ")]
    other = [l for l in lines if not l.startswith("f ")]
    n = len(faces)
    parts = [faces[:n // 5], faces[n // 5:2 * n // 5], faces[2 * n // 5:3 * n // 5], faces[3 * n // 5:4 * n // 5], faces[4 * n // 5:]]
    obj = "\n".join(other + parts[0] + ["mtllib mats.mtl", "usemtl shell"] + parts[1] + ["usemtl 7"] + parts[2] +
                    ["usemtl skipme"] + parts[3] + ["usemtl 2"] + parts[4] + ["usemtl shell"] + faces[:3]) + "\n"
    mtl = "\n".join(["# test library", "newmtl shell", "Kd 0.8 0.1 0.1", "Pmr 0.0 0.4 0.0", "ior 1.7", "newmtl 7",
                     "map_Kd tex/wood.png", "Kem 0.5 0.4 0.1", "dielectric 0.5", "newmtl 2", "map_Pmr tex/mr.png",
                     "pmr_swizzle 2 1 0 3", "map_Bump tex/nrm.png", "map_Kem tex/glow.png", "Ns 0", ""])
    scene = {
        "atlasRes": 64, "normalize": 1.5,
        "worldTransforms": [{"rotate": [{"axis": [0, 1, 0], "angle": 0.4}]}, {"translate": [0.2, -0.1, 0.3]}, {"rotate": []}],
        "props": [{"path": "models/ball.obj", "scale": 0.7, "rotate": [{"angle": 0.3, "axis": [1, 0, 0]}],
                   "translate": [0.1, 0.2, -0.3], "diffuse": [0.2, 0.3, 0.9], "emittance": [0, 0, 0],
                   "metallicRoughness": "models/tex/mr.png", "mrSwizzle": [1, 0, 2, 3], "normals": "smooth",
                   "skips": ["skipme"], "emission": [0.3, 0.3, 0.1]}],
        "static_props": [{"path": "synthetic/quad.obj", "scale": 5, "rotate": [], "translate": [0, -1, 0],
                          "emittance": [0, 0, 0], "normals": "flat", "normal": "models/tex/nrm.png", "ior": "10"}],
        "animated_props": {"a": {"path": "models/ball.obj", "scale": 0.3, "rotate": [], "translate": [1.0, 0.5, 0.0],
                                 "emittance": [1, 1, 1], "normals": "flat", "dielectric": 0}},
    }
    texts = {"models/ball.obj": obj, "synthetic/quad.obj": S.QUAD_OBJ}
    files = {"models/mats.mtl": mtl}
    heights = {"models/tex/wood.png": 16, "models/tex/mr.png": 32, "models/tex/nrm.png": 8, "models/tex/glow.png": 4}
    env, w, h = S.synthetic_env(32, 16)
    return scene, texts, files, heights, env, w, h


# (eye, dir) pairs for shootAutoFocusRay: the bench camera, axis-aligned directions (1/0 = Infinity, 0*Infinity =
# NaN inside the slab test), a miss, an un-normalised direction
FOCUS_RAYS = [[[-0.751, 0.665, 1.820], [0.304, -0.489, -0.818]], [[0.0, 0.0, 2.0], [0.0, 0.0, -1.0]],
              [[0.0, 5.0, 0.0], [0.0, -1.0, 0.0]], [[0.0, 0.5, 3.0], [0.0, 1.0, 0.0]], [[1.0, 2.0, 3.0], [-0.5, -1.1, -1.6]],
              [[0.1, 0.2, 0.3], [0.3, -0.2, 0.1]]]


def stand_in_images(heights):
    rng = np.random.default_rng(5)
    return {u: rng.integers(0, 256, size=(h, h, 4), dtype=np.uint8) for u, h in sorted(heights.items())}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_js():
    import js_ref as J
    for name, (scene, texts, files, heights, env, w, h) in (("small", small_inputs()), ("variant", variant_inputs()),
                                                            ("mtl", mtl_inputs())):
        # the reference's own getMaterial / TexturePacker / mergeSceneProps / obj_loader / mtl_loader / bvh
        out = J.run(J.full_scene_job(scene, texts, files, {u: {"height": hh} for u, hh in heights.items()},
                                     autofocus=FOCUS_RAYS))
        bins = J.run(J.env_job(env, w, h))["bins"]
        np.savez_compressed(os.path.join(GOLD, f"js_scene_{name}.npz"), bvh=out["bvh"], tri=out["tri"], mat=out["mat"],
                            norm=out["norm"], uv=out["uv"], bins=bins, depth=np.int32(out["depth"]),
                            scene=json.dumps(scene), texts=json.dumps(texts), files=json.dumps(files),
                            heights=json.dumps(heights), image_set=json.dumps(out["image_set"]),
                            focus_rays=np.array(FOCUS_RAYS, np.float64), focus=np.array(out["autofocus"], np.float64),
                            env=env, env_w=w, env_h=h)
        print("js", name, out["bvh"].size // 9, "nodes", out["tri"].size // 9, "tris", bins.size // 4, "bins",
              len(out["image_set"]), "layers")
    # env bins on an odd-sized image (NaN / fractional-edge quirks of env_sampler.js:25-47, 73)
    env, w, h = S.synthetic_env(100, 37, sun_deg=6.0)
    np.savez_compressed(os.path.join(GOLD, "js_env_bins_odd.npz"), env=env, env_w=w, env_h=h,
                        bins=J.run(J.env_job(env, w, h))["bins"])
    # 70k-triangle scene: digests only (arrays are ~17 MB)
    props = S.bunny_props()
    texts = {"synthetic/cube_sphere.obj": S.cube_sphere_obj(76), "synthetic/quad.obj": S.QUAD_OBJ}
    t0 = time.time()
    out = J.run(J.full_scene_job({"props": props}, texts))
    env, w, h = S.synthetic_env(2048, 1024)
    bins = J.run(J.env_job(env, w, h))["bins"]
    dig = {k: sha(out[k]) for k in ("bvh", "tri", "mat", "norm", "uv")}
    dig.update(bins=sha(bins), n_bins=int(bins.size // 4), depth=int(out["depth"]), n_nodes=int(out["bvh"].size // 9),
               n_tris=int(out["tri"].size // 9), js_seconds=round(time.time() - t0, 1), js_build_ms=out["build_ms"])
    # Known libm-vs-V8(fdlibm) last-bit differences: Math.atan2/Math.asin of the spherical-UV
    # fallback (obj_loader.js:64-71) feed the tangents; record where the native pipeline (glibc)
    # differs so the test can prove "identical except these floats".
    nat = S.build_scene(props, texts)
    exc = {}
    for k in ("bvh", "tri", "mat", "norm", "uv"):
        d = np.where(getattr(nat, k).view(np.uint32) != out[k].view(np.uint32))[0]
        if d.size:
            ulp = np.abs(getattr(nat, k).view(np.int32)[d].astype(np.int64) - out[k].view(np.int32)[d].astype(np.int64))
            exc[k] = {"index": d.tolist(), "js_bits": out[k].view(np.uint32)[d].tolist(), "max_ulp": int(ulp.max())}
    dig["libm_exceptions"] = exc
    json.dump(dig, open(os.path.join(GOLD, "js_scene_70k_digest.json"), "w"), indent=1)
    print("js 70k", dig)


def make_js_1m():
    """BASELINE configs[2]'s input (1 002 256 triangles, bench.py --config c3) through the REFERENCE's own obj_loader.js +
    bvh.js under Node (~2.5 min, ~4 GB): digests of its packed arrays, and where the native pipeline differs
    (glibc vs V8 fdlibm atan2 / asin in the tangents' spherical-UV fallback, obj_loader.js:64-71)."""
    import js_ref as J
    props = S.bunny_props()
    texts = {"synthetic/cube_sphere.obj": S.cube_sphere_obj(289), "synthetic/quad.obj": S.QUAD_OBJ}
    t0 = time.time()
    out = J.run(J.full_scene_job({"props": props}, texts), max_old_space_mb=24000, raw=True)
    js_s = time.time() - t0
    dig = {k: sha(out[k]) for k in ("bvh", "tri", "mat", "norm", "uv")}
    dig.update(depth=int(out["depth"]), n_nodes=int(out["bvh"].size // 9), n_tris=int(out["tri"].size // 9),
               js_seconds=round(js_s, 1), js_build_ms=out["build_ms"])
    t0 = time.time()
    nat = S.build_scene(props, texts)
    dig["native_seconds"] = round(time.time() - t0, 1)
    exc = {}
    for k in ("bvh", "tri", "mat", "norm", "uv"):
        d = np.where(getattr(nat, k).view(np.uint32) != out[k].view(np.uint32))[0]
        if d.size:
            ulp = np.abs(getattr(nat, k).view(np.int32)[d].astype(np.int64) - out[k].view(np.int32)[d].astype(np.int64))
            exc[k] = {"index": d.tolist(), "js_bits": out[k].view(np.uint32)[d].tolist(), "max_ulp": int(ulp.max())}
    dig["libm_exceptions"] = exc
    json.dump(dig, open(os.path.join(GOLD, "js_scene_1M_digest.json"), "w"), indent=1)
    print("js 1M", {k: v for k, v in dig.items() if k != "libm_exceptions"}, {k: (len(v["index"]), v["max_ulp"]) for k, v in exc.items()})


def load_scene(name):
    """The reference JS pipeline's arrays + the atlas of the same scene."""
    z = np.load(os.path.join(GOLD, f"js_scene_{name}.npz"))
    scene, texts, files = (json.loads(str(z[k])) for k in ("scene", "texts", "files"))
    images = stand_in_images(json.loads(str(z["heights"])))
    nat = S.build_scene_json(scene, texts, files, images)
    return S.SceneArrays(bvh=z["bvh"], tri=z["tri"], mat=z["mat"], norm=z["norm"], uv=z["uv"], atlas=nat.atlas,
                         atlas_res=nat.atlas_res, atlas_layers=nat.atlas_layers, env=z["env"], env_w=int(z["env_w"]),
                         env_h=int(z["env_h"]), bins=z["bins"], leaf_size=4, depth=int(z["depth"]))


FC_REP = "(floor(gl_FragCoord.xy * 0.5) + vec2(0.5))"

# One probe program for every stage (SwiftShader 4.1 crashes on the third program linked in a
# context): the reference's own functions are called unmodified, only main() is replaced and an
# extra `uniform int probeSel` picks what is written.
PROBE_MAIN = """uniform int probeSel;
void main(void) {
  vec2 FC = FCOORD;
  vec4 A = texelFetch(cameraPosTex, ivec2(FC), 0);
  vec4 B = texelFetch(cameraDirTex, ivec2(FC), 0);
  vec4 o = vec4(0.0);
  if (probeSel >= 200) {
    vec3 N = normalize(vec3(0.1, 1.0, 0.2));
    vec3 D = vec3(0.8, 0.6, 0.4);
    if (probeSel == 200) o = vec4(schlick(normalize(A.xyz), N, vec2(1.0, 1.4)), schlick(normalize(A.xyz), N, vec2(1.4, 1.0)),
                                  gtr2Pdf(normalize(A.xyz), N, vec2(B.w, A.w), normalize(B.xyz)), 0.0);
    if (probeSel == 201) o = vec4(evalSpecular(normalize(A.xyz), N, D, vec2(B.w, A.w), normalize(B.xyz)), 0.0);
    if (probeSel == 202) o = vec4(misWeights(A.x, A.y), lambertPdf(N, vec2(0.0), normalize(B.xyz)), 0.0);
    if (probeSel == 203) o = vec4(envSample(normalize(A.xyz)), 0.0);
  } else {
    Ray r = Ray(A.xyz, B.xyz);
    Hit h = intersectScene(r);
    if (probeSel == 100) {
      o = vec4(h.t, float(h.index), 0.0, 1.0);
    } else if (h.index < 0) {
      o = vec4(0.0, 0.0, 0.0, -1.0);
      if (probeSel == 0) o = vec4(envSample(r.dir), -1.0);
    } else {
      Material m = createMaterial(h.index);
      vec3 P = r.origin + r.dir * h.t;
      vec3 w = barycentricWeights(createTriangle(h.index), P);
      vec2 tc = barycentricTexCoord(w, createTexCoords(h.index));
      vec3 tN = (texture(texArray, vec3(tc, m.mapIndices.normal)).rgb - vec3(0.5, 0.5, 0.0)) * vec3(2.0, 2.0, 1.0);
      vec3 bN;
      vec3 mN = barycentricNormal(w, createNormals(h.index), tN, bN);
      if (probeSel == 0) o = vec4(P, h.t);
      if (probeSel == 1) o = vec4(w, float(h.index));
      if (probeSel == 2) o = vec4(tc, texture(texArray, vec3(tc, m.mapIndices.roughness)).rg);
      if (probeSel == 3) o = vec4(texture(texArray, vec3(tc, m.mapIndices.diffuse)).rgb, m.ior);
      if (probeSel == 4) o = vec4(tN, m.dielectric);
      if (probeSel == 5) o = vec4(mN, 1.0);
      if (probeSel == 6) o = vec4(bN, 1.0);
      if (probeSel == 7) o = vec4(texture(texArray, vec3(tc, m.mapIndices.specular)).rgb, 1.0);
    }
  }
  fragColor = o;
}
"""
HIT, BRDF = 100, 200

# ---- stochastic half of tracer.fs: samplers and one bounce-loop iteration, with the GLSL's rnd() values recorded ----
# 300-304: seed = B.x (injected), A = (normal.xyz, metallicRoughness.y).  rnd() eight times; then - seed reset - the
#          reference's UNMODIFIED sampleMicrofacet / sampleLambert / sampleEnv.
# 400-413: one iteration of main()'s bounce loop.  BODY1 / BODY2 are the lines of tracer.fs main() from
#          `Material mat = createMaterial(result.index);` to the seed assignment, and from there to
#          `vec2 weights = misWeights(envDirPdf.a, bsdfPdf);`, cut out of /root/reference/shader/tracer.fs when the
#          goldens are generated (bounce_body_parts) - the only inserted statement is the copy of `seed` between them.
# 420-421: BODY3 = the rest of the iteration (shadow ray, extension ray, MIS-weighted accumulation, tracer.fs:500-512)
#          inside a one-trip loop that gives its `break` something to leave.
SAMPLER_MAIN = """uniform int probeSel;
void main(void) {
  vec2 FC = FCOORD;
  vec4 A = texelFetch(cameraPosTex, ivec2(FC), 0);
  vec4 B = texelFetch(cameraDirTex, ivec2(FC), 0);
  vec4 o = vec4(0.0);
  if (probeSel == 100) {
    Hit h = intersectScene(Ray(A.xyz, B.xyz));
    o = vec4(h.t, float(h.index), 0.0, 1.0);
  } else if (probeSel < 400) {
    seed = B.x;
    float r1 = rnd(); float r2 = rnd(); float r3 = rnd(); float r4 = rnd();
    float r5 = rnd(); float r6 = rnd(); float r7 = rnd(); float r8 = rnd();
    if (probeSel == 300) o = vec4(r1, r2, r3, r4);
    if (probeSel == 301) o = vec4(r5, r6, r7, r8);
    seed = B.x;
    if (probeSel == 302) { vec3 m = sampleMicrofacet(A.xyz, vec2(0.0, A.w)); o = vec4(m, seed); }
    if (probeSel == 303) { vec3 m = sampleLambert(A.xyz); o = vec4(m, seed); }
    if (probeSel == 304) o = sampleEnv();
  } else {
    Ray ray = Ray(A.xyz, B.xyz);
    Hit result = intersectScene(ray);
    if (result.index < 0) {
      o = vec4(0.0, 0.0, 0.0, -1.0);
    } else {
      vec3 color = vec3(0);
      vec3 accumulatedReflectance = vec3(1);
      int i = 0;
BODY1
      float probeSeed0 = seed;
BODY2
      if (probeSel == 400) o = vec4(probeSeed0, float(inside), float(specular), bsdfPdf);
      if (probeSel == 403) o = vec4(ray.dir, weights.x);
      if (probeSel == 404) o = vec4(ray.origin, weights.y);
      if (probeSel == 405) o = vec4(bsdfThroughput, cosEnv);
      if (probeSel == 406) o = vec4(envThroughput, envDirPdf.a);
      if (probeSel == 407) o = vec4(envDirPdf.xyz, float(i));
      if (probeSel == 408) o = vec4(color, seed);
      if (probeSel == 409) o = vec4(microNormal, 0.0);
      if (probeSel == 410) o = vec4(macroNormal, mat.dielectric);
      vec2 rawMR = texture(texArray, vec3(texCoord, mat.mapIndices.roughness)).rg;
      if (probeSel == 411) o = vec4(texDiffuse, rawMR.x);
      if (probeSel == 412) o = vec4(texEmmissive, rawMR.y);
      if (probeSel == 413) o = vec4(texture(texArray, vec3(texCoord, mat.mapIndices.normal)).rgb, 0.0);
      seed = probeSeed0;
      float r1 = rnd(); float r2 = rnd(); float r3 = rnd(); float r4 = rnd();
      float r5 = rnd(); float r6 = rnd(); float r7 = rnd(); float r8 = rnd();
      if (probeSel == 401) o = vec4(r1, r2, r3, r4);
      if (probeSel == 402) o = vec4(r5, r6, r7, r8);
      if (probeSel >= 420) {
        for (int probeOnce = 0; probeOnce < 1; ++probeOnce) {
BODY3
        }
        if (probeSel == 420) o = vec4(color, float(result.index));
        if (probeSel == 421) o = vec4(accumulatedReflectance, result.t);
      }
    }
  }
  fragColor = o;
}
"""
BOUNCE_SELS = (400, 401, 402, 403, 404, 405, 406, 407, 408, 409, 410, 411, 412, 413, 420, 421)


def bounce_body_parts():
    """The bounce-loop body of tracer.fs main() (tracer.fs:447-499), read from the reference NOW."""
    import glsl_ref as G
    src = G.read_shader("tracer.fs")
    main = src[src.index("void main(void) {"):]
    first = "Material mat = createMaterial(result.index);"
    seed_line = "seed = origin.x * randBase * origin.y * 1.396529836 + origin.z * 4761.52835;"
    last = "vec2 weights = misWeights(envDirPdf.a, bsdfPdf);"
    for anchor in (first, seed_line, last):
        assert main.count(anchor) == 1, anchor
    a, b, c = main.index(first), main.index(seed_line) + len(seed_line), main.index(last) + len(last)
    brk = main.index("break;", c)
    assert main.count("break;") == 1
    d = main.index("}", brk) + 1
    return main[a:b], main[b:c], main[c:d]


def sampler_rays(arrays, W, H, cam, lens, seed):
    """Rays for the bounce probe: the upper half of the image are camera rays; the lower half start BEHIND a randomly
    chosen triangle and hit it from the back (inside = true: refraction out of the dielectric, total internal reflection
    at grazing angles, the Beer override on back-facing non-dielectrics, tracer.fs:461-463,497)."""
    import oracle as O
    pos, d = O.camera(W, H, cam["P"], cam["I"], cam["fov_scale"], lens, 777.0)
    rng = np.random.default_rng(seed)
    tri = arrays.tri.reshape(-1, 3, 3).astype(np.float64)
    n = (H // 2) * W
    pick = rng.integers(0, tri.shape[0], n)
    v = tri[pick]
    w = rng.dirichlet([2.0, 2.0, 2.0], n)
    p = (v * w[:, :, None]).sum(1)
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    nrm = np.cross(e1, e2)
    size = np.sqrt(np.linalg.norm(nrm, axis=1, keepdims=True))   # ~ edge length
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    t1 = e1 / np.maximum(np.linalg.norm(e1, axis=1, keepdims=True), 1e-30)
    t2 = np.cross(nrm, t1)
    depth = size * rng.uniform(0.3, 2.0, (n, 1))
    r = depth * np.tan(rng.uniform(0.0, 1.0, (n, 1)) ** 1.5 * 1.35)  # incidence angle up to 77 degrees, half of them below 27
    phi = rng.uniform(0, 2 * np.pi, (n, 1))
    side = np.where(rng.uniform(size=(n, 1)) < 0.6, -1.0, 1.0)   # from behind the winding normal, or from the front (shading
    #                                                             normals may point either way: 'mesh' normals, flipped quads)
    o = p + side * nrm * depth + (t1 * np.cos(phi) + t2 * np.sin(phi)) * r
    dd = p - o
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    pos[H // 2:, :, :3] = o.reshape(H // 2, W, 3).astype(np.float32)
    d[H // 2:, :, :3] = dd.reshape(H // 2, W, 3).astype(np.float32)
    return pos, d


def make_samplers(name):
    """Deterministic pins for sampleMicrofacet / sampleLambert / sampleEnv and one bounce-loop iteration."""
    import glsl_ref as G
    scene_name = name
    arrays = converged_scene(scene_name)
    cam = dict(S.BUNNY_CAMERA)
    if name in CONVERGED and CONVERGED[name][5] is not None:
        cam["P"], cam["I"] = CONVERGED[name][5], CONVERGED[name][6]
    lens = S.lens_features(cam["focal_depth"], cam["aperture"])
    g = G.GlslRef()
    g.scene(arrays)
    W, H = 64, 40
    g.target(W, H, replicate=True)
    b1, b2, b3 = bounce_body_parts()
    g.tracer(main_override=SAMPLER_MAIN.replace("BODY1", b1).replace("BODY2", b2).replace("BODY3", b3).replace("FCOORD", FC_REP))
    env_theta, rand_base = cam["env_theta"], 4242.5
    out = dict(W=W, H=H, renderer=g.renderer, env_theta=np.float32(env_theta), rand_base=np.float32(rand_base), scene=scene_name)

    def run(sel):
        g.set_int("probeSel", sel)
        g.draw_tracer(0, rand_base, env_theta)
        img, mism = g.read_screen(0)
        assert mism == 0, f"{mism} replica mismatches"
        return img

    # (i)/(ii) samplers on injected (normal, roughness, seed)
    rng = np.random.default_rng(21)
    N = rng.normal(size=(H, W, 3))
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    N[:4] = [0.0, 0.0, 1.0]; N[4:8] = [0.0, 0.0, -1.0]           # the |n.z| >= 0.999 frame (tracer.fs:259,275)
    N[8:10] = N[8:10] * [0.03, 0.03, 1.0]; N[8:10] /= np.linalg.norm(N[8:10], axis=-1, keepdims=True)
    A = np.zeros((H, W, 4), np.float32); A[..., :3] = N
    A[..., 3] = rng.uniform(0.0, 1.0, (H, W)) ** 2                # metallicRoughness.y arrives squared (tracer.fs:457)
    A[::5, ::3, 3] = 0.0004                                       # below the 0.001 clamp (tracer.fs:261)
    B = np.zeros((H, W, 4), np.float32)
    B[..., 0] = np.where(rng.uniform(size=(H, W)) < 0.5, rng.uniform(-30, 30, (H, W)), rng.uniform(-2e4, 2e4, (H, W)))
    g.set_camera(A, B)
    out["samp_A"], out["samp_B"] = A, B
    for sel, key in ((300, "samp_rnd_a"), (301, "samp_rnd_b"), (302, "samp_microfacet"), (303, "samp_lambert"), (304, "samp_env")):
        out[key] = run(sel)
    # (iii) one bounce-loop iteration
    pos, d = sampler_rays(arrays, W, H, cam, lens, 5)
    g.set_camera(pos, d)
    out["rays_pos"], out["rays_dir"] = pos, d
    hit = run(HIT)
    out["hit_t"], out["hit_index"] = hit[..., 0], hit[..., 1].astype(np.int32)
    for sel in BOUNCE_SELS:
        out[f"bounce{sel}"] = run(sel)
    ins = out["bounce400"][..., 1][out["hit_index"] >= 0]
    spec = out["bounce400"][..., 2][out["hit_index"] >= 0]
    refr = out["bounce407"][..., 3][out["hit_index"] >= 0]
    print("samplers", name, "hits", float((out["hit_index"] >= 0).mean()), "inside", float(ins.mean()), "specular", float(spec.mean()),
          "refracted", float((refr < 0).mean()))
    np.savez_compressed(os.path.join(GOLD, f"glsl_samplers_{name}.npz"), **out)




# ---- stage D6: the WHOLE path, tracer.fs main() unmodified, with every random number it drew recorded -------------------
# The reference's rnd(), intersectScene() and main() keep their bodies and get new names; wrappers with the old names
# count / record what goes through them (the only way to see inside a fragment shader is its one vec4 output, so the
# program is run once per group of four rnd() values: probeSel 2, probeBase = 0, 4, 8, ...; every run of the same
# (rays, randBase) is the same computation).  probeSel 0 leaves main()'s own output (tick 0: the clamped sample colour,
# tracer.fs:515-517); 1 = (rnd() calls, hash of the hit indices intersectScene returned - low / high 16 bits -, calls).
PATH_WRAP_RND = """
uniform int probeSel;
uniform int probeBase;
int probeCount = 0;
vec4 probeRec = vec4(0.0);
float rnd() {
  float r = rndRef();
  int k = probeCount - probeBase;
  if (k == 0) probeRec.x = r;
  if (k == 1) probeRec.y = r;
  if (k == 2) probeRec.z = r;
  if (k == 3) probeRec.w = r;
  probeCount++;
  return r;
}
"""
PATH_WRAP_HIT = """
uint probeSig = 0u;
int probeCalls = 0;
Hit intersectScene(Ray ray) {
  Hit h = intersectSceneRef(ray);
  probeSig = probeSig * 31u + uint(h.index + 2);
  probeCalls++;
  return h;
}
"""
PATH_WRAP_ENV = """
uniform int probeEnvIdx;
int probeEnvCount = 0;
vec3 probeEnv = vec3(0.0);
vec3 envSample(vec3 dir) {
  vec3 e = envSampleRef(dir);
  if (probeEnvCount == probeEnvIdx) probeEnv = e;
  probeEnvCount++;
  return e;
}
"""
# texture() is a built-in: main()'s four atlas fetches per iteration (tracer.fs:453-456; envColor's fetch sits above this
# point of the file and keeps the built-in) are routed through a counting wrapper by the preprocessor
PATH_WRAP_TEX = """
uniform int probeTexIdx;
int probeTexCount = 0;
vec4 probeTex = vec4(0.0);
vec4 probeTexture(sampler2DArray s, vec3 c) {
  vec4 v = texture(s, c);
  if (probeTexCount == probeTexIdx) probeTex = v;
  probeTexCount++;
  return v;
}
#define texture probeTexture
"""
PATH_WRAP_MAIN = """
#undef texture
void main(void) {
  mainRef();
  if (probeSel == 1) fragColor = vec4(float(probeCount), float(probeSig & 0xFFFFu), float(probeSig >> 16u), float(probeCalls));
  if (probeSel == 2) fragColor = probeRec;
  if (probeSel == 3) fragColor = vec4(probeEnv, float(probeEnvCount));
  if (probeSel == 4) fragColor = probeTex;
  if (probeSel == 5) fragColor = vec4(float(probeTexCount), 0.0, 0.0, 0.0);
}
"""


def path_wrap(src):
    """tracer.fs (as tracer_source made it: #defines, NUM_BOUNCES, the quad-replication substitutions) with the three
    wrappers spliced in behind the functions they wrap."""
    def after_function(src, head, new_head, wrapper):
        assert src.count(head) == 1, head
        i = src.index(head)
        depth, j = 0, src.index("{", i)
        while True:  # the function's closing brace
            depth += {"{": 1, "}": -1}.get(src[j], 0)
            j += 1
            if depth == 0:
                break
        return src[:i] + new_head + src[i + len(head):j] + "\n" + wrapper + src[j:]
    src = after_function(src, "float rnd() {", "float rndRef() {", PATH_WRAP_RND)
    src = after_function(src, "Hit intersectScene(Ray ray){", "Hit intersectSceneRef(Ray ray){", PATH_WRAP_HIT)
    src = after_function(src, "vec3 envSample(vec3 dir){", "vec3 envSampleRef(vec3 dir){", PATH_WRAP_ENV)
    assert src.count("texture(") == 5 and src.index("texture(envTex") < src.index("void main(void) {") < src.index("texture(texArray")
    i = src.index("void main(void) {")
    src = src[:i] + PATH_WRAP_TEX + src[i:]
    src = after_function(src, "void main(void) {", "void mainRef(void) {", PATH_WRAP_MAIN)
    return src


PATH_REPLAY_CAP = 192  # rnd() values recorded per pixel at most (8 bounces x 8 values; refraction adds iterations)


def make_path_replay(name):
    """D6 golden: camera rays of the oracle (injected), ONE tick of the unmodified main() at depth 8."""
    import glsl_ref as G
    import oracle as O
    scene_name = name
    arrays = converged_scene(scene_name)
    cam = dict(S.BUNNY_CAMERA)
    if name in CONVERGED and CONVERGED[name][5] is not None:
        cam["P"], cam["I"] = CONVERGED[name][5], CONVERGED[name][6]
    lens = S.lens_features(cam["focal_depth"], cam["aperture"])
    W, H, bounces = 64, 40, 8
    env_theta, rand_base = cam["env_theta"], 3217.25
    pos, d = O.camera(W, H, cam["P"], cam["I"], cam["fov_scale"], lens, 911.5)
    g = G.GlslRef()
    g.scene(arrays)
    g.target(W, H, replicate=True)
    g.tracer(num_bounces=bounces, post=path_wrap)
    g.set_camera(pos, d)

    def run(sel, base=0, env_idx=0, tex_idx=0):
        g.set_int("probeSel", sel)
        g.set_int("probeBase", base)
        g.set_int("probeEnvIdx", env_idx)
        g.set_int("probeTexIdx", tex_idx)
        g.clear()
        g.draw_tracer(0, rand_base, env_theta)
        img, mism = g.read_screen(0)
        assert mism == 0, f"{mism} replica mismatches"
        return img

    t0 = time.time()
    color = run(0)
    info = run(1)
    count = info[..., 0].astype(np.uint32)
    sig = (info[..., 1].astype(np.uint32) | (info[..., 2].astype(np.uint32) << 16))
    calls = info[..., 3].astype(np.uint32)
    n_rec = int(min(count.max(), PATH_REPLAY_CAP))
    n_rec = (n_rec + 3) // 4 * 4
    rec = np.zeros((H, W, max(n_rec, 4)), np.float32)
    for base in range(0, n_rec, 4):
        rec[..., base:base + 4] = run(2, base)
    # what envSample returned for the path's k-th environment lookup (primary miss / NEE / the ray that leaves the scene)
    e0 = run(3, 0, 0)
    env_count = e0[..., 3].astype(np.uint32)
    env = np.zeros((H, W, max(1, int(env_count.max())), 3), np.float32)
    env[..., 0, :] = e0[..., :3]
    for k in range(1, env.shape[2]):
        env[..., k, :] = run(3, 0, k)[..., :3]
    env_flat = np.concatenate([env[y, x, :env_count[y, x]].reshape(-1) for y in range(H) for x in range(W)])
    # the four texture() results of every loop iteration (12 floats per iteration: diffuse.rgb, emissive.rgb, mr.rg, normal.rgb, 0)
    tex_count = run(5)[..., 0].astype(np.uint32)
    assert (tex_count % 4 == 0).all()
    iters = tex_count // 4
    tex = np.zeros((H, W, max(1, int(iters.max())), 12), np.float32)
    for k in range(int(tex_count.max())):
        v = run(4, 0, 0, k)
        it, which = k // 4, k % 4
        if which == 0: tex[..., it, 0:3] = v[..., :3]
        elif which == 1: tex[..., it, 3:6] = v[..., :3]
        elif which == 2: tex[..., it, 6:8] = v[..., :2]
        else: tex[..., it, 8:11] = v[..., :3]
    tex_flat = np.concatenate([tex[y, x, :iters[y, x]].reshape(-1) for y in range(H) for x in range(W)])
    # ragged: only the values a pixel really drew
    keep = np.minimum(count, rec.shape[-1])
    flat = np.concatenate([rec[y, x, :keep[y, x]] for y in range(H) for x in range(W)]) if keep.sum() else np.zeros(0, np.float32)
    print("path_replay", name, "rnd calls mean %.1f max %d" % (count.mean(), count.max()), "intersect calls max", int(calls.max()),
          "hit pixels %.2f" % float((count > 0).mean()), "colour mean", color[..., :3].mean((0, 1)), round(time.time() - t0, 1), "s", flush=True)
    np.savez_compressed(os.path.join(GOLD, f"glsl_path_replay_{name}.npz"), W=W, H=H, bounces=bounces, scene=scene_name,
                        renderer=g.renderer, env_theta=np.float32(env_theta), rand_base=np.float32(rand_base),
                        rays_pos=pos, rays_dir=d, color=color[..., :3].copy(), rnd_count=count, hit_sig=sig, hit_calls=calls,
                        rnd_values=flat.astype(np.float32), cap=np.int32(rec.shape[-1]), env_count=env_count,
                        env_values=env_flat.astype(np.float32), tex_iters=iters, tex_values=tex_flat.astype(np.float32))


def probe(g, sel):
    if not getattr(g, "_probe_ready", False):
        g.tracer(main_override=PROBE_MAIN.replace("FCOORD", FC_REP if g.rep == 2 else "gl_FragCoord.xy"))
        g._probe_ready = True
    g.set_int("probeSel", sel)
    g.draw_tracer(0, 0.0, g._env_theta)
    img, mism = g.read_screen(0)
    assert mism == 0, f"{mism} replica mismatches"
    return img


def make_glsl(name="small"):
    import glsl_ref as G
    import oracle as O
    cam = dict(S.BUNNY_CAMERA)
    if name != "small":  # the camera of the converged golden of the same scene
        cam["P"], cam["I"] = CONVERGED[name][5], CONVERGED[name][6]
    lens = S.lens_features(cam["focal_depth"], cam["aperture"])
    g = G.GlslRef()
    print(g.renderer)
    arrays = load_scene(name)
    g.scene(arrays)
    g._env_theta = cam["env_theta"]
    W, H = 64, 40
    g.target(W, H, replicate=True)
    out = dict(W=W, H=H, renderer=g.renderer, env_theta=np.float32(cam["env_theta"]))
    # D1: the GLSL's own camera rays (informational: depends on SwiftShader's sin())
    g.draw_camera(cam["P"], cam["I"], cam["fov_scale"], lens, 1234.5)
    out["cam_pos"], out["cam_dir"] = g.read_camera()
    out["cam_rand_base"] = np.float32(1234.5)
    # rays both sides trace from here on: aperture 0 pinhole rays without jitter dependence -> take
    # the oracle's camera at a fixed randBase and inject them ("inject the oracle's ray textures")
    pos, d = O.camera(W, H, cam["P"], cam["I"], cam["fov_scale"], lens, 4321.0)
    g.set_camera(pos, d)
    out["rays_pos"], out["rays_dir"] = pos, d
    t0 = time.time()
    hit = probe(g, HIT)                     # D2
    out["hit_t"], out["hit_index"] = hit[..., 0], hit[..., 1].astype(np.int32)
    for sel in range(8):                             # D3
        out[f"shade{sel}"] = probe(g, sel)
    rng = np.random.default_rng(7)
    A = rng.normal(size=(H, W, 4)).astype(np.float32)
    B = rng.normal(size=(H, W, 4)).astype(np.float32)
    A[..., 1] = np.abs(A[..., 1]) + 0.05           # incident in N's hemisphere (mostly)
    B[..., 1] = np.abs(B[..., 1]) + 0.05
    A[..., 3] = rng.uniform(0.002, 1.0, size=(H, W))  # rough
    B[..., 3] = rng.uniform(0.0, 1.0, size=(H, W))    # metallic
    g.set_camera(A, B)
    out["brdf_A"], out["brdf_B"] = A, B
    for sel in range(3):
        out[f"brdf{sel}"] = probe(g, BRDF + sel)
    A2 = A.copy(); A2[..., 3] = np.float32(cam["env_theta"])  # envSample uses the envTheta uniform
    g.set_camera(A2, B)
    out["brdf3"] = probe(g, BRDF + 3)
    print("probes", round(time.time() - t0, 1), "s", "hits", float((out["hit_index"] >= 0).mean()))
    np.savez_compressed(os.path.join(GOLD, f"glsl_stages_{name}.npz"), **out)



# key -> (scene, W, H, spp, NUM_BOUNCES, P, I).  16 384 spp: GLSL-vs-GLSL noise floor ~0.015 (small) / ~0.026 (variant);
# depth 8 is the BASELINE depth (tracer.fs:9 ships 4: kept for 'small_d4'); 'textured' has 16x16 image maps on every layer.
def camera_probe_source(sel):
    """The reference's camera.fs, cut and patched when the goldens are made (never stored): its rnd() renamed and wrapped
    so that the four values it returns are kept, and main()'s two outputs carrying two of them (recv[2 sel], recv[2 sel + 1])
    in their w components (the reference writes the constant 1 there)."""
    import glsl_ref as G
    src = G.read_shader("camera.fs")
    assert src.count("float rnd()") == 1 and src.count("vec3 getScreen(") == 1
    src = src.replace("float rnd()", "float rnd0()")
    wrap = "float recv[4]; int reci = 0;\nfloat rnd() { float v = rnd0(); recv[reci] = v; reci++; return v; }\n\n"
    src = src.replace("vec3 getScreen(", wrap + "vec3 getScreen(")
    end = src.rstrip().rfind("}")
    return src[:end] + f"  fragColor[0].w = recv[{2 * sel}]; fragColor[1].w = recv[{2 * sel + 1}];\n" + src[end:]


def make_camera_replay():
    """D1 with replay: camera.fs main() on SwiftShader with the values its rnd() returned recorded next to the rays
    (two draws per case: the w components hold two values each), for the oracle to replay (oracle_camera_probe)."""
    import glsl_ref as G
    g = G.GlslRef()
    W, H = 64, 40
    g.target(W, H, replicate=False)
    cam = dict(S.BUNNY_CAMERA)
    cases = [("bunny", cam["P"], cam["I"], cam["aperture"], 1234.5), ("bunny_dof", cam["P"], cam["I"], 0.1, 977.25),
             ("variant", [0.3, 1.2, 3.4], [-0.05, -0.3, -0.95], 0.02, 5000.75), ("down", [0.1, 3.0, 0.2], [0.02, -1.0, 0.05], 0.05, 0.5)]
    out = dict(W=W, H=H, renderer=g.renderer, names=json.dumps([c[0] for c in cases]))
    for name, P, I, ap, rb in cases:
        lens = S.lens_features(cam["focal_depth"], ap)
        rec = np.zeros((H, W, 4), np.float32)
        rays = []
        for sel in (0, 1):
            g._ck(g.lib.gh_camera_program(G.read_shader("camera.vs").encode(), camera_probe_source(sel).encode()))
            g.draw_camera(P, I, cam["fov_scale"], lens, rb)
            pos, d = g.read_camera()
            rec[..., 2 * sel], rec[..., 2 * sel + 1] = pos[..., 3], d[..., 3]
            rays.append((pos[..., :3].copy(), d[..., :3].copy()))
        assert np.array_equal(rays[0][0], rays[1][0]) and np.array_equal(rays[0][1], rays[1][1])  # the two draws are the same draw
        assert (rec >= 0).all() and (rec < 1).all()
        out.update({f"{name}_P": np.float32(P), f"{name}_I": np.float32(I), f"{name}_lens": np.float32(lens),
                    f"{name}_fov": np.float32(cam["fov_scale"]), f"{name}_rand_base": np.float32(rb), f"{name}_rec": rec,
                    f"{name}_pos": rays[0][0], f"{name}_dir": rays[0][1]})
        print("camera replay", name, "rnd mean", rec.mean((0, 1)))
    np.savez_compressed(os.path.join(GOLD, "glsl_camera_replay.npz"), **out)


CONVERGED = {"small": ("small", 48, 32, 16384, 8, None, None),
             "small_d4": ("small", 48, 32, 16384, 4, None, None),
             "variant": ("variant", 48, 32, 16384, 8, [0.3, 1.2, 3.4], [-0.05, -0.3, -0.95]),
             "textured": ("textured", 48, 32, 16384, 8, None, None),
             # BASELINE configs[4]'s lens (aperture 0.1 instead of 0.02): getDOF's disc is 5x wider, every sample's origin moves
             "small_dof": ("small", 48, 32, 16384, 8, None, None, 0.1),
             # depth 1: direct light only - one shading event, its NEE sample and its MIS-weighted extension ray, nothing averaged over later bounces
             # (on the flat-colour scene: with emitters in view a single RGBE-bias factor no longer fits every pixel at depth 1)
             "small_d1": ("small", 48, 32, 16384, 1, None, None)}


def converged_scene(scene_name):
    return S.textured_test_scene() if scene_name == "textured" else load_scene(scene_name)


def make_converged(name):
    """D5: converged mean of the unmodified tracer.fs (quad-replicated), two independent
    randBase streams (their difference is the oracle-vs-oracle noise floor).  One scene per
    process: a fresh GL context holds exactly one tracer program."""
    import glsl_ref as G
    import oracle as O
    cam = dict(S.BUNNY_CAMERA)
    scene_name, Wc, Hc, spp, bounces, P, I = CONVERGED[name][:7]
    lens = S.lens_features(cam["focal_depth"], CONVERGED[name][7] if len(CONVERGED[name]) > 7 else cam["aperture"])
    P = P or cam["P"]; I = I or cam["I"]
    arrays = converged_scene(scene_name)
    g = G.GlslRef()
    g.scene(arrays)
    g.target(Wc, Hc, replicate=True)
    g.tracer(num_bounces=bounces)
    imgs = []
    for stream in (11, 12):
        rbs = O.rand_base_stream(stream, 2 * spp)
        g.clear()
        t0 = time.time()
        for k in range(spp):
            g.draw_camera(P, I, cam["fov_scale"], lens, rbs[2 * k])
            g.draw_tracer(k, rbs[2 * k + 1], cam["env_theta"])
        img, mm = g.read_screen((spp - 1) % 2)
        print("converged", name, "stream", stream, spp, "spp", round(time.time() - t0, 1), "s", "mismatch", mm,
              "mean", img[..., :3].mean((0, 1)), flush=True)
        assert mm == 0
        imgs.append(img)
    np.savez_compressed(os.path.join(GOLD, f"glsl_converged_{name}.npz"), a=imgs[0], b=imgs[1], W=Wc, H=Hc, spp=spp,
                        bounces=bounces, P=np.float32(P), I=np.float32(I), fov_scale=np.float32(cam["fov_scale"]),
                        lens=np.float32(lens), env_theta=np.float32(cam["env_theta"]), renderer=g.renderer,
                        streams=np.int32([11, 12]), scene=scene_name)


def make_bvh_test():
    """bvh_test.fs (mode=test) on SwiftShader: traversal-iteration counts per pixel for injected camera rays, two
    ticks (the second folds into the running mean) -> exact integers, stage D2's work counters."""
    import glsl_ref as G
    import oracle as O
    out = {}
    for name, (W, H) in (("small", (64, 40)), ("mtl", (48, 32))):
        cam = dict(S.BUNNY_CAMERA)
        lens = S.lens_features(cam["focal_depth"], cam["aperture"])
        g = G.GlslRef()
        arrays = load_scene(name)
        g.scene(arrays)
        g.target(W, H, replicate=True)
        g.tracer_test()
        imgs = []
        for tick, rb in enumerate((4321.0, 99.5)):
            pos, d = O.camera(W, H, cam["P"], cam["I"], cam["fov_scale"], lens, rb)
            g.set_camera(pos, d)
            g.draw_tracer(tick, 1.0, 0.0)
            img, mism = g.read_screen(tick)
            assert mism == 0, mism
            out[f"{name}_pos{tick}"], out[f"{name}_dir{tick}"], out[f"{name}_img{tick}"] = pos, d, img
            imgs.append(img)
        print("bvh_test", name, "max count", float(imgs[0][..., 0].max()) / 0.001, "replica mismatches 0")
        del g
    np.savez_compressed(os.path.join(GOLD, "glsl_bvh_test.npz"), **out)


def make_fuzz_bvh_test():
    """bvh_test.fs on the random triangle soups of tests/test_parity_gpu.py::_fuzz_scene (irregular trees, leaf
    sizes 1-5, degenerate / duplicate / sliver / huge triangles): exact traversal-iteration counts per pixel."""
    import glsl_ref as G
    import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_parity_gpu import _fuzz_scene
    out = {}
    W, H = 40, 24
    for seed in (0, 1, 2, 3, 5, 8):
        arrays, cam, _, _, _ = _fuzz_scene(seed)
        if arrays.env is None:  # the harness always binds an environment texture; traversal does not read it
            arrays.env, arrays.env_w, arrays.env_h = np.zeros(2 * 2 * 4, np.uint8), 2, 2
        g = G.GlslRef()
        g.scene(arrays)
        g.target(W, H, replicate=True)
        g.tracer_test(leaf_size=arrays.leaf_size)
        pos, d = O.camera(W, H, cam["P"], cam["I"], cam["fov_scale"], cam["lens"], 4321.0)
        g.set_camera(pos, d)
        g.draw_tracer(0, 1.0, 0.0)
        img, mism = g.read_screen(0)
        assert mism == 0, mism
        acc = np.zeros((H, W, 4), np.float32)
        O.trace_test(arrays, W, H, pos, d, 0, acc)
        same = np.array_equal(acc.view(np.uint32), img.view(np.uint32))
        print("fuzz bvh_test seed", seed, "leaf", arrays.leaf_size, "tris", arrays.n_tris, "max count", img[..., 0].max() / 0.001,
              "oracle == glsl:", same, "" if same else int((acc != img).any(-1).sum()))
        for k in ("bvh", "tri"):
            out[f"s{seed}_{k}"] = getattr(arrays, k)
        out[f"s{seed}_leaf"] = np.int32(arrays.leaf_size)
        out[f"s{seed}_pos"], out[f"s{seed}_dir"], out[f"s{seed}_img"] = pos, d, img
        del g
        # D2 on the same rays through tracer.fs's own intersectScene (probe main): closest hit incl. the
        # first-visited-wins ties of the duplicated triangles
        g = G.GlslRef()
        g.scene(arrays)
        g._env_theta = 0.0
        g.target(W, H, replicate=True)
        g.set_camera(pos, d)
        src_leaf = arrays.leaf_size
        g.tracer(main_override=PROBE_MAIN.replace("FCOORD", FC_REP), leaf_size=src_leaf)
        g._probe_ready = True
        hit = probe(g, HIT)
        t, idx, _, _ = O.intersect(arrays, np.concatenate([pos[..., :3], d[..., :3]], -1).reshape(-1, 6))
        gi = hit[..., 1].astype(np.int32).reshape(-1)
        rel = np.abs(hit[..., 0].reshape(-1)[idx >= 0] - t[idx >= 0]) / np.maximum(t[idx >= 0], 1e-6)
        print("   D2 index equal:", float((gi == idx).mean()), "hits", int((idx >= 0).sum()), "t rel max", float(rel.max()) if rel.size else 0.0)
        out[f"s{seed}_hit_t"], out[f"s{seed}_hit_index"] = hit[..., 0], hit[..., 1].astype(np.int32)
        del g
    np.savez_compressed(os.path.join(GOLD, "glsl_fuzz_bvh_test.npz"), **out)


def make_draw():
    """draw.fs (tonemap + firefly filter) on SwiftShader: HDR input = a converged golden image, with a few
    injected fireflies so the 5x5 filter has something to do."""
    import glsl_ref as G
    z = np.load(os.path.join(GOLD, "glsl_converged_small.npz"))
    hdr = z["a"].copy()
    rng = np.random.default_rng(3)
    ys, xs = rng.integers(3, hdr.shape[0] - 3, 12), rng.integers(3, hdr.shape[1] - 3, 12)
    hdr[ys, xs, :3] *= rng.uniform(20, 200, size=(12, 1)).astype(np.float32)
    g = G.GlslRef()
    out = {"hdr": hdr}
    for i, (exp, sat, den, sig) in enumerate([(1.0, 1.0, False, 3.0), (2.5, 0.6, False, 3.0), (1.0, 1.0, True, 3.0),
                                              (0.7, 1.3, True, 1.5)]):
        out[f"rgba{i}"] = g.draw(hdr, exp, sat, den, sig)
        out[f"params{i}"] = np.float32([exp, sat, float(den), sig])
    # draw.fs's `scale` uniform (resScale = 0.25 while the camera moves, main.js:819,840)
    for i, (exp, sat, den, sig, scale) in enumerate([(1.0, 1.0, False, 3.0, 0.25), (1.4, 0.9, True, 2.0, 0.25)]):
        out[f"scaled_rgba{i}"] = g.draw(hdr, exp, sat, den, sig, scale)
        out[f"scaled_params{i}"] = np.float32([exp, sat, float(den), sig, scale])
    np.savez_compressed(os.path.join(GOLD, "glsl_draw.npz"), renderer=g.renderer, **out)
    print("draw goldens", out["rgba0"].shape, out["rgba0"][..., :3].mean())


def make_textured():
    """D3 on a scene with image maps (atlas res 16): the GLSL's bilinear RGBA8 texture() against the oracle's."""
    import glsl_ref as G
    import oracle as O
    arrays = S.textured_test_scene()
    cam = dict(S.BUNNY_CAMERA)
    lens = S.lens_features(cam["focal_depth"], cam["aperture"])
    g = G.GlslRef()
    g.scene(arrays)
    g._env_theta = cam["env_theta"]
    W, H = 64, 40
    g.target(W, H, replicate=True)
    pos, d = O.camera(W, H, cam["P"], cam["I"], cam["fov_scale"], lens, 4321.0)
    g.set_camera(pos, d)
    out = dict(W=W, H=H, rays_pos=pos, rays_dir=d, env_theta=np.float32(cam["env_theta"]), atlas_res=arrays.atlas_res,
               atlas_layers=arrays.atlas_layers)
    hit = probe(g, HIT)
    out["hit_t"], out["hit_index"] = hit[..., 0], hit[..., 1].astype(np.int32)
    for sel in (2, 3, 4, 5, 7):
        out[f"shade{sel}"] = probe(g, sel)
    np.savez_compressed(os.path.join(GOLD, "glsl_stages_textured.npz"), **out)
    print("textured stages", arrays.atlas_res, arrays.atlas_layers, (out["hit_index"] >= 0).mean())


def make_atlas():
    """Reference atlas writer (texture_packer.js WebGLTextureWriter) on synthetic source images."""
    import glsl_ref as G
    g = G.GlslRef()
    rng = np.random.default_rng(11)
    out = {}
    cases = [(37, 23, 16, False, (0, 1, 2, 3)), (64, 64, 32, True, (0, 1, 2, 3)), (20, 48, 32, False, (2, 1, 0, 3)),
             (33, 17, 8, True, (1, 0, 2, 3))]
    for i, (w, h, res, corrected, swz) in enumerate(cases):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 255 // max(1, w - 1)), (yy * 255 // max(1, h - 1)), rng.integers(0, 256, (h, w)),
                        np.where((xx + yy) % 5 == 0, 128, 255)], -1).astype(np.uint8)
        out[f"src{i}"] = img
        out[f"params{i}"] = np.int32([res, int(corrected), *swz])
        out[f"dst{i}"] = g.write_texture(img, res, corrected, swz)
    np.savez_compressed(os.path.join(GOLD, "glsl_atlas_writer.npz"), renderer=g.renderer, **out)
    print("atlas goldens", [out[f"dst{i}"].shape for i in range(len(cases))])


if __name__ == "__main__":
    import subprocess
    what = sys.argv[1:] or ["js", "glsl", "converged"]
    os.makedirs(GOLD, exist_ok=True)
    for w in what:
        if w == "js":
            make_js()
        elif w == "js1m":
            make_js_1m()
        elif w == "glsl":
            make_glsl()
        elif w.startswith("glsl:"):
            make_glsl(w.split(":", 1)[1])
        elif w == "converged":
            for name in CONVERGED:
                subprocess.check_call([sys.executable, "-u", os.path.abspath(__file__), "converged:" + name])
        elif w == "camera_replay":
            make_camera_replay()
        elif w == "textured":
            make_textured()
        elif w == "samplers":
            for name in ("small", "variant", "textured"):
                subprocess.check_call([sys.executable, "-u", os.path.abspath(__file__), "samplers:" + name])
        elif w.startswith("samplers:"):
            make_samplers(w.split(":", 1)[1])
        elif w == "path_replay":
            for name in ("small", "variant", "textured"):
                subprocess.check_call([sys.executable, "-u", os.path.abspath(__file__), "path_replay:" + name])
        elif w.startswith("path_replay:"):
            make_path_replay(w.split(":", 1)[1])
        elif w == "atlas":
            make_atlas()
        elif w == "bvhtest":
            make_bvh_test()
        elif w == "fuzzbvh":
            make_fuzz_bvh_test()
        elif w == "draw":
            make_draw()
        elif w.startswith("converged:"):
            make_converged(w.split(":", 1)[1])
