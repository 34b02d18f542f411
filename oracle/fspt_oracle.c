/*
 * fspt_oracle.c — TEST INFRASTRUCTURE.  CPU restatement of the reference's
 * path-trace hot path, in plain C, on the REFERENCE's data layout.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the product (fspt_amd/, libfspt.so) never does.
 *
 * Follows, function by function (cites are /root/reference/shader/...):
 *   camera.fs:19-46     rnd, getScreen, getAA, getDOF, main
 *   tracer.fs:100-179   indexToCoords/create* accessors (flat arrays here)
 *   tracer.fs:181       rnd
 *   tracer.fs:194-298   misWeights ... evalLambert
 *   tracer.fs:300-326   rayTriangleIntersect, rayBoxIntersect
 *   tracer.fs:328-353   barycentric*
 *   tracer.fs:355-404   processLeaf, intersectScene (explicit stack[64])
 *   tracer.fs:410-434   envColor, envSample, sampleEnv
 *   tracer.fs:436-518   main (bounce loop, running-mean accumulate)
 * Arithmetic: oracle_math.h ("fspt-math" spec).  Texture fetches follow the
 * GL ES 3.0 sampler state the reference sets (main.js:170-180, 548-559,
 * 570-574): data textures NEAREST (flat arrays), atlas bilinear REPEAT,
 * environment bilinear on the RGBE bytes with S=REPEAT, T=CLAMP_TO_EDGE.
 *
 * Parity status: PINNED against the reference GLSL executed on SwiftShader in
 * the build container (tests/golden/glsl_*.npz, made by tools/make_glsl_goldens.py)
 * — staged as SURVEY.md App. D (primary hits, first-hit shading inputs,
 * converged means); see DESIGN.md §Parity.
 *
 * Documented deviations from the GLSL:
 *   - refraction's unbounded `i--` (tracer.fs:488) is capped at
 *     ORACLE_MAX_PATH_ITERS loop iterations;
 *   - radianceBins index clamped to ENV_BINS-1 (tracer.fs:423-424);
 *   - triangle fetches past the end of triTex return the padBuffer() fill
 *     value -1 (main.js:150-152);
 *   - asin input clamped to [-1,1].
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle_math.h"

#define MAX_T 100000.0f
#define EPSILON 0.000001f
#define M_PI_F 3.14159265f
#define M_TAU_F (M_PI_F * 2.0f)
#define INV_PI_F (1.0f / M_PI_F)
#define ORACLE_MAX_PATH_ITERS 64

typedef struct { float x, y, z; } vec3;
typedef struct { float x, y; } vec2;

typedef struct {
  const float *bvh; uint32_t n_nodes;
  const float *tri; uint32_t n_tris;
  const float *mat; const float *norm; const float *uv;
  const uint8_t *atlas; uint32_t atlas_res, atlas_layers;
  const uint8_t *env; uint32_t env_w, env_h;
  const uint32_t *bins; uint32_t n_bins;
  uint32_t leaf_size;
} oracle_scene;

typedef struct {
  uint64_t samples, rays, steps, leaves, shades, env_lookups;
} oracle_counters;

/* ---- vec helpers (explicit fma placement = the spec) ------------------- */
static inline vec3 v3(float x, float y, float z) { vec3 v = {x, y, z}; return v; }
static inline vec3 v_add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v_sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 v_mul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 v_scale(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 v_neg(vec3 a) { return v3(-a.x, -a.y, -a.z); }
/* a*s + b, fused */
static inline vec3 v_fma(vec3 a, float s, vec3 b) {
  return v3(om_fma(a.x, s, b.x), om_fma(a.y, s, b.y), om_fma(a.z, s, b.z));
}
static inline float v_dot(vec3 a, vec3 b) {
  return om_fma(a.z, b.z, om_fma(a.y, b.y, a.x * b.x));
}
static inline vec3 v_cross(vec3 a, vec3 b) {
  return v3(om_fma(a.y, b.z, -(a.z * b.y)), om_fma(a.z, b.x, -(a.x * b.z)),
            om_fma(a.x, b.y, -(a.y * b.x)));
}
static inline vec3 v_normalize(vec3 a) {
  float inv = 1.0f / sqrtf(v_dot(a, a));
  return v_scale(a, inv);
}
/* w.x*a + w.y*b + w.z*c */
static inline vec3 v_bary(vec3 w, vec3 a, vec3 b, vec3 c) {
  return v3(om_fma(w.z, c.x, om_fma(w.y, b.x, w.x * a.x)),
            om_fma(w.z, c.y, om_fma(w.y, b.y, w.x * a.y)),
            om_fma(w.z, c.z, om_fma(w.y, b.z, w.x * a.z)));
}
static inline float f_lerp(float x, float y, float a) { return om_fma(a, y - x, x); }

/* ---- rnd (tracer.fs:181, camera.fs:19) --------------------------------- */
static inline float rnd_seed(float *seed) {
  *seed = *seed + 0.211324865405187f;
  return om_fract(om_sin(*seed) * 43758.5453123f);
}
/* The tracer's `seed` global.  `rec` (probes only): the values the reference GLSL's own rnd()
 * returned for the same calls, replayed in call order - takes the GLSL implementation's sin()
 * out of a comparison of everything downstream of rnd(). */
typedef struct { float seed; const float *rec; uint32_t used; uint32_t limit; /* values in rec (0: unbounded) */ } rng_t;
static inline float rnd(rng_t *g) {
  if (g->rec) {
    if (g->limit && g->used >= g->limit) { g->used++; return 0.5f; } /* past the recording: the caller sees used > limit */
    return g->rec[g->used++];
  }
  return rnd_seed(&g->seed);
}

/* ---- data accessors (tracer.fs:105-179) -------------------------------- */
static inline int32_t node_word(const oracle_scene *s, int node, int w) {
  return (int32_t)om_f2bits(s->bvh[(size_t)node * 9 + w]);
}
static inline void fetch_tri(const oracle_scene *s, int index, vec3 *a, vec3 *b, vec3 *c) {
  if (index < 0 || (uint32_t)index >= s->n_tris) { /* padBuffer fill, main.js:150-152 */
    *a = *b = *c = v3(-1.0f, -1.0f, -1.0f);
    return;
  }
  const float *p = s->tri + (size_t)index * 9;
  *a = v3(p[0], p[1], p[2]); *b = v3(p[3], p[4], p[5]); *c = v3(p[6], p[7], p[8]);
}

/* ---- rayTriangleIntersect (tracer.fs:300-315) --------------------------- */
static inline float ray_tri(vec3 o, vec3 d, vec3 v1, vec3 v2, vec3 v3_) {
  vec3 e1 = v_sub(v2, v1);
  vec3 e2 = v_sub(v3_, v1);
  vec3 p = v_cross(d, e2);
  float det = v_dot(e1, p);
  if (om_abs(det) < EPSILON) return MAX_T;
  float invDet = 1.0f / det;
  vec3 t = v_sub(o, v1);
  float u = v_dot(t, p) * invDet;
  if (u < 0.0f || u > 1.0f) return MAX_T;
  vec3 q = v_cross(t, e1);
  float v = v_dot(d, q) * invDet;
  if (v < 0.0f || u + v > 1.0f) return MAX_T;
  float dist = v_dot(e2, q) * invDet;
  return dist > EPSILON ? dist : MAX_T;
}

/* ---- rayBoxIntersect (tracer.fs:317-326) -------------------------------- */
static inline float ray_box(const float *bmin, const float *bmax, vec3 o, vec3 inv) {
  float t1x = (bmin[0] - o.x) * inv.x, t2x = (bmax[0] - o.x) * inv.x;
  float t1y = (bmin[1] - o.y) * inv.y, t2y = (bmax[1] - o.y) * inv.y;
  float t1z = (bmin[2] - o.z) * inv.z, t2z = (bmax[2] - o.z) * inv.z;
  float tMax = om_min(om_min(om_max(t1x, t2x), om_max(t1y, t2y)), om_max(t1z, t2z));
  float tMin = om_max(om_max(om_min(t1x, t2x), om_min(t1y, t2y)), om_min(t1z, t2z));
  return (tMax >= tMin && tMax > 0.0f) ? tMin : MAX_T;
}

/* ---- intersectScene (tracer.fs:366-404) + processLeaf (355-364) --------- */
typedef struct { float t; int index; } hit_t;

static hit_t intersect_scene(const oracle_scene *s, vec3 o, vec3 d, oracle_counters *c,
                             uint32_t *steps_out, uint32_t *leaves_out) {
  hit_t result = {MAX_T, -1};
  int stack[64];
  int ptr = 0;
  stack[ptr++] = -1;
  int idx = 0;
  uint32_t steps = 0, leaves = 0;
  /* tracer.fs:318 recomputes 1/dir in every box test; same value hoisted */
  vec3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  while (idx > -1) {
    steps++;
    int left = node_word(s, idx, 0);
    int right = node_word(s, idx, 1);
    int tris = node_word(s, idx, 2);
    if (tris > -1) {
      leaves++;
      for (uint32_t i = 0; i < s->leaf_size; ++i) {
        vec3 a, b, cc;
        fetch_tri(s, tris + (int)i, &a, &b, &cc);
        float res = ray_tri(o, d, a, b, cc);
        if (res < result.t) { result.index = tris + (int)i; result.t = res; }
      }
    } else {
      float leftHit = ray_box(s->bvh + (size_t)left * 9 + 3, s->bvh + (size_t)left * 9 + 6, o, inv);
      float rightHit = ray_box(s->bvh + (size_t)right * 9 + 3, s->bvh + (size_t)right * 9 + 6, o, inv);
      if (leftHit < result.t && rightHit < result.t) {
        int deferred;
        if (leftHit > rightHit) { idx = right; deferred = left; }
        else { idx = left; deferred = right; }
        if (ptr < 64) stack[ptr++] = deferred;
        continue;
      } else if (leftHit < result.t) { idx = left; continue; }
      else if (rightHit < result.t) { idx = right; continue; }
    }
    idx = stack[--ptr];
  }
  if (c) { c->rays++; c->steps += steps; c->leaves += leaves; }
  if (steps_out) *steps_out = steps;
  if (leaves_out) *leaves_out = leaves;
  return result;
}

/* ---- texture fetches ---------------------------------------------------- */
static inline int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
static inline int wrap_clamp(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }
static inline float safe_floor_coord(float u) {
  float f = om_floor(u);
  if (!(f > -1.0e9f && f < 1.0e9f)) f = 0.0f;
  return f;
}
/* bilinear RGBA8, returns 4 channels in [0,1] */
static void tex_bilinear(const uint8_t *texels, int w, int h, float s, float t,
                         int repeat_t, float out[4]) {
  float u = om_fma(s, (float)w, -0.5f), v = om_fma(t, (float)h, -0.5f);
  float fu = safe_floor_coord(u), fv = safe_floor_coord(v);
  float a = u - fu, b = v - fv;
  if (!(a >= 0.0f && a <= 1.0f)) a = 0.0f;
  if (!(b >= 0.0f && b <= 1.0f)) b = 0.0f;
  int i0 = (int)fu, j0 = (int)fv;
  int i1 = wrap_repeat(i0 + 1, w); i0 = wrap_repeat(i0, w);
  int j1, j0w;
  if (repeat_t) { j1 = wrap_repeat(j0 + 1, h); j0w = wrap_repeat(j0, h); }
  else { j1 = wrap_clamp(j0 + 1, h); j0w = wrap_clamp(j0, h); }
  const uint8_t *p00 = texels + ((size_t)j0w * w + i0) * 4;
  const uint8_t *p10 = texels + ((size_t)j0w * w + i1) * 4;
  const uint8_t *p01 = texels + ((size_t)j1 * w + i0) * 4;
  const uint8_t *p11 = texels + ((size_t)j1 * w + i1) * 4;
  for (int ch = 0; ch < 4; ++ch) {
    float t00 = (float)p00[ch] / 255.0f, t10 = (float)p10[ch] / 255.0f;
    float t01 = (float)p01[ch] / 255.0f, t11 = (float)p11[ch] / 255.0f;
    out[ch] = f_lerp(f_lerp(t00, t10, a), f_lerp(t01, t11, a), b);
  }
}
/* texture(texArray, vec3(uv, layer)) (tracer.fs:453-456; sampler main.js:548-555) */
static void atlas_fetch(const oracle_scene *s, vec2 uv, float layer, float out[4]) {
  int l = (int)om_floor(layer + 0.5f);
  if (l < 0) l = 0;
  if (l > (int)s->atlas_layers - 1) l = (int)s->atlas_layers - 1;
  const uint8_t *base = s->atlas + (size_t)l * s->atlas_res * s->atlas_res * 4;
  tex_bilinear(base, (int)s->atlas_res, (int)s->atlas_res, uv.x, uv.y, 1, out);
}
/* envColor (tracer.fs:410-414) */
static vec3 env_color(const oracle_scene *s, float cx, float cy) {
  if (!s->env) return v3(0.0f, 0.0f, 0.0f); /* black default env, main.js:303-307 */
  float rgbe[4];
  tex_bilinear(s->env, (int)s->env_w, (int)s->env_h, cx, cy, 0, rgbe);
  float sc = om_exp2(om_fma(rgbe[3], 255.0f, -128.0f));
  return v3(rgbe[0] * sc, rgbe[1] * sc, rgbe[2] * sc);
}
/* envSample (tracer.fs:416-419) */
static vec3 env_sample(const oracle_scene *s, vec3 dir, float envTheta, oracle_counters *c) {
  if (c) c->env_lookups++;
  float cx = envTheta + om_atan2(dir.z, dir.x) / M_TAU_F;
  float cy = om_fma(om_asin(-dir.y), INV_PI_F, 0.5f);
  return env_color(s, cx, cy);
}
/* sampleEnv (tracer.fs:421-434) */
static void sample_env(const oracle_scene *s, float envTheta, rng_t *seed, vec3 *dir, float *pdf) {
  float nb = (float)s->n_bins;
  int idx = (int)(nb * rnd(seed));
  if (idx > (int)s->n_bins - 1) idx = (int)s->n_bins - 1;
  if (idx < 0) idx = 0;
  const uint32_t *b = s->bins + (size_t)idx * 4;
  float bx = (float)b[0], by = (float)b[1], bz = (float)b[2], bw = (float)b[3];
  float dx = (float)s->env_w, dy = (float)s->env_h;
  if (!s->env) { dx = 1.0f; dy = 2048.0f; } /* createEnvironmentMapPixels, main.js:183 */
  float r1 = rnd(seed);
  float r2 = rnd(seed);
  float ux = -envTheta + om_fma(bz - bx, r1, bx) / dx;
  float uy = 0.0f + om_fma(bw - by, r2, by) / dy;
  float theta = ux * M_TAU_F;
  float phi = uy * M_PI_F;
  float sinPhi = om_sin(phi);
  *dir = v3(om_cos(theta) * sinPhi, om_cos(phi), om_sin(theta) * sinPhi);
  float nominal = (dx * dy) / nb;
  *pdf = nominal / (((((bz - bx) * (bw - by)) * M_TAU_F) * M_PI_F) * sinPhi);
}

/* ---- BRDF helpers (tracer.fs:194-298) ----------------------------------- */
static inline vec2 mis_weights(float a, float b) {
  vec2 r;
  if (a > EPSILON && b > EPSILON) {
    float a2 = a * a, b2 = b * b, s = a2 + b2;
    r.x = a2 / s; r.y = b2 / s;
  } else { r.x = 1.0f; r.y = 0.0f; }
  return r;
}
static inline float gtr2(float ndh, float a) {
  float a2 = a * a;
  float t = om_fma((a2 - 1.0f) * ndh, ndh, 1.0f);
  return a2 / ((M_PI_F * t) * t);
}
static inline float smithG(float ndv, float alphaG) {
  float a = alphaG * alphaG, b = ndv * ndv;
  return 1.0f / (ndv + sqrtf(om_fma(-a, b, a + b)));
}
static inline float gtr2_pdf(vec3 incident, vec3 normal, float rough, vec3 bsdfDir) {
  float alpha = om_max(0.001f, rough);
  vec3 h = v_normalize(v_add(bsdfDir, incident));
  float cosTheta = om_abs(v_dot(h, normal));
  float pdf = gtr2(cosTheta, alpha) * cosTheta;
  return pdf / (4.0f * om_abs(v_dot(bsdfDir, h)));
}
static inline float schlick(vec3 incident, vec3 normal, float nx, float ny) {
  float r0 = (nx - ny) / (nx + ny);
  r0 *= r0;
  float cosTheta = v_dot(normal, incident);
  if (nx > ny) {
    float n = nx / ny;
    float sinTheta2 = (n * n) * om_fma(-cosTheta, cosTheta, 1.0f);
    if (sinTheta2 > 1.0f) return 1.0f;
    cosTheta = sqrtf(1.0f - sinTheta2);
  }
  float x = 1.0f - cosTheta;
  float q = ((((1.0f - r0) * x) * x) * x) * x;
  return om_fma(q, x, r0);
}
static inline void local_frame(vec3 n, vec3 *tangent, vec3 *bitangent) {
  vec3 up = (om_abs(n.z) < 0.999f) ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
  *tangent = v_normalize(v_cross(up, n));
  *bitangent = v_cross(n, *tangent);
}
static inline vec3 frame_combine(vec3 t, vec3 b, vec3 n, vec3 h) {
  return v3(om_fma(n.x, h.z, om_fma(b.x, h.y, t.x * h.x)),
            om_fma(n.y, h.z, om_fma(b.y, h.y, t.y * h.x)),
            om_fma(n.z, h.z, om_fma(b.z, h.y, t.z * h.x)));
}
static vec3 sample_microfacet(vec3 normal, float rough, rng_t *seed) {
  float r1 = rnd(seed), r2 = rnd(seed);
  vec3 t, b; local_frame(normal, &t, &b);
  float a = om_max(0.001f, rough);
  float phi = r1 * M_TAU_F;
  float cosTheta = sqrtf((1.0f - r2) / om_fma(om_fma(a, a, -1.0f), r2, 1.0f));
  float sinTheta = om_clamp(sqrtf(om_fma(-cosTheta, cosTheta, 1.0f)), 0.0f, 1.0f);
  float sinPhi = om_sin(phi), cosPhi = om_cos(phi);
  vec3 h = v3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
  return frame_combine(t, b, normal, h);
}
static vec3 sample_lambert(vec3 normal, rng_t *seed) {
  float r1 = rnd(seed), r2 = rnd(seed);
  vec3 t, b; local_frame(normal, &t, &b);
  float r = sqrtf(r1);
  float phi = M_TAU_F * r2;
  vec3 d;
  d.x = r * om_cos(phi);
  d.y = r * om_sin(phi);
  d.z = sqrtf(om_max(0.0f, om_fma(-d.y, d.y, om_fma(-d.x, d.x, 1.0f))));
  return frame_combine(t, b, normal, d);
}
static vec3 eval_specular(vec3 incident, vec3 normal, vec3 diffuse, float metallic, float rough,
                          vec3 bsdfDir) {
  float ndl = v_dot(normal, bsdfDir);
  float ndv = v_dot(normal, incident);
  vec3 H = v_normalize(v_add(bsdfDir, incident));
  float ndh = v_dot(normal, H);
  float a = om_max(0.001f, rough);
  float Ds = gtr2(ndh, a);
  float om = 1.0f - metallic;
  vec3 Fs = v3(om_fma(diffuse.x, metallic, om), om_fma(diffuse.y, metallic, om),
               om_fma(diffuse.z, metallic, om));
  float roughg = om_fma(rough, 0.5f, 0.5f);
  roughg = roughg * roughg;
  float Gs = smithG(ndl, roughg) * smithG(ndv, roughg);
  return v3((Gs * Fs.x) * Ds, (Gs * Fs.y) * Ds, (Gs * Fs.z) * Ds);
}

/* ---- barycentricWeights (tracer.fs:339-353) ----------------------------- */
static inline vec3 bary_weights(vec3 a, vec3 b, vec3 c, vec3 p) {
  vec3 v0 = v_sub(b, a), v1 = v_sub(c, a), v2 = v_sub(p, a);
  float d00 = v_dot(v0, v0), d01 = v_dot(v0, v1), d11 = v_dot(v1, v1);
  float d20 = v_dot(v2, v0), d21 = v_dot(v2, v1);
  float invDenom = 1.0f / om_fma(d00, d11, -(d01 * d01));
  float v = om_fma(d11, d20, -(d01 * d21)) * invDenom;
  float w = om_fma(d00, d21, -(d01 * d20)) * invDenom;
  float u = (1.0f - v) - w;
  return v3(u, v, w);
}

/* Per-pixel debug record of the first shading event (parity stage D3). */
typedef struct {
  float t; int32_t index;
  float origin[3]; float bary[3]; float uv[2];
  float diffuse[3]; float emissive[3]; float mr[2]; float tex_normal[3];
  float macro_normal[3]; float bary_normal[3];
} oracle_first_hit;

/* ---- one iteration of the bounce loop, tracer.fs:447-499 ------------------
 * From `Material mat = createMaterial(result.index)` to `vec2 weights = misWeights(...)`: everything
 * between two intersectScene calls that does not touch `color` / `accumulatedReflectance`.  Shared by
 * trace_pixel and by oracle_bounce_probe (the probe replays the GLSL's recorded rnd() values). */
typedef struct {
  vec3 origin;             /* hit point (tracer.fs:450) */
  vec3 ro, rd;             /* the extension ray (ray.origin / ray.dir after the body) */
  vec3 texDiffuse, texEmissive;
  vec3 macroNormal, microNormal;
  vec3 envDir; float envPdf, cosEnv;
  vec3 bsdfThroughput, envThroughput;
  float bsdfPdf; vec2 weights;
  float dielectric;
  float seed0;             /* tracer.fs:458 */
  int inside, specular, refracted;
} bounce_t;

static inline void bounce_body(const oracle_scene *s, vec3 ro, vec3 rd, hit_t result, float randBase,
                               float envTheta, rng_t *g, oracle_first_hit *fh, const float *tex, bounce_t *o) {
  int ti = result.index;
  const float *m = s->mat + (size_t)ti * 12;
  float layDiffuse = m[0], laySpec = m[1], layNormal = m[2], layRough = m[3];
  float ior = m[9], dielectric = m[10];
  vec3 a, b, cc;
  fetch_tri(s, ti, &a, &b, &cc);
  const float *tuv = s->uv + (size_t)ti * 6;
  vec3 origin = v_fma(rd, result.t, ro);
  vec3 w = bary_weights(a, b, cc, origin);
  vec2 tc;
  tc.x = om_fma(w.z, tuv[4], om_fma(w.y, tuv[2], w.x * tuv[0]));
  tc.y = om_fma(w.z, tuv[5], om_fma(w.y, tuv[3], w.x * tuv[1]));
  float td[4], te[4], tm[4], tn[4];
  atlas_fetch(s, tc, layDiffuse, td);
  atlas_fetch(s, tc, laySpec, te);
  atlas_fetch(s, tc, layRough, tm);
  atlas_fetch(s, tc, layNormal, tn);
  if (tex) { /* probes only: the four texture() results as the GLSL's sampler returned them */
    td[0] = tex[0]; td[1] = tex[1]; td[2] = tex[2]; te[0] = tex[3]; te[1] = tex[4]; te[2] = tex[5];
    tm[0] = tex[6]; tm[1] = tex[7]; tn[0] = tex[8]; tn[1] = tex[9]; tn[2] = tex[10];
  }
  vec3 texDiffuse = v3(td[0], td[1], td[2]);
  vec3 texEmissive = v3(te[0], te[1], te[2]);
  float metallic = tm[0], rough = tm[1];
  vec3 texNormal = v3((tn[0] - 0.5f) * 2.0f, (tn[1] - 0.5f) * 2.0f, (tn[2] - 0.0f) * 1.0f);
  rough = rough * rough;
  g->seed = om_fma(origin.z, 4761.52835f, ((origin.x * randBase) * origin.y) * 1.396529836f);
  o->seed0 = g->seed;
  const float *nn = s->norm + (size_t)ti * 27;
  vec3 n1 = v3(nn[0], nn[1], nn[2]), t1 = v3(nn[3], nn[4], nn[5]), b1 = v3(nn[6], nn[7], nn[8]);
  vec3 n2 = v3(nn[9], nn[10], nn[11]), t2 = v3(nn[12], nn[13], nn[14]), b2 = v3(nn[15], nn[16], nn[17]);
  vec3 n3 = v3(nn[18], nn[19], nn[20]), t3 = v3(nn[21], nn[22], nn[23]), b3 = v3(nn[24], nn[25], nn[26]);
  vec3 baryNormal = v_bary(w, n1, n2, n3);
  vec3 baryTangent = v_bary(w, t1, t2, t3);
  vec3 baryBitangent = v_bary(w, b1, b2, b3);
  vec3 macroNormal = v_normalize(
      v3(om_fma(texNormal.z, baryNormal.x, om_fma(texNormal.y, baryBitangent.x, texNormal.x * baryTangent.x)),
         om_fma(texNormal.z, baryNormal.y, om_fma(texNormal.y, baryBitangent.y, texNormal.x * baryTangent.y)),
         om_fma(texNormal.z, baryNormal.z, om_fma(texNormal.y, baryBitangent.z, texNormal.x * baryTangent.z))));
  if (fh) {
    fh->origin[0] = origin.x; fh->origin[1] = origin.y; fh->origin[2] = origin.z;
    fh->bary[0] = w.x; fh->bary[1] = w.y; fh->bary[2] = w.z;
    fh->uv[0] = tc.x; fh->uv[1] = tc.y;
    fh->diffuse[0] = td[0]; fh->diffuse[1] = td[1]; fh->diffuse[2] = td[2];
    fh->emissive[0] = te[0]; fh->emissive[1] = te[1]; fh->emissive[2] = te[2];
    fh->mr[0] = tm[0]; fh->mr[1] = tm[1];
    fh->tex_normal[0] = texNormal.x; fh->tex_normal[1] = texNormal.y; fh->tex_normal[2] = texNormal.z;
    fh->macro_normal[0] = macroNormal.x; fh->macro_normal[1] = macroNormal.y; fh->macro_normal[2] = macroNormal.z;
    fh->bary_normal[0] = baryNormal.x; fh->bary_normal[1] = baryNormal.y; fh->bary_normal[2] = baryNormal.z;
  }
  int inside = v_dot(v_neg(rd), baryNormal) < 0.0f;
  float nsx = inside ? ior : 1.0f, nsy = inside ? 1.0f : ior;
  if (inside) macroNormal = v_neg(macroNormal);
  vec3 off = v_scale(v_scale(macroNormal, EPSILON), 2.0f);
  ro = v_add(origin, off);

  vec3 incident = v_neg(rd);
  vec3 envThroughput, bsdfThroughput;
  float bsdfPdf;
  vec3 microNormal = sample_microfacet(macroNormal, rough, g);
  vec3 envDir; float envPdf;
  sample_env(s, envTheta, g, &envDir, &envPdf);
  float cosEnv = v_dot(macroNormal, envDir);
  float F = schlick(incident, microNormal, nsx, nsy);
  int specular = om_fma(1.0f, metallic, F * (1.0f - metallic)) > rnd(g);
  int refracted = 0;
  if (specular) {
    /* reflect(-incident, microNormal) = I - 2*dot(N,I)*N */
    vec3 I = v_neg(incident);
    float k = 2.0f * v_dot(microNormal, I);
    rd = v3(om_fma(-k, microNormal.x, I.x), om_fma(-k, microNormal.y, I.y), om_fma(-k, microNormal.z, I.z));
    bsdfPdf = gtr2_pdf(incident, macroNormal, rough, rd);
    vec3 es = eval_specular(incident, macroNormal, texDiffuse, metallic, rough, rd);
    float cl = om_clamp(v_dot(macroNormal, rd), 0.0f, 1.0f);
    bsdfThroughput = v3((es.x * cl) / bsdfPdf, (es.y * cl) / bsdfPdf, (es.z * cl) / bsdfPdf);
    vec3 ee = eval_specular(incident, macroNormal, texDiffuse, metallic, rough, envDir);
    float ce = om_clamp(cosEnv, 0.0f, 1.0f);
    envThroughput = v3((ee.x * ce) / envPdf, (ee.y * ce) / envPdf, (ee.z * ce) / envPdf);
  } else if (dielectric >= 0.0f) {
    bsdfPdf = 1.0f;
    bsdfThroughput = v3(1.0f, 1.0f, 1.0f);
    envThroughput = v3(0.0f, 0.0f, 0.0f);
    ro = v_sub(origin, off);
    /* refract(-incident, microNormal, ns.x/ns.y) */
    vec3 I = v_neg(incident);
    float eta = nsx / nsy;
    float dNI = v_dot(microNormal, I);
    float kk = 1.0f - (eta * eta) * (1.0f - dNI * dNI);
    if (kk < 0.0f) rd = v3(0.0f, 0.0f, 0.0f);
    else {
      float sc = om_fma(eta, dNI, sqrtf(kk));
      rd = v3(om_fma(eta, I.x, -(sc * microNormal.x)), om_fma(eta, I.y, -(sc * microNormal.y)),
              om_fma(eta, I.z, -(sc * microNormal.z)));
    }
    refracted = 1; /* i--, tracer.fs:488 */
  } else {
    rd = sample_lambert(macroNormal, g);
    bsdfPdf = om_abs(v_dot(rd, macroNormal)) * INV_PI_F;
    float cl = om_clamp(v_dot(macroNormal, rd), 0.0f, 1.0f);
    bsdfThroughput = v3(((texDiffuse.x * INV_PI_F) * cl) / bsdfPdf, ((texDiffuse.y * INV_PI_F) * cl) / bsdfPdf,
                        ((texDiffuse.z * INV_PI_F) * cl) / bsdfPdf);
    float ce = om_clamp(cosEnv, 0.0f, 1.0f);
    envThroughput = v3(((texDiffuse.x * INV_PI_F) * ce) / envPdf, ((texDiffuse.y * INV_PI_F) * ce) / envPdf,
                       ((texDiffuse.z * INV_PI_F) * ce) / envPdf);
  }
  if (inside) { /* tracer.fs:497 */
    bsdfThroughput = v3(om_max(1.0f - (((1.0f - texDiffuse.x) * result.t) * dielectric), 0.0f),
                        om_max(1.0f - (((1.0f - texDiffuse.y) * result.t) * dielectric), 0.0f),
                        om_max(1.0f - (((1.0f - texDiffuse.z) * result.t) * dielectric), 0.0f));
  }
  o->origin = origin; o->ro = ro; o->rd = rd;
  o->texDiffuse = texDiffuse; o->texEmissive = texEmissive;
  o->macroNormal = macroNormal; o->microNormal = microNormal;
  o->envDir = envDir; o->envPdf = envPdf; o->cosEnv = cosEnv;
  o->bsdfThroughput = bsdfThroughput; o->envThroughput = envThroughput;
  o->bsdfPdf = bsdfPdf; o->weights = mis_weights(envPdf, bsdfPdf);
  o->dielectric = dielectric;
  o->inside = inside; o->specular = specular; o->refracted = refracted;
}

/* ---- the rest of the iteration, tracer.fs:467 and 500-512: emission, NEE shadow ray, extension ray ------
 * Returns 1 when the extension ray left the scene (`break`). */
/* Whole-path replay only (NULL otherwise): sig[0] a hash of the hit indices every intersectScene call of the path
 * returned, in call order (h = h * 31 + index + 2, uint32 wrap-around), sig[1] the number of calls; `env` (optional):
 * the rgb the reference GLSL's envSample returned for the path's environment lookups, in call order (env_limit of
 * them), replayed in place of the oracle's own lookup - takes the GLSL implementation's RGBA8 / RGBE decode out. */
typedef struct { uint32_t sig[2]; const float *env; uint32_t env_used, env_limit;
                 /* `tex` (optional): the four texture() results of the path's k-th loop iteration (12 floats, bounce_body's layout) */
                 const float *tex; uint32_t tex_limit; } replay_t;
static inline void sig_add(replay_t *rp, int index) {
  if (rp) { rp->sig[0] = rp->sig[0] * 31u + (uint32_t)(index + 2); rp->sig[1]++; }
}
static inline vec3 env_sample_rp(const oracle_scene *s, vec3 dir, float envTheta, oracle_counters *c, replay_t *rp) {
  vec3 own = env_sample(s, dir, envTheta, c);
  if (rp) {
    const uint32_t k = rp->env_used++; /* (counted whether or not the lookups are replayed) */
    if (rp->env && k < rp->env_limit) return v3(rp->env[k * 3], rp->env[k * 3 + 1], rp->env[k * 3 + 2]);
  }
  return own;
}
static inline int bounce_tail(const oracle_scene *s, const bounce_t *b, float envTheta, oracle_counters *c,
                              vec3 *thr_io, vec3 *color_io, hit_t *result, replay_t *sig) {
  vec3 thr = *thr_io, color = *color_io;
  /* tracer.fs:467 */
  color = v3(om_fma((thr.x * b->texEmissive.x) * b->texDiffuse.x, 30.0f, color.x),
             om_fma((thr.y * b->texEmissive.y) * b->texDiffuse.y, 30.0f, color.y),
             om_fma((thr.z * b->texEmissive.z) * b->texDiffuse.z, 30.0f, color.z));
  if (b->dielectric < 0.0f && b->cosEnv > 0.0f) {
    hit_t shadow = intersect_scene(s, b->ro, b->envDir, c, NULL, NULL);
    sig_add(sig, shadow.index);
    if (shadow.index == -1) {
      vec3 es = env_sample_rp(s, b->envDir, envTheta, c, sig);
      color = v3(om_fma((thr.x * b->envThroughput.x) * es.x, b->weights.x, color.x),
                 om_fma((thr.y * b->envThroughput.y) * es.y, b->weights.x, color.y),
                 om_fma((thr.z * b->envThroughput.z) * es.z, b->weights.x, color.z));
    }
  }
  *result = intersect_scene(s, b->ro, b->rd, c, NULL, NULL);
  sig_add(sig, result->index);
  thr = v_mul(thr, b->bsdfThroughput);
  int left = 0;
  if (result->index == -1) {
    vec3 es = env_sample_rp(s, b->rd, envTheta, c, sig);
    color = v3(om_fma(thr.x * es.x, b->weights.y, color.x), om_fma(thr.y * es.y, b->weights.y, color.y),
               om_fma(thr.z * es.z, b->weights.y, color.z));
    left = 1;
  }
  *thr_io = thr; *color_io = color;
  return left;
}

/* ---- tracer.fs main (436-518) for one pixel ----------------------------- */
/* ... the path itself (tracer.fs:439-514): the sample's colour before the clamp of tracer.fs:515.  `g`: the tracer's
 * random numbers - its own sin-hash, or (whole-path replay) the values the reference GLSL's rnd() returned. */
static vec3 trace_path(const oracle_scene *s, vec3 ro, vec3 rd, float randBase, float envTheta, uint32_t numBounces,
                       rng_t *gp, oracle_counters *c, oracle_first_hit *fh, replay_t *sig) {
  rng_t g = *gp;
  hit_t result = intersect_scene(s, ro, rd, c, NULL, NULL);
  sig_add(sig, result.index);
  vec3 color = v3(0.0f, 0.0f, 0.0f);
  if (fh) { memset(fh, 0, sizeof(*fh)); fh->t = result.t; fh->index = result.index; }
  if (result.index < 0) {
    color = v_add(color, env_sample_rp(s, rd, envTheta, c, sig));
  } else {
    vec3 thr = v3(1.0f, 1.0f, 1.0f);
    int iters = 0;
    for (int i = 0; i < (int)numBounces && iters < ORACLE_MAX_PATH_ITERS; ++i, ++iters) {
      if (c) c->shades++;
      bounce_t b;
      bounce_body(s, ro, rd, result, randBase, envTheta, &g, (fh && iters == 0) ? fh : NULL,
                  (sig && sig->tex && (uint32_t)iters < sig->tex_limit) ? sig->tex + (size_t)iters * 12 : NULL, &b);
      ro = b.ro; rd = b.rd;
      if (b.refracted) i--; /* tracer.fs:488 */
      if (bounce_tail(s, &b, envTheta, c, &thr, &color, &result, sig)) break;
    }
  }
  *gp = g;
  return color;
}
static void trace_pixel(const oracle_scene *s, vec3 ro, vec3 rd, uint32_t tick, float randBase,
                        float envTheta, uint32_t numBounces, float *accum /*rgba*/,
                        oracle_counters *c, oracle_first_hit *fh) {
  rng_t g = {0.0f, NULL, 0, 0};
  if (c) c->samples++;
  vec3 color = trace_path(s, ro, rd, randBase, envTheta, numBounces, &g, c, fh, NULL);
  color = v3(om_clamp(color.x, 0.0f, 1024.0f), om_clamp(color.y, 0.0f, 1024.0f), om_clamp(color.z, 0.0f, 1024.0f));
  float ft = (float)tick;
  float den = ft + 1.0f;
  accum[0] = om_fma(accum[0], ft, color.x) / den;
  accum[1] = om_fma(accum[1], ft, color.y) / den;
  accum[2] = om_fma(accum[2], ft, color.z) / den;
  accum[3] = 1.0f;
}

/* ---- camera.fs main (37-46) for one pixel ------------------------------- */
static void camera_pixel(uint32_t x, uint32_t y, uint32_t W, uint32_t H, const float P[3], const float I[3],
                         float fovScale, const float lens[2], float randBase, const float *rec, float *pos, float *dir) {
  float fx = (float)x + 0.5f, fy = (float)y + 0.5f; /* gl_FragCoord */
  float resx = (float)W, resy = (float)H;
  /* uv = interpolated clip-space corner (camera.vs:8, main.js:601-605) */
  float uvx = om_fma(fx / resx, 2.0f, -1.0f), uvy = om_fma(fy / resy, 2.0f, -1.0f);
  rng_t g = {om_fma(fx, resy, randBase) + fy, rec, 0, 0}; /* (rec: the four values the GLSL's rnd() returned, probes only) */
  vec3 Iv = v3(I[0], I[1], I[2]), Pv = v3(P[0], P[1], P[2]);
  vec3 basisX = v_normalize(v_cross(Iv, v3(0.0f, 1.0f, 0.0f)));
  vec3 basisY = v_normalize(v_cross(basisX, Iv));
  /* getScreen (camera.fs:21-24) */
  float icx = uvx * (resx / resy), icy = uvy * 1.0f;
  vec3 screen;
  screen.x = (om_fma(icy * basisY.x, fovScale, (icx * basisX.x) * fovScale) + Iv.x) + Pv.x;
  screen.y = (om_fma(icy * basisY.y, fovScale, (icx * basisX.y) * fovScale) + Iv.y) + Pv.y;
  screen.z = (om_fma(icy * basisY.z, fovScale, (icx * basisX.z) * fovScale) + Iv.z) + Pv.z;
  /* getAA (camera.fs:26-30) */
  float theta = (rnd(&g) * M_PI_F) * 2.0f;
  float r = sqrtf(rnd(&g)) * 1.414f;
  float ct = om_cos(theta), st = om_sin(theta);
  vec3 aa;
  aa.x = (r * ((basisX.x * ct) / resx + (basisY.x * st) / resy)) * fovScale;
  aa.y = (r * ((basisX.y * ct) / resx + (basisY.y * st) / resy)) * fovScale;
  aa.z = (r * ((basisX.z * ct) / resx + (basisY.z * st) / resy)) * fovScale;
  /* getDOF (camera.fs:32-35) */
  float theta2 = (rnd(&g) * M_PI_F) * 2.0f;
  float c2 = om_cos(theta2), s2 = om_sin(theta2);
  float sq = sqrtf(rnd(&g));
  vec3 dof;
  dof.x = (om_fma(s2, basisY.x, c2 * basisX.x) * lens[1]) * sq;
  dof.y = (om_fma(s2, basisY.y, c2 * basisX.y) * lens[1]) * sq;
  dof.z = (om_fma(s2, basisY.z, c2 * basisX.z) * lens[1]) * sq;
  vec3 o = v_add(Pv, dof);
  vec3 tgt = v3(om_fma(dof.x, lens[0], screen.x + aa.x), om_fma(dof.y, lens[0], screen.y + aa.y),
                om_fma(dof.z, lens[0], screen.z + aa.z));
  vec3 d = v_normalize(v_sub(tgt, o));
  pos[0] = o.x; pos[1] = o.y; pos[2] = o.z; pos[3] = 1.0f;
  dir[0] = d.x; dir[1] = d.y; dir[2] = d.z; dir[3] = 1.0f;
}

/* ======================= exported entry points ========================== */

void oracle_camera(uint32_t W, uint32_t H, const float P[3], const float I[3], float fovScale,
                   const float lens[2], float randBase, float *pos, float *dir) {
#pragma omp parallel for schedule(static)
  for (int64_t y = 0; y < (int64_t)H; ++y)
    for (uint32_t x = 0; x < W; ++x) {
      size_t o = ((size_t)y * W + x) * 4;
      camera_pixel(x, (uint32_t)y, W, H, P, I, fovScale, lens, randBase, NULL, pos + o, dir + o);
    }
}

/* camera.fs main with the four values the reference GLSL's own rnd() returned per pixel (rec: [H][W][4], call order)
 * replayed: everything behind rnd() - getAA, getDOF, the ray - without the GLSL implementation's sin() of a large
 * argument in between. */
void oracle_camera_probe(uint32_t W, uint32_t H, const float P[3], const float I[3], float fovScale,
                         const float lens[2], const float *rec, float *pos, float *dir) {
  for (uint32_t y = 0; y < H; ++y)
    for (uint32_t x = 0; x < W; ++x) {
      size_t o = ((size_t)y * W + x) * 4;
      camera_pixel(x, y, W, H, P, I, fovScale, lens, 0.0f, rec + o, pos + o, dir + o);
    }
}

/* One tick of drawTracer (main.js:758-807): every pixel whose tile belongs to
 * this shard (tile index % n_shards == shard; n_shards = 1 -> all). */
void oracle_trace(const oracle_scene *s, uint32_t W, uint32_t H, const float *pos, const float *dir,
                  uint32_t tick, float randBase, float envTheta, uint32_t numBounces, float *accum,
                  oracle_counters *counters, oracle_first_hit *first_hits, uint32_t shard,
                  uint32_t n_shards, uint32_t tile) {
  oracle_counters total;
  memset(&total, 0, sizeof(total));
  if (n_shards == 0) n_shards = 1;
  if (tile == 0) tile = 32;
  uint32_t tiles_x = (W + tile - 1) / tile;
#pragma omp parallel
  {
    oracle_counters local;
    memset(&local, 0, sizeof(local));
#pragma omp for schedule(dynamic, 4)
    for (int64_t y = 0; y < (int64_t)H; ++y)
      for (uint32_t x = 0; x < W; ++x) {
        uint32_t tid = ((uint32_t)y / tile) * tiles_x + x / tile;
        if (tid % n_shards != shard) continue;
        size_t p = (size_t)y * W + x;
        vec3 ro = v3(pos[p * 4], pos[p * 4 + 1], pos[p * 4 + 2]);
        vec3 rd = v3(dir[p * 4], dir[p * 4 + 1], dir[p * 4 + 2]);
        trace_pixel(s, ro, rd, tick, randBase, envTheta, numBounces, accum + p * 4,
                    counters ? &local : NULL, first_hits ? first_hits + p : NULL);
      }
#pragma omp critical
    {
      total.samples += local.samples; total.rays += local.rays; total.steps += local.steps;
      total.leaves += local.leaves; total.shades += local.shades; total.env_lookups += local.env_lookups;
    }
  }
  if (counters) {
    counters->samples += total.samples; counters->rays += total.rays; counters->steps += total.steps;
    counters->leaves += total.leaves; counters->shades += total.shades;
    counters->env_lookups += total.env_lookups;
  }
}

/* bvh_test.fs main (224-232), the reference's `mode=test` replacement of tracer.fs (main.js:879-883): the
 * number of traversal-loop iterations of the camera ray (bvh_test.fs:183, same loop as tracer.fs:366-404),
 * times 0.001, in every colour channel, folded into the running mean WITHOUT the clamp of tracer.fs:516. */
void oracle_trace_test(const oracle_scene *s, uint32_t W, uint32_t H, const float *pos, const float *dir,
                       uint32_t tick, float *accum, uint32_t shard, uint32_t n_shards, uint32_t tile) {
  if (n_shards == 0) n_shards = 1;
  if (tile == 0) tile = 32;
  uint32_t tiles_x = (W + tile - 1) / tile;
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t y = 0; y < (int64_t)H; ++y)
    for (uint32_t x = 0; x < W; ++x) {
      uint32_t tid = ((uint32_t)y / tile) * tiles_x + x / tile;
      if (tid % n_shards != shard) continue;
      size_t p = (size_t)y * W + x;
      vec3 ro = v3(pos[p * 4], pos[p * 4 + 1], pos[p * 4 + 2]);
      vec3 rd = v3(dir[p * 4], dir[p * 4 + 1], dir[p * 4 + 2]);
      uint32_t st, lv;
      (void)intersect_scene(s, ro, rd, NULL, &st, &lv);
      float c = (float)st * 0.001f;
      float ft = (float)tick, den = ft + 1.0f;
      float *a = accum + p * 4;
      a[0] = om_fma(a[0], ft, c) / den;
      a[1] = om_fma(a[1], ft, c) / den;
      a[2] = om_fma(a[2], ft, c) / den;
      a[3] = 1.0f;
    }
}

void oracle_intersect(const oracle_scene *s, const float *rays, uint32_t n, float *t_out, int32_t *index_out,
                      uint32_t *steps_out, uint32_t *leaves_out) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    vec3 o = v3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]);
    vec3 d = v3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    uint32_t st, lv;
    hit_t h = intersect_scene(s, o, d, NULL, &st, &lv);
    t_out[i] = h.t; index_out[i] = h.index;
    if (steps_out) steps_out[i] = st;
    if (leaves_out) leaves_out[i] = lv;
  }
}

/* Host PRNG replacing Math.random()*10000 (main.js:748,777): xorshift64*. */
float oracle_rand_base_next(uint64_t *state) {
  uint64_t x = *state;
  x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
  *state = x;
  uint64_t r = x * 2685821657736338717ULL;
  return ((float)(r >> 40) * (1.0f / 16777216.0f)) * 10000.0f;
}

/* n_ticks x (drawCamera + drawTracer), the reference tick() loop (main.js:838-857). */
void oracle_render(const oracle_scene *s, uint32_t W, uint32_t H, const float P[3], const float I[3],
                   float fovScale, const float lens[2], float envTheta, uint32_t numBounces,
                   uint32_t first_tick, uint32_t n_ticks, uint64_t seed, float *accum,
                   oracle_counters *counters, uint32_t shard, uint32_t n_shards, uint32_t tile) {
  float *pos = (float *)malloc((size_t)W * H * 4 * sizeof(float));
  float *dir = (float *)malloc((size_t)W * H * 4 * sizeof(float));
  uint64_t st = seed;
  for (uint32_t k = 0; k < n_ticks; ++k) {
    float rb_cam = oracle_rand_base_next(&st);
    float rb_trace = oracle_rand_base_next(&st);
    oracle_camera(W, H, P, I, fovScale, lens, rb_cam, pos, dir);
    oracle_trace(s, W, H, pos, dir, first_tick + k, rb_trace, envTheta, numBounces, accum, counters, NULL,
                 shard, n_shards, tile);
  }
  free(pos); free(dir);
}

/* math primitives for bitwise comparison with the device (fspt_math_eval) */
void oracle_math_eval(int op, const float *a, const float *b, uint32_t n, float *out) {
  for (uint32_t i = 0; i < n; ++i) {
    float x = a[i], y = b ? b[i] : 0.0f;
    switch (op) {
      case 0: out[i] = om_sin(x); break;
      case 1: out[i] = om_cos(x); break;
      case 2: out[i] = om_atan2(x, y); break;
      case 3: out[i] = om_asin(x); break;
      case 4: out[i] = om_exp2(x); break;
      case 5: out[i] = x / y; break;
      case 6: out[i] = sqrtf(x); break;
      case 7: { float sd = x; out[i] = rnd_seed(&sd); break; }
      case 8: out[i] = om_fract(x); break;
      case 9: out[i] = om_log2(x); break;
      case 10: out[i] = om_pow(x, y); break;
      default: out[i] = 0.0f;
    }
  }
}

int oracle_has_fma(void) {
#if defined(__x86_64__)
  return __builtin_cpu_supports("fma") ? 1 : 0;
#else
  return 1;
#endif
}

/* Probes of the deterministic BRDF / environment helpers on caller-supplied
 * inputs (parity stage D3 against the GLSL run with the same inputs).
 * in: n x 8 floats = (a.xyz, rough, b.xyz, metallic); out: n x 4 floats.
 *   which 0: schlick(normalize(a), N, ns=(1,1.4)), schlick(.., ns=(1.4,1)), gtr2Pdf(normalize(a),N,rough^2-free,normalize(b)), 0
 *   which 1: evalSpecular(normalize(a), N, D, metallic, rough, normalize(b)).rgb, smith-free 0
 *   which 2: misWeights(a.x, a.y), lambertPdf(N, normalize(b)), 0
 *   which 3: envSample(normalize(a)) with envTheta = rough, 0
 * N = normalize(0.1,1,0.2), D = (0.8,0.6,0.4). */
void oracle_brdf_probe(const oracle_scene *s, int which, const float *in, uint32_t n, float *out) {
  vec3 N = v_normalize(v3(0.1f, 1.0f, 0.2f));
  vec3 D = v3(0.8f, 0.6f, 0.4f);
  for (uint32_t i = 0; i < n; ++i) {
    const float *p = in + (size_t)i * 8;
    vec3 a = v3(p[0], p[1], p[2]), b = v3(p[4], p[5], p[6]);
    float rough = p[3], metallic = p[7];
    float *o = out + (size_t)i * 4;
    o[0] = o[1] = o[2] = o[3] = 0.0f;
    if (which == 0) {
      vec3 an = v_normalize(a), bn = v_normalize(b);
      o[0] = schlick(an, N, 1.0f, 1.4f);
      o[1] = schlick(an, N, 1.4f, 1.0f);
      o[2] = gtr2_pdf(an, N, rough, bn);
    } else if (which == 1) {
      vec3 r = eval_specular(v_normalize(a), N, D, metallic, rough, v_normalize(b));
      o[0] = r.x; o[1] = r.y; o[2] = r.z;
    } else if (which == 2) {
      vec2 w = mis_weights(a.x, a.y);
      o[0] = w.x; o[1] = w.y;
      o[2] = om_abs(v_dot(v_normalize(b), N)) * INV_PI_F;
    } else if (which == 3) {
      vec3 r = env_sample(s, v_normalize(a), rough, NULL);
      o[0] = r.x; o[1] = r.y; o[2] = r.z;
    }
  }
}

/* Probes of the STOCHASTIC functions with the reference GLSL's own random numbers replayed
 * (tools/make_goldens.py records what tracer.fs's rnd() returned for every call; `rec` holds them
 * in call order, `rec_stride` floats per item), so that everything downstream of rnd() is compared
 * deterministically and the GLSL implementation's sin() drops out.
 *   in: n x 4 floats (normal.xyz as given - not re-normalised, tracer.fs passes macroNormal -, metallicRoughness.y)
 *   which 0: sampleMicrofacet (tracer.fs:256-270)  -> halfvector.xyz, number of rnd() calls
 *   which 1: sampleLambert    (tracer.fs:272-280)  -> dir.xyz, number of rnd() calls
 *   which 2: sampleEnv        (tracer.fs:421-434)  -> dir.xyz, pdf */
void oracle_sampler_probe(const oracle_scene *s, int which, const float *in, const float *rec,
                          uint32_t rec_stride, float envTheta, uint32_t n, float *out) {
  for (uint32_t i = 0; i < n; ++i) {
    const float *p = in + (size_t)i * 4;
    rng_t g = {0.0f, rec + (size_t)i * rec_stride, 0, 0};
    vec3 nrm = v3(p[0], p[1], p[2]);
    float *o = out + (size_t)i * 4;
    vec3 r = v3(0.0f, 0.0f, 0.0f);
    float w = 0.0f;
    if (which == 0) { r = sample_microfacet(nrm, p[3], &g); w = (float)g.used; }
    else if (which == 1) { r = sample_lambert(nrm, &g); w = (float)g.used; }
    else if (which == 2) { sample_env(s, envTheta, &g, &r, &w); }
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = w;
  }
}

/* One iteration of the bounce loop (bounce_body = tracer.fs:447-499) for caller-supplied rays and hits
 * (t, index as the GLSL's own intersectScene returned them), rnd() replayed from `rec` (8 per item; NULL:
 * the oracle's own sin-hash from the tracer.fs:458 seed) and, optionally, the four texture() results of
 * tracer.fs:453-456 replayed from `tex` (12 per item: diffuse.rgb, emissive.rgb, metallicRoughness.rg,
 * normal-map.rgb, pad; NULL: the oracle's own bilinear fetch) - the GLSL implementation's sampler precision
 * is pinned separately (stage D3).  out: n x ORACLE_BOUNCE_FLOATS floats:
 *   0 seed (tracer.fs:458)  1 inside  2 specular  3 bsdfPdf | 4-6 ray.dir  7 weights.x | 8-10 ray.origin  11 weights.y
 *   12-14 bsdfThroughput  15 cosEnv | 16-18 envThroughput  19 envDirPdf.a | 20-22 envDirPdf.xyz  23 refracted (i--)
 *   24-26 colour added by tracer.fs:467 at accumulatedReflectance 1  27 rnd() calls | 28-30 microNormal  31 0
 *   32-34 macroNormal (after the `inside` flip)  35 mat.dielectric
 *   after the rest of the iteration (tracer.fs:467,500-512; the oracle's own shadow / extension rays and environment lookups):
 *   36-38 color  39 result.index | 40-42 accumulatedReflectance  43 result.t */
#define ORACLE_BOUNCE_FLOATS 44
void oracle_bounce_probe(const oracle_scene *s, const float *rays, const float *t_in, const int32_t *index_in,
                         float randBase, float envTheta, const float *rec, const float *tex, uint32_t n,
                         float *out) {
  for (uint32_t i = 0; i < n; ++i) {
    float *o = out + (size_t)i * ORACLE_BOUNCE_FLOATS;
    for (int k = 0; k < ORACLE_BOUNCE_FLOATS; ++k) o[k] = 0.0f;
    if (index_in[i] < 0) continue;
    vec3 ro = v3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]);
    vec3 rd = v3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    hit_t h = {t_in[i], index_in[i]};
    rng_t g = {0.0f, rec ? rec + (size_t)i * 8 : NULL, 0, 0};
    bounce_t b;
    bounce_body(s, ro, rd, h, randBase, envTheta, &g, NULL, tex ? tex + (size_t)i * 12 : NULL, &b);
    o[0] = b.seed0; o[1] = (float)b.inside; o[2] = (float)b.specular; o[3] = b.bsdfPdf;
    o[4] = b.rd.x; o[5] = b.rd.y; o[6] = b.rd.z; o[7] = b.weights.x;
    o[8] = b.ro.x; o[9] = b.ro.y; o[10] = b.ro.z; o[11] = b.weights.y;
    o[12] = b.bsdfThroughput.x; o[13] = b.bsdfThroughput.y; o[14] = b.bsdfThroughput.z; o[15] = b.cosEnv;
    o[16] = b.envThroughput.x; o[17] = b.envThroughput.y; o[18] = b.envThroughput.z; o[19] = b.envPdf;
    o[20] = b.envDir.x; o[21] = b.envDir.y; o[22] = b.envDir.z; o[23] = (float)b.refracted;
    o[24] = (b.texEmissive.x * b.texDiffuse.x) * 30.0f; o[25] = (b.texEmissive.y * b.texDiffuse.y) * 30.0f;
    o[26] = (b.texEmissive.z * b.texDiffuse.z) * 30.0f; o[27] = (float)g.used;
    o[28] = b.microNormal.x; o[29] = b.microNormal.y; o[30] = b.microNormal.z;
    o[32] = b.macroNormal.x; o[33] = b.macroNormal.y; o[34] = b.macroNormal.z; o[35] = b.dielectric;
    vec3 thr = v3(1.0f, 1.0f, 1.0f), color = v3(0.0f, 0.0f, 0.0f);
    hit_t next;
    (void)bounce_tail(s, &b, envTheta, NULL, &thr, &color, &next, NULL);
    o[36] = color.x; o[37] = color.y; o[38] = color.z; o[39] = (float)next.index;
    o[40] = thr.x; o[41] = thr.y; o[42] = thr.z; o[43] = next.t;
  }
}

/* Stage D6 - the WHOLE path with the reference GLSL's random numbers replayed.  tools/make_goldens.py path_replay runs
 * the reference's tracer.fs main() (436-518) unmodified at depth `numBounces` on SwiftShader and records, per pixel,
 * every value its rnd() returned (in call order: `rec`, rec_stride floats per pixel of which rec_count[i] are valid),
 * the number of rnd() calls, and a hash + count of the hit indices its intersectScene calls returned.  Here the same
 * rays walk trace_path with those values in place of the sin-hash: out_color = the sample's colour after the clamp of
 * tracer.fs:515 (what tick 0 writes), out_used = rnd() calls made (> rec_count[i]: the path asked for more values than
 * were recorded - it took another branch somewhere), out_sig = {hash, calls} as sig_add defines them.  env_rec: also
 * replay what the GLSL's envSample returned for the path's k-th environment lookup, tex_rec: the four texture() results
 * of its k-th loop iteration (replay_t) - with all three the comparison is of the path LOGIC and its arithmetic alone. */
void oracle_path_replay(const oracle_scene *s, const float *pos /* n x 4 */, const float *dir /* n x 4 */, uint32_t n,
                        const float *rec, uint32_t rec_stride, const uint32_t *rec_count, float randBase, float envTheta,
                        uint32_t numBounces, const float *env_rec /* n x env_stride x 3, or NULL */, uint32_t env_stride,
                        const float *tex_rec /* n x tex_stride x 12, or NULL */, uint32_t tex_stride,
                        float *out_color /* n x 3 */, uint32_t *out_used, uint32_t *out_sig /* n x 2 */, uint32_t *out_env_used) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    rng_t g = {0.0f, rec + (size_t)i * rec_stride, 0, rec_count[i] < rec_stride ? rec_count[i] : rec_stride};
    if (g.limit == 0) g.limit = 0xFFFFFFFFu; /* (no values recorded: every draw is past the recording; 0 means unbounded) */
    replay_t rp = {{0u, 0u}, env_rec ? env_rec + (size_t)i * env_stride * 3 : NULL, 0u, env_stride,
                   tex_rec ? tex_rec + (size_t)i * tex_stride * 12 : NULL, tex_stride};
    vec3 color = trace_path(s, v3(pos[i * 4], pos[i * 4 + 1], pos[i * 4 + 2]), v3(dir[i * 4], dir[i * 4 + 1], dir[i * 4 + 2]),
                            randBase, envTheta, numBounces, &g, NULL, NULL, &rp);
    if (rec_count[i] == 0 && g.used) g.used = 0xFFFFFFFFu;
    out_color[i * 3] = om_clamp(color.x, 0.0f, 1024.0f);
    out_color[i * 3 + 1] = om_clamp(color.y, 0.0f, 1024.0f);
    out_color[i * 3 + 2] = om_clamp(color.z, 0.0f, 1024.0f);
    out_used[i] = g.used;
    out_sig[i * 2] = rp.sig[0]; out_sig[i * 2 + 1] = rp.sig[1];
    if (out_env_used) out_env_used[i] = rp.env_used;
  }
}

/* rnd() k times from a seed (tracer.fs:181): the sequence the probes above are replayed against. */
void oracle_rnd_sequence(const float *seeds, uint32_t n, uint32_t k, float *out) {
  for (uint32_t i = 0; i < n; ++i) {
    float sd = seeds[i];
    for (uint32_t j = 0; j < k; ++j) out[(size_t)i * k + j] = rnd_seed(&sd);
  }
}

/* ---- draw.fs (1-93): exposure -> ACES fit -> saturation -> gamma, optional 5x5 firefly filter ----
 * in: RGBA32F accumulator (W*H*4), out: RGBA8 (W*H*4) as the canvas would hold it
 * (byte = floor(clamp(c,0,1)*255 + 0.5)).  Texel fetches outside the buffer return 0
 * (robust-access texelFetch).  scale (main.js resScale) is 1. */
static inline float draw_luma(vec3 c) { return v_dot(c, v3(0.2126f, 0.7152f, 0.0722f)); }
static inline vec3 draw_fetch(const float *acc, int W, int H, int x, int y) {
  if (x < 0 || y < 0 || x >= W || y >= H) return v3(0.0f, 0.0f, 0.0f);
  const float *p = acc + ((size_t)y * W + x) * 4;
  return v3(p[0], p[1], p[2]);
}
static inline float rrt_odt(float v) { /* draw.fs:32-37 */
  float a = om_fma(v, v + 0.0245786f, -0.000090537f);
  float b = om_fma(v, om_fma(0.983729f, v, 0.4329510f), 0.238081f);
  return a / b;
}
void oracle_draw_scaled(const float *acc, uint32_t W, uint32_t H, float exposure, float saturation, int denoise,
                        float maxSigma, float scale, uint8_t *out);
void oracle_draw(const float *acc, uint32_t W, uint32_t H, float exposure, float saturation, int denoise,
                 float maxSigma, uint8_t *out) {
  oracle_draw_scaled(acc, W, H, exposure, saturation, denoise, maxSigma, 1.0f, out);
}

/* draw.fs with its `scale` uniform: every fetch is at ivec2(gl_FragCoord * scale) (+ the filter offset), draw.fs:59,87 */
void oracle_draw_scaled(const float *acc, uint32_t W, uint32_t H, float exposure, float saturation, int denoise,
                        float maxSigma, float scale, uint8_t *out) {
#pragma omp parallel for schedule(static)
  for (int64_t yy = 0; yy < (int64_t)H; ++yy)
    for (uint32_t xx = 0; xx < W; ++xx) {
      int x = (int)(((float)xx + 0.5f) * scale), y = (int)(((float)yy + 0.5f) * scale);
      vec3 c;
      if (denoise) { /* filterFireflies, draw.fs:52-80 */
        float sum = 0.0f, sq = 0.0f, middleLuma = 0.0f;
        vec3 middle = v3(0.0f, 0.0f, 0.0f);
        for (int i = 0; i < 5; ++i)
          for (int j = 0; j < 5; ++j) {
            int ox = i - 2, oy = j - 2;
            vec3 col = draw_fetch(acc, (int)W, (int)H, (int)x + ox, (int)y + oy);
            float l = draw_luma(col);
            if (ox == 0 && oy == 0) { middle = col; middleLuma = l; continue; }
            sum += l;
            sq = om_fma(l, l, sq);
          }
        float mean = sum / 24.0f;
        float variance = om_fma(-mean, mean, sq / 24.0f);
        float sigma = sqrtf(variance);
        if (om_abs(middleLuma - mean) > maxSigma * sigma) middle = v_scale(middle, mean / middleLuma);
        c = v_scale(middle, exposure);
      } else {
        c = v_scale(draw_fetch(acc, (int)W, (int)H, (int)x, (int)y), exposure);
      }
      /* ACESFitted (draw.fs:39-50): row vector times the column-major constant matrices */
      vec3 a = v3(v_dot(c, v3(0.59719f, 0.35458f, 0.04823f)), v_dot(c, v3(0.07600f, 0.90834f, 0.01566f)),
                  v_dot(c, v3(0.02840f, 0.13383f, 0.83777f)));
      a = v3(rrt_odt(a.x), rrt_odt(a.y), rrt_odt(a.z));
      vec3 m = v3(v_dot(a, v3(1.60475f, -0.53108f, -0.07367f)), v_dot(a, v3(-0.10208f, 1.10813f, -0.00605f)),
                  v_dot(a, v3(-0.00327f, -0.07276f, 1.07602f)));
      m = v3(om_clamp(m.x, 0.0f, 1.0f), om_clamp(m.y, 0.0f, 1.0f), om_clamp(m.z, 0.0f, 1.0f));
      float l = draw_luma(m);
      float os = 1.0f - saturation;
      m = v3(om_fma(m.x, saturation, l * os), om_fma(m.y, saturation, l * os), om_fma(m.z, saturation, l * os));
      float g[3] = {om_pow(m.x, 0.454545f), om_pow(m.y, 0.454545f), om_pow(m.z, 0.454545f)};
      uint8_t *o = out + ((size_t)yy * W + xx) * 4;
      for (int k = 0; k < 3; ++k) o[k] = (uint8_t)om_floor(om_fma(om_clamp(g[k], 0.0f, 1.0f), 255.0f, 0.5f));
      o[3] = 255;
    }
}
