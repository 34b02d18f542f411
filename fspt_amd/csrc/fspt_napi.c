/*
 * fspt_napi.c — thin N-API addon over the libfspt C ABI (include/fspt.h).
 *
 * The reference's host language is JavaScript; this is the binding a
 * maintainer loads from main.js in place of the WebGL2 calls.
 * Every export is a 1:1 wrapper: TypedArrays in, a caller-owned Float32Array
 * out, libfspt error codes become JS Errors carrying fspt_last_error().
 *
 *   sceneCreate({bvh,tri,mat,norm,uv: Float32Array, atlas: Uint8Array, atlasRes,
 *                atlasLayers, env: Uint8Array|null, envW, envH, bins: Uint32Array,
 *                leafSize}, device)                -> scene handle   (initBVH uploads, main.js:408-437)
 *   targetCreate(scene, W, H)                      -> target handle  (initBuffers, main.js:598-617)
 *   camera(target, P[3], I[3], fovScale, lens[2], randBase)          (drawCamera, main.js:741-756)
 *   trace(target, tick, randBase, envTheta, numBounces)              (drawTracer, main.js:758-807)
 *   traceTest(target, tick)                                          (drawTracer with bvh_test.fs, mode=test)
 *   render(target, {P,I,fovScale,lens,envTheta,numBounces}, firstTick, nTicks, seed)
 *   clear(target) / sync(target)                                     (clear, main.js:826-836)
 *   readRadiance(target, Float32Array(W*H*4))                        (what draw.fs:87 reads)
 *   draw(target, exposure, saturation, denoise, maxSigma, Uint8Array(W*H*4))   (drawQuad, main.js:809-824)
 *   setShard(target, shard, nShards, tile) / setPipeline(target, pipeline, batch)
 *   setMemoryLimit(target, bytes) / pathStateBytes(target) -> {bytes, batchTicks} / prepare(target) / setTail(target, round) / setDeferred(target, on)
 *   renderAsync(target, params, firstTick, nTicks, seed) -> Promise   (fspt_render + fspt_sync as napi_async_work)
 *   multiCreate(sceneDesc, [devices], W, H) -> multi handle; multiCamera / multiTrace / multiRender / multiRenderAsync /
 *   multiClear / multiSync / multiReadRadiance / multiDraw / multiTarget(multi, i) / multiDestroy   (fspt_multi_*: one frame
 *   over several GPUs from this one JS thread, tiles gathered onto the first device at read-out)
 *   multiSetExchange / multiGetExchange / multiLastStageMs(multi, nDevices) -> Float32Array(nDevices * 4)
 * Handles are napi externals with finalizers (a dropped tracer frees its device memory when it is collected), and while
 * a renderAsync job runs every other call on its target / multi throws Error('render in flight') - see "handles" below.
 *   enableCounters(target, on) / counters(target) -> object
 *   builderCreate / builderParseObj / builderCommit / builderNormalize / builderBuild / builderAutofocus /
 *   builderDestroy                                                   (native obj_loader.js + bvh.js, 1:1 fspt_builder_*)
 *   envBins(Uint8Array rgbe, w, h) -> Uint32Array                    (native env_sampler.js)
 *   sceneDestroy / targetDestroy / deviceCount / abiVersion / setTextureInterleaveBudget
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fspt.h"
#include "fspt_tuning.h"

#define NAPI_OK(call)                                                  \
  do {                                                                 \
    if ((call) != napi_ok) {                                           \
      napi_throw_error(env, NULL, "fspt_napi: N-API call failed: " #call); \
      return NULL;                                                     \
    }                                                                  \
  } while (0)

#define FSPT_OK_OR_THROW(call)                                         \
  do {                                                                 \
    int rc_ = (call);                                                  \
    if (rc_ != 0) {                                                    \
      char msg_[640];                                                  \
      snprintf(msg_, sizeof(msg_), "libfspt error %d: %s", rc_, fspt_last_error()); \
      napi_throw_error(env, NULL, msg_);                               \
      return NULL;                                                     \
    }                                                                  \
  } while (0)

static napi_value undefined(napi_env env) { napi_value u; napi_get_undefined(env, &u); return u; }

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
    napi_throw_type_error(env, NULL, "fspt_napi: wrong number of arguments");
    return -1;
  }
  return 0;
}

/* typed array -> pointer + element count; NULL/undefined allowed when optional */
static int typed(napi_env env, napi_value v, napi_typedarray_type want, int optional, void **data, size_t *len) {
  napi_valuetype vt;
  napi_typeof(env, v, &vt);
  if (optional && (vt == napi_undefined || vt == napi_null)) { *data = NULL; *len = 0; return 0; }
  bool is = false;
  napi_is_typedarray(env, v, &is);
  napi_typedarray_type t;
  if (!is || napi_get_typedarray_info(env, v, &t, len, data, NULL, NULL) != napi_ok || t != want) {
    napi_throw_type_error(env, NULL, "fspt_napi: expected a TypedArray of the documented element type");
    return -1;
  }
  return 0;
}

static int prop(napi_env env, napi_value obj, const char *name, napi_value *out) {
  if (napi_get_named_property(env, obj, name, out) != napi_ok) {
    napi_throw_type_error(env, NULL, "fspt_napi: missing property");
    return -1;
  }
  return 0;
}
static int prop_u32(napi_env env, napi_value obj, const char *name, uint32_t *out) {
  napi_value v;
  if (prop(env, obj, name, &v)) return -1;
  if (napi_get_value_uint32(env, v, out) != napi_ok) { napi_throw_type_error(env, NULL, name); return -1; }
  return 0;
}
static int get_f64(napi_env env, napi_value v, double *out) {
  if (napi_get_value_double(env, v, out) != napi_ok) { napi_throw_type_error(env, NULL, "fspt_napi: expected a number"); return -1; }
  return 0;
}
static int prop_f64(napi_env env, napi_value obj, const char *name, double dflt, double *out) {
  napi_value v; napi_valuetype vt;
  if (napi_get_named_property(env, obj, name, &v) != napi_ok) { *out = dflt; return 0; }
  napi_typeof(env, v, &vt);
  if (vt == napi_undefined || vt == napi_null) { *out = dflt; return 0; }
  return get_f64(env, v, out);
}
/* NUM_BOUNCES from JS: a finite integer in [0, FSPT_MAX_BOUNCES].  A plain uint32 conversion would turn -1 into
 * 4294967295 and 2.5 into 2 silently; anything outside the range is a caller bug, reported as a RangeError. */
static int get_bounces(napi_env env, napi_value v, uint32_t *out) {
  double d;
  if (get_f64(env, v, &d)) return -1;
  if (!(d >= 0.0 && d <= (double)FSPT_MAX_BOUNCES) || d != (double)(uint32_t)d) {
    napi_throw_range_error(env, NULL, "fspt_napi: numBounces must be an integer in [0, 64]");
    return -1;
  }
  *out = (uint32_t)d;
  return 0;
}
static int float_list(napi_env env, napi_value arr, float *out, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
    napi_value e; double d;
    if (napi_get_element(env, arr, i, &e) != napi_ok || napi_get_value_double(env, e, &d) != napi_ok) {
      napi_throw_type_error(env, NULL, "fspt_napi: expected an array of numbers");
      return -1;
    }
    out[i] = (float)d;
  }
  return 0;
}
/* ---------------------------------------------------------------- handles
 * A handle is an napi external around a small box, not the bare library pointer:
 *   - the box knows what it holds, so a scene passed where a target is expected is a TypeError, not a crash;
 *   - `busy` counts the renderAsync jobs in flight on it: while a libuv worker is inside fspt_render, every other addon
 *     call on that target / multi (readRadiance, tick, clear, destroy, ...) throws Error('render in flight') instead
 *     of racing the worker (the library's contract is one thread at a time per target, include/fspt.h; the reference's
 *     tick() is single-threaded, main.js:838-857).  Set and cleared on the JS thread only;
 *   - the external has a finalizer: a tracer that is dropped without close() gives its device memory back when the
 *     JS object is collected (fspt_target_destroy waits for the target's stream first);
 *   - a target keeps its scene alive (a strong reference to the scene's external), a multi's per-device target
 *     (multiTarget) keeps the multi alive and dies with it; sceneDestroy refuses while targets of the scene exist. */
enum { H_SCENE = 1, H_TARGET = 2, H_MULTI = 3, H_BUILDER = 4 };
static const char *const kind_name[] = {"?", "scene", "target", "multi", "builder"};
typedef struct fspt_handle {
  int kind;
  void *ptr;                   /* NULL once destroyed */
  int owned;                   /* 0: a multi's per-device target: owned by the multi */
  int busy;                    /* renderAsync jobs in flight on this handle (a target's job also counts on its scene) */
  int children;                /* scene: live targets made from it */
  struct fspt_handle *parent;  /* target -> scene box; multi's target -> multi box (valid while parent_ref is held) */
  napi_ref parent_ref;
} fspt_handle;

static void handle_release(napi_env env, fspt_handle *b) {
  if (b->ptr && b->owned) {
    switch (b->kind) {
      case H_SCENE: fspt_scene_destroy((fspt_scene *)b->ptr); break;
      case H_TARGET: fspt_target_destroy((fspt_target *)b->ptr); break; /* waits for its stream, drops recorded ticks */
      case H_MULTI: fspt_multi_destroy((fspt_multi *)b->ptr); break;
      case H_BUILDER: fspt_builder_destroy((fspt_builder *)b->ptr); break;
    }
  }
  b->ptr = NULL;
  if (b->parent) {
    if (b->kind == H_TARGET && b->owned && b->parent->children > 0) b->parent->children--;
    b->parent = NULL;
  }
  if (b->parent_ref) { napi_delete_reference(env, b->parent_ref); b->parent_ref = NULL; }
}
static void handle_finalize(napi_env env, void *data, void *hint) {
  (void)hint;
  fspt_handle *b = (fspt_handle *)data;
  handle_release(env, b); /* (never busy here: a job in flight holds a reference to the external) */
  free(b);
}
/* parent_val / parent: the external and box this handle keeps alive (NULL: none) */
static napi_value make_handle(napi_env env, int kind, void *ptr, int owned, napi_value parent_val, fspt_handle *parent) {
  fspt_handle *b = (fspt_handle *)calloc(1, sizeof(fspt_handle));
  napi_value ext;
  if (!b) { napi_throw_error(env, NULL, "fspt_napi: out of memory"); return NULL; }
  b->kind = kind; b->ptr = ptr; b->owned = owned; b->parent = parent;
  if (parent && napi_create_reference(env, parent_val, 1, &b->parent_ref) != napi_ok) b->parent = NULL;
  if (napi_create_external(env, b, handle_finalize, NULL, &ext) != napi_ok) {
    handle_release(env, b); free(b);
    napi_throw_error(env, NULL, "fspt_napi: napi_create_external failed");
    return NULL;
  }
  if (b->parent && kind == H_TARGET && owned) b->parent->children++;
  return ext;
}
static int handle_busy(const fspt_handle *b) { return b->busy || (!b->owned && b->parent && b->parent->busy); }
static int unwrap_box(napi_env env, napi_value v, int kind, fspt_handle **out) {
  void *d = NULL;
  napi_valuetype vt;
  if (napi_typeof(env, v, &vt) != napi_ok || vt != napi_external || napi_get_value_external(env, v, &d) != napi_ok || !d ||
      ((fspt_handle *)d)->kind != kind) {
    char msg[96];
    snprintf(msg, sizeof msg, "fspt_napi: expected a %s handle", kind_name[kind]);
    napi_throw_type_error(env, NULL, msg);
    return -1;
  }
  fspt_handle *b = (fspt_handle *)d;
  if (!b->ptr || (!b->owned && (!b->parent || !b->parent->ptr))) {
    char msg[96];
    snprintf(msg, sizeof msg, "fspt_napi: the %s handle was destroyed", kind_name[kind]);
    napi_throw_error(env, NULL, msg);
    return -1;
  }
  *out = b;
  return 0;
}
/* the library pointer behind a handle; throws (and returns non-zero) on a wrong / destroyed handle and while a
 * renderAsync job is in flight on it */
static int unwrap_k(napi_env env, napi_value v, int kind, void **out) {
  fspt_handle *b;
  if (unwrap_box(env, v, kind, &b)) return -1;
  if (handle_busy(b)) { napi_throw_error(env, NULL, "render in flight"); return -1; }
  *out = b->ptr;
  return 0;
}

/* ------------------------------------------------------------------ scene */
/* {bvh, tri, mat, norm, uv, atlas, atlasRes, atlasLayers, env, envW, envH, bins, leafSize} -> fspt_scene_desc viewing
 * the TypedArrays (valid while the JS object is alive, i.e. for the duration of the call) */
static int parse_scene_desc(napi_env env, napi_value obj, fspt_scene_desc *d) {
  memset(d, 0, sizeof(*d));
  napi_value v; void *p; size_t n;
  if (prop(env, obj, "bvh", &v) || typed(env, v, napi_float32_array, 0, &p, &n)) return -1;
  d->bvh = (const float *)p; d->n_nodes = (uint32_t)(n / 9);
  if (prop(env, obj, "tri", &v) || typed(env, v, napi_float32_array, 0, &p, &n)) return -1;
  d->tri = (const float *)p; d->n_tris = (uint32_t)(n / 9);
  size_t nm, nn, nu;
  if (prop(env, obj, "mat", &v) || typed(env, v, napi_float32_array, 0, &p, &nm)) return -1;
  d->mat = (const float *)p;
  if (prop(env, obj, "norm", &v) || typed(env, v, napi_float32_array, 0, &p, &nn)) return -1;
  d->norm = (const float *)p;
  if (prop(env, obj, "uv", &v) || typed(env, v, napi_float32_array, 0, &p, &nu)) return -1;
  d->uv = (const float *)p;
  if (nm != (size_t)d->n_tris * 12 || nn != (size_t)d->n_tris * 27 || nu != (size_t)d->n_tris * 6) {
    napi_throw_range_error(env, NULL, "fspt_napi: mat/norm/uv lengths do not match tri (12/27/6 floats per triangle)");
    return -1;
  }
  if (prop(env, obj, "atlas", &v) || typed(env, v, napi_uint8_array, 0, &p, &n)) return -1;
  d->atlas = (const uint8_t *)p;
  if (prop_u32(env, obj, "atlasRes", &d->atlas_res) || prop_u32(env, obj, "atlasLayers", &d->atlas_layers)) return -1;
  if (n != (size_t)d->atlas_res * d->atlas_res * d->atlas_layers * 4) {
    napi_throw_range_error(env, NULL, "fspt_napi: atlas length != atlasRes^2 * atlasLayers * 4");
    return -1;
  }
  if (prop(env, obj, "env", &v) || typed(env, v, napi_uint8_array, 1, &p, &n)) return -1;
  d->env = (const uint8_t *)p;
  if (d->env) {
    if (prop_u32(env, obj, "envW", &d->env_w) || prop_u32(env, obj, "envH", &d->env_h)) return -1;
    if (n != (size_t)d->env_w * d->env_h * 4) { napi_throw_range_error(env, NULL, "fspt_napi: env length != envW*envH*4"); return -1; }
  }
  if (prop(env, obj, "bins", &v) || typed(env, v, napi_uint32_array, 0, &p, &n)) return -1;
  d->bins = (const uint32_t *)p; d->n_bins = (uint32_t)(n / 4);
  if (prop_u32(env, obj, "leafSize", &d->leaf_size)) return -1;
  return 0;
}
static napi_value SceneCreate(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  fspt_scene_desc d;
  if (parse_scene_desc(env, a[0], &d)) return NULL;
  int32_t device = 0;
  napi_get_value_int32(env, a[1], &device);
  fspt_scene *s = NULL;
  FSPT_OK_OR_THROW(fspt_scene_create(&d, device, &s));
  napi_value ext = make_handle(env, H_SCENE, s, 1, NULL, NULL);
  if (!ext) fspt_scene_destroy(s);
  return ext;
}
static napi_value SceneDestroy(napi_env env, napi_callback_info info) {
  napi_value a[1]; void *h; fspt_handle *hb;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_SCENE, &h) || unwrap_box(env, a[0], H_SCENE, &hb)) return NULL;
  if (hb->children > 0) { napi_throw_error(env, NULL, "fspt_napi: the scene still has targets (destroy them first)"); return NULL; }
  handle_release(env, hb);
  return undefined(env);
}

/* ----------------------------------------------------------------- target */
static napi_value TargetCreate(napi_env env, napi_callback_info info) {
  napi_value a[3]; void *h; uint32_t W, H;
  if (get_args(env, info, 3, a) || unwrap_k(env, a[0], H_SCENE, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &W));
  NAPI_OK(napi_get_value_uint32(env, a[2], &H));
  fspt_target *t = NULL;
  fspt_handle *sb;
  if (unwrap_box(env, a[0], H_SCENE, &sb)) return NULL;
  FSPT_OK_OR_THROW(fspt_target_create((fspt_scene *)h, W, H, &t));
  napi_value ext = make_handle(env, H_TARGET, t, 1, a[0], sb);
  if (!ext) fspt_target_destroy(t);
  return ext;
}
static napi_value TargetDestroy(napi_env env, napi_callback_info info) {
  napi_value a[1]; void *h; fspt_handle *hb;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_TARGET, &h) || unwrap_box(env, a[0], H_TARGET, &hb)) return NULL;
  if (!hb->owned) { napi_throw_error(env, NULL, "fspt_napi: a multi's per-device target is destroyed with the multi"); return NULL; }
  handle_release(env, hb);
  return undefined(env);
}
static napi_value Camera(napi_env env, napi_callback_info info) {
  napi_value a[6]; void *h; float P[3], I[3], lens[2]; double fov, rb;
  if (get_args(env, info, 6, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  if (float_list(env, a[1], P, 3) || float_list(env, a[2], I, 3) || get_f64(env, a[3], &fov) || float_list(env, a[4], lens, 2) ||
      get_f64(env, a[5], &rb)) return NULL;
  FSPT_OK_OR_THROW(fspt_camera((fspt_target *)h, P, I, (float)fov, lens, (float)rb));
  return undefined(env);
}
static napi_value TraceTest(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; uint32_t tick;
  if (get_args(env, info, 2, a)) return NULL;
  if (unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &tick));
  FSPT_OK_OR_THROW(fspt_trace_test((fspt_target *)h, tick));
  return undefined(env);
}
static napi_value Trace(napi_env env, napi_callback_info info) {
  napi_value a[5]; void *h; uint32_t tick, nb; double rb, theta;
  if (get_args(env, info, 5, a)) return NULL;
  /* scalars first: a bad numBounces is reported as such even when the handle is bad too */
  if (get_f64(env, a[2], &rb) || get_f64(env, a[3], &theta) || get_bounces(env, a[4], &nb)) return NULL;
  if (unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &tick));
  FSPT_OK_OR_THROW(fspt_trace((fspt_target *)h, tick, (float)rb, (float)theta, nb));
  return undefined(env);
}
static int parse_camera_params(napi_env env, napi_value obj, fspt_camera_params *cp) {
  napi_value v; double d;
  memset(cp, 0, sizeof(*cp));
  if (prop(env, obj, "P", &v) || float_list(env, v, cp->P, 3)) return -1;
  if (prop(env, obj, "I", &v) || float_list(env, v, cp->I, 3)) return -1;
  if (prop(env, obj, "lens", &v) || float_list(env, v, cp->lens, 2)) return -1;
  if (prop_f64(env, obj, "fovScale", 0.5, &d)) return -1;
  cp->fov_scale = (float)d;
  if (prop_f64(env, obj, "envTheta", 0.0, &d)) return -1;
  cp->env_theta = (float)d;
  napi_value nbv; napi_valuetype nbt = napi_undefined;
  cp->num_bounces = 4; /* tracer.fs:9 */
  if (napi_get_named_property(env, obj, "numBounces", &nbv) == napi_ok) napi_typeof(env, nbv, &nbt);
  if (nbt != napi_undefined && nbt != napi_null && get_bounces(env, nbv, &cp->num_bounces)) return -1;
  return 0;
}
/* seed: BigInt (full 64-bit xorshift state) or a number < 2^53 */
static int parse_seed(napi_env env, napi_value v, uint64_t *seed64) {
  napi_valuetype st;
  napi_typeof(env, v, &st);
  if (st == napi_bigint) {
    bool lossless = true;
    if (napi_get_value_bigint_uint64(env, v, seed64, &lossless) != napi_ok) { napi_throw_type_error(env, NULL, "fspt_napi: bad seed"); return -1; }
    return 0;
  }
  double seed;
  if (get_f64(env, v, &seed)) return -1;
  *seed64 = (uint64_t)seed;
  return 0;
}
static napi_value Render(napi_env env, napi_callback_info info) {
  napi_value a[5]; void *h; uint32_t first, n; uint64_t seed64 = 0;
  if (get_args(env, info, 5, a)) return NULL;
  fspt_camera_params cp;
  if (parse_camera_params(env, a[1], &cp)) return NULL;
  if (unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[2], &first));
  NAPI_OK(napi_get_value_uint32(env, a[3], &n));
  if (parse_seed(env, a[4], &seed64)) return NULL;
  FSPT_OK_OR_THROW(fspt_render((fspt_target *)h, &cp, first, n, seed64));
  return undefined(env);
}

/* renderAsync(target, params, firstTick, nTicks, seed) -> Promise<undefined>
 * fspt_render + fspt_sync on a libuv worker thread (napi_async_work), so an interactive host keeps its event loop
 * while a long batch runs.  The caller must not touch the target until the promise settles: all calls for
 * one target still come from one thread AT A TIME (include/fspt.h). */
typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  fspt_handle *box;      /* the handle the job runs on: busy until render_complete */
  napi_ref box_ref;      /* keeps its external (and so the box and the target) alive while the worker runs */
  fspt_target *target;
  fspt_multi *multi;
  fspt_camera_params cp;
  uint32_t first, n;
  uint64_t seed;
  int rc;
  char err[600];
} render_job;
static void render_execute(napi_env env, void *data) {
  (void)env;
  render_job *j = (render_job *)data;
  if (j->multi) {
    j->rc = fspt_multi_render(j->multi, &j->cp, j->first, j->n, j->seed);
    if (j->rc == 0) j->rc = fspt_multi_sync(j->multi);
  } else {
    j->rc = fspt_render(j->target, &j->cp, j->first, j->n, j->seed);
    if (j->rc == 0) j->rc = fspt_sync(j->target);
  }
  /* fspt_last_error() is thread-local: fetch it on the thread that made the call */
  if (j->rc != 0) snprintf(j->err, sizeof(j->err), "libfspt error %d: %s", j->rc, fspt_last_error());
}
static void render_complete(napi_env env, napi_status status, void *data) {
  render_job *j = (render_job *)data;
  napi_value v;
  /* the target is the JS thread's again BEFORE the promise settles (its reactions may call readRadiance at once) */
  if (j->box) { j->box->busy--; if (j->box->kind == H_TARGET && j->box->owned && j->box->parent) j->box->parent->busy--; }
  if (status == napi_ok && j->rc == 0) {
    napi_get_undefined(env, &v);
    napi_resolve_deferred(env, j->deferred, v);
  } else {
    napi_value msg;
    napi_create_string_utf8(env, status == napi_ok ? j->err : "fspt_napi: async work cancelled", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &v);
    napi_reject_deferred(env, j->deferred, v);
  }
  if (j->box_ref) napi_delete_reference(env, j->box_ref);
  napi_delete_async_work(env, j->work);
  free(j);
}
static napi_value render_async(napi_env env, napi_callback_info info, int multi) {
  napi_value a[5], promise, name; fspt_handle *hb = NULL; uint64_t seed64 = 0;
  if (get_args(env, info, 5, a)) return NULL;
  render_job *j = (render_job *)calloc(1, sizeof(render_job));
  if (!j) { napi_throw_error(env, NULL, "fspt_napi: out of memory"); return NULL; }
  if (parse_camera_params(env, a[1], &j->cp) || unwrap_box(env, a[0], multi ? H_MULTI : H_TARGET, &hb) || parse_seed(env, a[4], &seed64) ||
      napi_get_value_uint32(env, a[2], &j->first) != napi_ok || napi_get_value_uint32(env, a[3], &j->n) != napi_ok) {
    bool pending = false;
    napi_is_exception_pending(env, &pending);
    if (!pending) napi_throw_type_error(env, NULL, "fspt_napi: renderAsync(target, params, firstTick, nTicks, seed)");
    free(j);
    return NULL;
  }
  if (handle_busy(hb)) { free(j); napi_throw_error(env, NULL, "render in flight"); return NULL; } /* one job at a time per target */
  if (multi) j->multi = (fspt_multi *)hb->ptr; else j->target = (fspt_target *)hb->ptr;
  j->seed = seed64;
  if (napi_create_reference(env, a[0], 1, &j->box_ref) != napi_ok) { free(j); napi_throw_error(env, NULL, "fspt_napi: napi_create_reference failed"); return NULL; }
  if (napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
      napi_create_string_utf8(env, "fspt_render", NAPI_AUTO_LENGTH, &name) != napi_ok ||
      napi_create_async_work(env, NULL, name, render_execute, render_complete, j, &j->work) != napi_ok ||
      napi_queue_async_work(env, j->work) != napi_ok) {
    if (j->work) napi_delete_async_work(env, j->work); /* created but not queued */
    napi_delete_reference(env, j->box_ref);
    free(j);
    napi_throw_error(env, NULL, "fspt_napi: could not queue the async work");
    return NULL;
  }
  /* from here until render_complete every other call on this handle throws 'render in flight' (unwrap_k) */
  j->box = hb;
  hb->busy++;
  if (hb->kind == H_TARGET && hb->owned && hb->parent) hb->parent->busy++;
  return promise;
}
static napi_value RenderAsync(napi_env env, napi_callback_info info) { return render_async(env, info, 0); }
static napi_value MultiRenderAsync(napi_env env, napi_callback_info info) { return render_async(env, info, 1); }

/* ------------------------------------------------------- several devices (fspt_multi_*) */
static napi_value MultiCreate(napi_env env, napi_callback_info info) {
  /* multiCreate(sceneDesc, [device, ...], W, H) */
  napi_value a[4]; uint32_t n = 0, W, H; int devices[64];
  if (get_args(env, info, 4, a)) return NULL;
  fspt_scene_desc d;
  if (parse_scene_desc(env, a[0], &d)) return NULL;
  if (napi_get_array_length(env, a[1], &n) != napi_ok || n == 0 || n > 64) { napi_throw_range_error(env, NULL, "fspt_napi: devices must be an array of 1..64 ordinals"); return NULL; }
  for (uint32_t i = 0; i < n; ++i) {
    napi_value e; int32_t v;
    if (napi_get_element(env, a[1], i, &e) != napi_ok || napi_get_value_int32(env, e, &v) != napi_ok) { napi_throw_type_error(env, NULL, "fspt_napi: device ordinals must be numbers"); return NULL; }
    devices[i] = v;
  }
  NAPI_OK(napi_get_value_uint32(env, a[2], &W));
  NAPI_OK(napi_get_value_uint32(env, a[3], &H));
  fspt_multi *m = NULL;
  FSPT_OK_OR_THROW(fspt_multi_create(&d, devices, n, W, H, &m));
  napi_value ext = make_handle(env, H_MULTI, m, 1, NULL, NULL);
  if (!ext) fspt_multi_destroy(m);
  return ext;
}
static napi_value MultiDestroy(napi_env env, napi_callback_info info) {
  napi_value a[1]; void *h; fspt_handle *hb;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_MULTI, &h) || unwrap_box(env, a[0], H_MULTI, &hb)) return NULL;
  handle_release(env, hb); /* its per-device target handles (multiTarget) are dead from here on */
  return undefined(env);
}
static napi_value MultiTarget(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; uint32_t i; fspt_target *t = NULL; fspt_handle *mb;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_MULTI, &h) || unwrap_box(env, a[0], H_MULTI, &mb)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &i));
  FSPT_OK_OR_THROW(fspt_multi_target((fspt_multi *)h, i, &t));
  return make_handle(env, H_TARGET, t, 0, a[0], mb); /* borrowed: lives and dies with the multi */
}
static napi_value MultiCamera(napi_env env, napi_callback_info info) {
  napi_value a[6]; void *h; float P[3], I[3], lens[2]; double fov, rb;
  if (get_args(env, info, 6, a) || unwrap_k(env, a[0], H_MULTI, &h)) return NULL;
  if (float_list(env, a[1], P, 3) || float_list(env, a[2], I, 3) || get_f64(env, a[3], &fov) || float_list(env, a[4], lens, 2) ||
      get_f64(env, a[5], &rb)) return NULL;
  FSPT_OK_OR_THROW(fspt_multi_camera((fspt_multi *)h, P, I, (float)fov, lens, (float)rb));
  return undefined(env);
}
static napi_value MultiTrace(napi_env env, napi_callback_info info) {
  napi_value a[5]; void *h; uint32_t tick, nb; double rb, theta;
  if (get_args(env, info, 5, a)) return NULL;
  if (get_f64(env, a[2], &rb) || get_f64(env, a[3], &theta) || get_bounces(env, a[4], &nb)) return NULL;
  if (unwrap_k(env, a[0], H_MULTI, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &tick));
  FSPT_OK_OR_THROW(fspt_multi_trace((fspt_multi *)h, tick, (float)rb, (float)theta, nb));
  return undefined(env);
}
static napi_value MultiRender(napi_env env, napi_callback_info info) {
  napi_value a[5]; void *h; uint32_t first, n; uint64_t seed64 = 0;
  if (get_args(env, info, 5, a)) return NULL;
  fspt_camera_params cp;
  if (parse_camera_params(env, a[1], &cp)) return NULL;
  if (unwrap_k(env, a[0], H_MULTI, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[2], &first));
  NAPI_OK(napi_get_value_uint32(env, a[3], &n));
  if (parse_seed(env, a[4], &seed64)) return NULL;
  FSPT_OK_OR_THROW(fspt_multi_render((fspt_multi *)h, &cp, first, n, seed64));
  return undefined(env);
}
static napi_value MultiClear(napi_env env, napi_callback_info info) {
  napi_value a[1]; void *h;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_MULTI, &h)) return NULL;
  FSPT_OK_OR_THROW(fspt_multi_clear((fspt_multi *)h));
  return undefined(env);
}
/* multiSetExchange(multi, mode): 0 peer copies, 1 RCCL send / recv of the packed tiles, 2 RCCL sum-reduce (fspt_multi.h) */
static napi_value MultiSetExchange(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; int32_t mode;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_MULTI, &h)) return NULL;
  NAPI_OK(napi_get_value_int32(env, a[1], &mode));
  FSPT_OK_OR_THROW(fspt_multi_set_exchange((fspt_multi *)h, mode));
  return undefined(env);
}
/* multiGetExchange(multi) -> {mode, rcclVersion} */
static napi_value MultiGetExchange(napi_env env, napi_callback_info info) {
  napi_value a[1], o, v; void *h; int mode = 0, ver = 0;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_MULTI, &h)) return NULL;
  FSPT_OK_OR_THROW(fspt_multi_get_exchange((fspt_multi *)h, &mode, &ver));
  NAPI_OK(napi_create_object(env, &o));
  NAPI_OK(napi_create_int32(env, mode, &v));
  NAPI_OK(napi_set_named_property(env, o, "mode", v));
  NAPI_OK(napi_create_int32(env, ver, &v));
  NAPI_OK(napi_set_named_property(env, o, "rcclVersion", v));
  return o;
}
/* multiLastStageMs(multi, nDevices) -> Float32Array(nDevices * 4): per device render / pack / transfer / scatter ms of the
 * most recent render + read-out (fspt_multi_last_stage_ms, include/fspt_multi.h; -1 = the stage did not run) */
static napi_value MultiLastStageMs(napi_env env, napi_callback_info info) {
  napi_value a[2], ab, ta; void *h, *data; uint32_t n;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_MULTI, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &n));
  if (n == 0 || n > 64) { napi_throw_range_error(env, NULL, "fspt_napi: nDevices must be 1..64"); return NULL; }
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 16, &data, &ab));
  NAPI_OK(napi_create_typedarray(env, napi_float32_array, (size_t)n * 4, ab, 0, &ta));
  FSPT_OK_OR_THROW(fspt_multi_last_stage_ms((fspt_multi *)h, (float *)data, n));
  return ta;
}
static napi_value MultiSync(napi_env env, napi_callback_info info) {
  napi_value a[1]; void *h;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_MULTI, &h)) return NULL;
  FSPT_OK_OR_THROW(fspt_multi_sync((fspt_multi *)h));
  return undefined(env);
}
/* The library writes W*H*4 elements into the caller's TypedArray: a shorter one is a heap overflow. */
static int check_frame_len(napi_env env, size_t n, uint32_t W, uint32_t H) {
  if (n == (size_t)W * H * 4u) return 0;
  char msg[160];
  snprintf(msg, sizeof msg, "fspt_napi: array of %zu elements, the %ux%u frame needs %zu", n, W, H, (size_t)W * H * 4u);
  napi_throw_range_error(env, NULL, msg);
  return 1;
}
static int check_target_len(napi_env env, void *h, size_t n) {
  uint32_t W = 0, H = 0;
  if (fspt_target_size((fspt_target *)h, &W, &H) != FSPT_OK) { napi_throw_error(env, NULL, fspt_last_error()); return 1; }
  return check_frame_len(env, n, W, H);
}
static int check_multi_len(napi_env env, void *h, size_t n) {
  uint32_t W = 0, H = 0;
  if (fspt_multi_size((fspt_multi *)h, &W, &H) != FSPT_OK) { napi_throw_error(env, NULL, fspt_last_error()); return 1; }
  return check_frame_len(env, n, W, H);
}
static napi_value MultiReadRadiance(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h, *p; size_t n;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_MULTI, &h) || typed(env, a[1], napi_float32_array, 0, &p, &n)) return NULL;
  if (check_multi_len(env, h, n)) return NULL;
  FSPT_OK_OR_THROW(fspt_multi_read_radiance((fspt_multi *)h, (float *)p));
  return a[1];
}
static napi_value MultiDraw(napi_env env, napi_callback_info info) {
  napi_value a[6]; void *h, *p; size_t n; double ex, sat, sig; bool den;
  if (get_args(env, info, 6, a) || unwrap_k(env, a[0], H_MULTI, &h)) return NULL;
  if (get_f64(env, a[1], &ex) || get_f64(env, a[2], &sat)) return NULL;
  NAPI_OK(napi_get_value_bool(env, a[3], &den));
  if (get_f64(env, a[4], &sig) || typed(env, a[5], napi_uint8_array, 0, &p, &n)) return NULL;
  if (check_multi_len(env, h, n)) return NULL;
  FSPT_OK_OR_THROW(fspt_multi_draw((fspt_multi *)h, (float)ex, (float)sat, den ? 1 : 0, (float)sig, (uint8_t *)p));
  return a[5];
}
static napi_value SetDeferred(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; bool on;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_bool(env, a[1], &on));
  FSPT_OK_OR_THROW(fspt_target_set_deferred((fspt_target *)h, on ? 1 : 0));
  return undefined(env);
}
static napi_value SetTail(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; int32_t r;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_int32(env, a[1], &r));
  FSPT_OK_OR_THROW(fspt_target_set_tail((fspt_target *)h, r));
  return undefined(env);
}
/* setStageTiming(target, on): the per-launch HIP event pairs behind fspt_last_stage_ms (fspt_tuning.h) */
static napi_value SetStageTiming(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; bool on;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_bool(env, a[1], &on));
  FSPT_OK_OR_THROW(fspt_target_set_stage_timing((fspt_target *)h, on ? 1 : 0));
  return undefined(env);
}
static napi_value Clear(napi_env env, napi_callback_info info) {
  napi_value a[1]; void *h;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  FSPT_OK_OR_THROW(fspt_clear((fspt_target *)h));
  return undefined(env);
}
static napi_value Sync(napi_env env, napi_callback_info info) {
  napi_value a[1]; void *h;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  FSPT_OK_OR_THROW(fspt_sync((fspt_target *)h));
  return undefined(env);
}
static napi_value ReadRadiance(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h, *p; size_t n;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_TARGET, &h) || typed(env, a[1], napi_float32_array, 0, &p, &n)) return NULL;
  if (check_target_len(env, h, n)) return NULL;
  FSPT_OK_OR_THROW(fspt_read_radiance((fspt_target *)h, (float *)p));
  return a[1];
}
static napi_value Draw(napi_env env, napi_callback_info info) {
  /* draw(target, exposure, saturation, denoise, maxSigma, Uint8Array(W*H*4)[, scale]) : drawQuad (main.js:809-824) */
  napi_value a[7]; void *h, *p; size_t n, argc = 7; double ex, sat, sig, scale = 1.0; bool den;
  if (napi_get_cb_info(env, info, &argc, a, NULL, NULL) != napi_ok || argc < 6) {
    napi_throw_type_error(env, NULL, "fspt_napi: wrong number of arguments");
    return NULL;
  }
  if (unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  if (get_f64(env, a[1], &ex) || get_f64(env, a[2], &sat)) return NULL;
  NAPI_OK(napi_get_value_bool(env, a[3], &den));
  if (get_f64(env, a[4], &sig) || typed(env, a[5], napi_uint8_array, 0, &p, &n)) return NULL;
  if (argc > 6) { napi_valuetype vt; napi_typeof(env, a[6], &vt); if (vt == napi_number && get_f64(env, a[6], &scale)) return NULL; }
  if (check_target_len(env, h, n)) return NULL;
  FSPT_OK_OR_THROW(fspt_draw_scaled((fspt_target *)h, (float)ex, (float)sat, den ? 1 : 0, (float)sig, (float)scale, (uint8_t *)p));
  return a[5];
}
static napi_value SetViewport(napi_env env, napi_callback_info info) {
  napi_value a[3]; void *h; uint32_t w, hh;
  if (get_args(env, info, 3, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &w));
  NAPI_OK(napi_get_value_uint32(env, a[2], &hh));
  FSPT_OK_OR_THROW(fspt_target_set_viewport((fspt_target *)h, w, hh));
  return undefined(env);
}
static napi_value SetShard(napi_env env, napi_callback_info info) {
  napi_value a[4]; void *h; uint32_t s, n, tile;
  if (get_args(env, info, 4, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &s));
  NAPI_OK(napi_get_value_uint32(env, a[2], &n));
  NAPI_OK(napi_get_value_uint32(env, a[3], &tile));
  FSPT_OK_OR_THROW(fspt_target_set_shard((fspt_target *)h, s, n, tile));
  return undefined(env);
}
static napi_value SetPipeline(napi_env env, napi_callback_info info) {
  napi_value a[3]; void *h; int32_t p; uint32_t b;
  if (get_args(env, info, 3, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_int32(env, a[1], &p));
  NAPI_OK(napi_get_value_uint32(env, a[2], &b));
  FSPT_OK_OR_THROW(fspt_target_set_pipeline((fspt_target *)h, p, b));
  return undefined(env);
}
static napi_value SetPool(napi_env env, napi_callback_info info) {
  /* setPool(target, paths, drainIterations, maxIterations, overlap): fspt_target_set_pool (stream scheduler) */
  napi_value a[5]; void *h; uint32_t paths, cap; int32_t drain, overlap;
  if (get_args(env, info, 5, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &paths));
  NAPI_OK(napi_get_value_int32(env, a[2], &drain));
  NAPI_OK(napi_get_value_uint32(env, a[3], &cap));
  NAPI_OK(napi_get_value_int32(env, a[4], &overlap));
  FSPT_OK_OR_THROW(fspt_target_set_pool((fspt_target *)h, paths, drain, cap, overlap));
  return undefined(env);
}
static napi_value SetTraceBudget(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; uint32_t steps;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &steps));
  FSPT_OK_OR_THROW(fspt_target_set_trace_budget((fspt_target *)h, steps));
  return undefined(env);
}
static napi_value SetMemoryLimit(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; double bytes;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_TARGET, &h) || get_f64(env, a[1], &bytes)) return NULL;
  if (!(bytes >= 0.0 && bytes < 1.8e19)) { napi_throw_range_error(env, NULL, "fspt_napi: memory limit must be >= 0 bytes"); return NULL; }
  FSPT_OK_OR_THROW(fspt_target_set_memory_limit((fspt_target *)h, (uint64_t)bytes));
  return undefined(env);
}
static napi_value SetTextureInterleaveBudget(napi_env env, napi_callback_info info) {
  napi_value a[1]; double bytes;
  if (get_args(env, info, 1, a) || get_f64(env, a[0], &bytes)) return NULL;
  if (!(bytes >= 0.0 && bytes < 1.8e19)) { napi_throw_range_error(env, NULL, "fspt_napi: budget must be >= 0 bytes"); return NULL; }
  FSPT_OK_OR_THROW(fspt_set_texture_interleave_budget((uint64_t)bytes));
  return undefined(env);
}
static napi_value PathStateBytes(napi_env env, napi_callback_info info) {
  napi_value a[1], o, v; void *h; uint64_t bytes = 0; uint32_t batch = 0;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  FSPT_OK_OR_THROW(fspt_target_path_state_bytes((fspt_target *)h, &bytes, &batch));
  NAPI_OK(napi_create_object(env, &o));
  NAPI_OK(napi_create_double(env, (double)bytes, &v));
  NAPI_OK(napi_set_named_property(env, o, "bytes", v));
  NAPI_OK(napi_create_uint32(env, batch, &v));
  NAPI_OK(napi_set_named_property(env, o, "batchTicks", v));
  return o;
}
static napi_value Prepare(napi_env env, napi_callback_info info) {
  napi_value a[1]; void *h;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  FSPT_OK_OR_THROW(fspt_target_prepare((fspt_target *)h));
  return undefined(env);
}
static napi_value EnableCounters(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; bool on;
  if (get_args(env, info, 2, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  NAPI_OK(napi_get_value_bool(env, a[1], &on));
  FSPT_OK_OR_THROW(fspt_enable_counters((fspt_target *)h, on ? 1 : 0));
  return undefined(env);
}
static napi_value GetCounters(napi_env env, napi_callback_info info) {
  napi_value a[1]; void *h;
  if (get_args(env, info, 1, a) || unwrap_k(env, a[0], H_TARGET, &h)) return NULL;
  fspt_counters c;
  FSPT_OK_OR_THROW(fspt_get_counters((fspt_target *)h, &c));
  napi_value o, v;
  NAPI_OK(napi_create_object(env, &o));
  const char *names[6] = {"samples", "rays", "steps", "leaves", "shades", "envLookups"};
  uint64_t vals[6] = {c.samples, c.rays, c.steps, c.leaves, c.shades, c.env_lookups};
  for (int i = 0; i < 6; ++i) {
    NAPI_OK(napi_create_double(env, (double)vals[i], &v));
    NAPI_OK(napi_set_named_property(env, o, names[i], v));
  }
  return o;
}

/* --------------------------------------------------------- scene pipeline */
static napi_value f32_out(napi_env env, size_t n, float **data) {
  napi_value ab, ta;
  if (napi_create_arraybuffer(env, n * 4, (void **)data, &ab) != napi_ok) return NULL;
  if (napi_create_typedarray(env, napi_float32_array, n, ab, 0, &ta) != napi_ok) return NULL;
  return ta;
}
/* --- native scene builder, one call per fspt_builder_* entry point (orchestrated by fspt.js buildScene) --- */
static int read_rotations(napi_env env, napi_value obj, double *rot, uint32_t cap, uint32_t *n, bool *present) {
  napi_value v;
  *n = 0; *present = false;
  if (napi_get_named_property(env, obj, "rotate", &v) != napi_ok) return 0;
  bool isarr = false;
  napi_is_array(env, v, &isarr);
  if (!isarr) return 0;
  *present = true;
  napi_get_array_length(env, v, n);
  if (*n > cap) { napi_throw_range_error(env, NULL, "fspt_napi: too many rotations (max 16)"); return -1; }
  for (uint32_t r = 0; r < *n; ++r) {
    napi_value re, ax, e;
    napi_get_element(env, v, r, &re);
    if (prop(env, re, "axis", &ax)) return -1;
    for (uint32_t k = 0; k < 3; ++k) { napi_get_element(env, ax, k, &e); if (get_f64(env, e, &rot[4 * r + k])) return -1; }
    if (prop_f64(env, re, "angle", 0.0, &rot[4 * r + 3])) return -1;
  }
  return 0;
}
static int read_vec3(napi_env env, napi_value obj, const char *name, double *out, bool *present) {
  napi_value v;
  *present = false;
  if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
  bool isarr = false;
  napi_is_array(env, v, &isarr);
  if (!isarr) return 0;
  *present = true;
  for (uint32_t k = 0; k < 3; ++k) { napi_value e; napi_get_element(env, v, k, &e); if (get_f64(env, e, &out[k])) return -1; }
  return 0;
}
static char *dup_string(napi_env env, napi_value v, size_t *len) {
  if (napi_get_value_string_utf8(env, v, NULL, 0, len) != napi_ok) { napi_throw_type_error(env, NULL, "fspt_napi: expected a string"); return NULL; }
  char *text = (char *)malloc(*len + 1);
  napi_get_value_string_utf8(env, v, text, *len + 1, len);
  return text;
}
static napi_value BuilderCreate(napi_env env, napi_callback_info info) {
  (void)info;
  fspt_builder *b = NULL;
  FSPT_OK_OR_THROW(fspt_builder_create(&b));
  napi_value h = make_handle(env, H_BUILDER, b, 1, NULL, NULL);
  if (!h) fspt_builder_destroy(b);
  return h;
}
static napi_value BuilderDestroy(napi_env env, napi_callback_info info) {
  napi_value a[1]; void *h;
  if (get_args(env, info, 1, a)) return NULL;
  fspt_handle *hb;
  if (unwrap_k(env, a[0], H_BUILDER, &h) || unwrap_box(env, a[0], H_BUILDER, &hb)) return NULL;
  handle_release(env, hb);
  return undefined(env);
}
/* builderParseObj(b, objText, {rotate, scale, translate, normals}, worldTransforms[] | null, skips[] | null)
 *   -> [{name, nTris, mtllib: string | null}]   groups in the reference's iteration order */
#define MAX_WORLD 16
static napi_value BuilderParseObj(napi_env env, napi_callback_info info) {
  napi_value a[5]; void *h;
  if (get_args(env, info, 5, a)) return NULL;
  if (unwrap_k(env, a[0], H_BUILDER, &h)) return NULL;
  fspt_builder *b = (fspt_builder *)h;
  fspt_prop_desc pd;
  memset(&pd, 0, sizeof(pd));
  double rot[4 * 16];
  uint32_t nrot = 0; bool present;
  if (read_rotations(env, a[2], rot, 16, &nrot, &present)) return NULL;
  pd.rotate = rot; pd.n_rotate = nrot;
  if (prop_f64(env, a[2], "scale", 1.0, &pd.scale)) return NULL;
  if (read_vec3(env, a[2], "translate", pd.translate, &present)) return NULL;
  napi_value v;
  if (napi_get_named_property(env, a[2], "normals", &v) == napi_ok) {
    char buf[16]; size_t len = 0;
    if (napi_get_value_string_utf8(env, v, buf, sizeof(buf), &len) == napi_ok) {
      if (!strcmp(buf, "smooth")) pd.normals_mode = 1; else if (!strcmp(buf, "mesh")) pd.normals_mode = 2;
    }
  }
  /* scene.worldTransforms */
  fspt_world_transform wt[MAX_WORLD];
  static double wrot[MAX_WORLD][4 * 16];
  uint32_t nworld = 0;
  bool isarr = false;
  memset(wt, 0, sizeof(wt));
  napi_is_array(env, a[3], &isarr);
  if (isarr) {
    napi_get_array_length(env, a[3], &nworld);
    if (nworld > MAX_WORLD) { napi_throw_range_error(env, NULL, "fspt_napi: too many worldTransforms (max 16)"); return NULL; }
    for (uint32_t i = 0; i < nworld; ++i) {
      napi_value e; bool has;
      napi_get_element(env, a[3], i, &e);
      if (read_rotations(env, e, wrot[i], 16, &wt[i].n_rotate, &has)) return NULL;
      wt[i].rotate = wrot[i]; wt[i].has_rotate = has ? 1u : 0u;
      if (!has) { if (read_vec3(env, e, "translate", wt[i].translate, &has)) return NULL; wt[i].has_translate = has ? 1u : 0u; }
    }
  }
  /* prop.skips */
  char *skips[64]; uint32_t nskips = 0;
  napi_is_array(env, a[4], &isarr);
  if (isarr) {
    uint32_t n = 0;
    napi_get_array_length(env, a[4], &n);
    if (n > 64) n = 64;
    for (uint32_t i = 0; i < n; ++i) {
      napi_value e; size_t len;
      napi_get_element(env, a[4], i, &e);
      napi_value str;
      if (napi_coerce_to_string(env, e, &str) != napi_ok) continue;
      skips[nskips] = dup_string(env, str, &len);
      if (skips[nskips]) nskips++;
    }
  }
  size_t len = 0;
  char *text = dup_string(env, a[1], &len);
  napi_value result = NULL;
  if (text) {
    uint32_t ng = 0;
    int rc = fspt_builder_parse_obj(b, text, len, &pd, wt, nworld, (const char *const *)skips, nskips, &ng);
    free(text);
    if (rc) {
      char msg[640]; snprintf(msg, sizeof(msg), "libfspt error %d: %s", rc, fspt_last_error()); napi_throw_error(env, NULL, msg);
    } else {
      napi_create_array_with_length(env, ng, &result);
      for (uint32_t g = 0; g < ng; ++g) {
        const char *name; uint32_t nt; int32_t mi;
        fspt_builder_group_info(b, g, &name, &nt, &mi);
        napi_value o, vv;
        napi_create_object(env, &o);
        napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &vv); napi_set_named_property(env, o, "name", vv);
        napi_create_uint32(env, nt, &vv); napi_set_named_property(env, o, "nTris", vv);
        if (mi >= 0) { const char *ln; fspt_builder_mtllib_name(b, (uint32_t)mi, &ln); napi_create_string_utf8(env, ln, NAPI_AUTO_LENGTH, &vv); }
        else napi_get_null(env, &vv);
        napi_set_named_property(env, o, "mtllib", vv);
        napi_set_element(env, result, g, o);
      }
    }
  }
  for (uint32_t i = 0; i < nskips; ++i) free(skips[i]);
  return result;
}
/* builderCommit(b, [{diffuseIndex, specularIndex, normalIndex, roughnessIndex, emittance[3], ior, dielectric}]) */
static napi_value BuilderCommit(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h;
  if (get_args(env, info, 2, a)) return NULL;
  if (unwrap_k(env, a[0], H_BUILDER, &h)) return NULL;
  uint32_t n = 0;
  NAPI_OK(napi_get_array_length(env, a[1], &n));
  fspt_group_material *gm = (fspt_group_material *)calloc(n ? n : 1, sizeof(*gm));
  napi_value result = NULL;
  for (uint32_t i = 0; i < n; ++i) {
    napi_value m; bool present;
    napi_get_element(env, a[1], i, &m);
    if (prop_f64(env, m, "diffuseIndex", 0, &gm[i].diffuse_layer) || prop_f64(env, m, "specularIndex", 0, &gm[i].emissive_layer) ||
        prop_f64(env, m, "normalIndex", 0, &gm[i].normal_layer) || prop_f64(env, m, "roughnessIndex", 0, &gm[i].mr_layer) ||
        prop_f64(env, m, "ior", 1.4, &gm[i].ior) || prop_f64(env, m, "dielectric", -1, &gm[i].dielectric) ||
        read_vec3(env, m, "emittance", gm[i].emittance, &present)) goto done;
  }
  {
    int rc = fspt_builder_commit_obj((fspt_builder *)h, gm, n);
    if (rc) { char msg[640]; snprintf(msg, sizeof(msg), "libfspt error %d: %s", rc, fspt_last_error()); napi_throw_error(env, NULL, msg); goto done; }
    result = undefined(env);
  }
done:
  free(gm);
  return result;
}
static napi_value BuilderNormalize(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; double size;
  if (get_args(env, info, 2, a)) return NULL;
  if (unwrap_k(env, a[0], H_BUILDER, &h)) return NULL;
  if (get_f64(env, a[1], &size)) return NULL;
  FSPT_OK_OR_THROW(fspt_builder_normalize((fspt_builder *)h, size));
  return undefined(env);
}
/* builderBuild(b, leafSize) -> {bvh,tri,mat,norm,uv: Float32Array, depth} */
static napi_value BuilderBuild(napi_env env, napi_callback_info info) {
  napi_value a[2]; void *h; uint32_t leaf = 4;
  if (get_args(env, info, 2, a)) return NULL;
  if (unwrap_k(env, a[0], H_BUILDER, &h)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &leaf));
  fspt_builder *b = (fspt_builder *)h;
  FSPT_OK_OR_THROW(fspt_builder_build(b, leaf));
  uint32_t nn, nt, depth;
  fspt_builder_counts(b, &nn, &nt, &depth);
  float *bvh, *tri, *mat, *norm, *uv;
  napi_value o, v;
  napi_create_object(env, &o);
  napi_value tb = f32_out(env, (size_t)nn * 9, &bvh), tt = f32_out(env, (size_t)nt * 9, &tri), tm = f32_out(env, (size_t)nt * 12, &mat),
             tn = f32_out(env, (size_t)nt * 27, &norm), tu = f32_out(env, (size_t)nt * 6, &uv);
  if (!tb || !tt || !tm || !tn || !tu) { napi_throw_error(env, NULL, "fspt_napi: allocation failed"); return NULL; }
  fspt_builder_get(b, bvh, tri, mat, norm, uv);
  napi_set_named_property(env, o, "bvh", tb); napi_set_named_property(env, o, "tri", tt); napi_set_named_property(env, o, "mat", tm);
  napi_set_named_property(env, o, "norm", tn); napi_set_named_property(env, o, "uv", tu);
  napi_create_uint32(env, depth, &v); napi_set_named_property(env, o, "depth", v);
  return o;
}
/* builderAutofocus(b, eye[3], dir[3]) -> distance (shootAutoFocusRay, main.js:447-546) */
static napi_value BuilderAutofocus(napi_env env, napi_callback_info info) {
  napi_value a[3]; void *h; double eye[3], dir[3], dist = 0;
  if (get_args(env, info, 3, a)) return NULL;
  if (unwrap_k(env, a[0], H_BUILDER, &h)) return NULL;
  for (uint32_t k = 0; k < 3; ++k) {
    napi_value e;
    napi_get_element(env, a[1], k, &e); if (get_f64(env, e, &eye[k])) return NULL;
    napi_get_element(env, a[2], k, &e); if (get_f64(env, e, &dir[k])) return NULL;
  }
  FSPT_OK_OR_THROW(fspt_builder_autofocus((fspt_builder *)h, eye, dir, &dist));
  napi_value v;
  NAPI_OK(napi_create_double(env, dist, &v));
  return v;
}
static napi_value EnvBins(napi_env env, napi_callback_info info) {
  napi_value a[3]; void *p; size_t n; uint32_t w, h;
  if (get_args(env, info, 3, a) || typed(env, a[0], napi_uint8_array, 0, &p, &n)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[1], &w));
  NAPI_OK(napi_get_value_uint32(env, a[2], &h));
  if (n != (size_t)w * h * 4) { napi_throw_range_error(env, NULL, "fspt_napi: rgbe length != w*h*4"); return NULL; }
  uint32_t nb = 0;
  FSPT_OK_OR_THROW(fspt_env_bins((const uint8_t *)p, w, h, NULL, 0, &nb));
  napi_value ab, ta; void *data;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nb * 16, &data, &ab));
  NAPI_OK(napi_create_typedarray(env, napi_uint32_array, (size_t)nb * 4, ab, 0, &ta));
  FSPT_OK_OR_THROW(fspt_env_bins((const uint8_t *)p, w, h, (uint32_t *)data, nb, &nb));
  return ta;
}
static napi_value RandBaseNext(napi_env env, napi_callback_info info) {
  /* state is a BigUint64Array(1) so the 64-bit xorshift state survives the JS boundary */
  napi_value a[1]; void *p; size_t n;
  if (get_args(env, info, 1, a) || typed(env, a[0], napi_biguint64_array, 0, &p, &n) || n < 1) return NULL;
  napi_value v;
  NAPI_OK(napi_create_double(env, (double)fspt_rand_base_next((uint64_t *)p), &v));
  return v;
}
static napi_value DeviceCount(napi_env env, napi_callback_info info) {
  napi_value v; napi_create_int32(env, fspt_device_count(), &v); return v;
}
/* deviceMemory(device) -> {free, total} in bytes (fspt_device_memory) */
static napi_value DeviceMemory(napi_env env, napi_callback_info info) {
  napi_value a[1], o, v; int32_t dev; uint64_t f = 0, t = 0;
  if (get_args(env, info, 1, a)) return NULL;
  NAPI_OK(napi_get_value_int32(env, a[0], &dev));
  FSPT_OK_OR_THROW(fspt_device_memory(dev, &f, &t));
  NAPI_OK(napi_create_object(env, &o));
  NAPI_OK(napi_create_double(env, (double)f, &v));
  NAPI_OK(napi_set_named_property(env, o, "free", v));
  NAPI_OK(napi_create_double(env, (double)t, &v));
  NAPI_OK(napi_set_named_property(env, o, "total", v));
  return o;
}
static napi_value AbiVersion(napi_env env, napi_callback_info info) {
  napi_value v; napi_create_int32(env, fspt_abi_version(), &v); return v;
}

static napi_value Init(napi_env env, napi_value exports) {
  struct { const char *name; napi_callback fn; } fns[] = {
      {"sceneCreate", SceneCreate}, {"sceneDestroy", SceneDestroy}, {"targetCreate", TargetCreate},
      {"targetDestroy", TargetDestroy}, {"camera", Camera}, {"trace", Trace}, {"traceTest", TraceTest}, {"render", Render}, {"clear", Clear},
      {"sync", Sync}, {"readRadiance", ReadRadiance}, {"draw", Draw}, {"setShard", SetShard}, {"setViewport", SetViewport}, {"setPipeline", SetPipeline}, {"setPool", SetPool}, {"setTraceBudget", SetTraceBudget},
      {"setMemoryLimit", SetMemoryLimit}, {"setTextureInterleaveBudget", SetTextureInterleaveBudget}, {"pathStateBytes", PathStateBytes}, {"prepare", Prepare}, {"setTail", SetTail}, {"setDeferred", SetDeferred}, {"setStageTiming", SetStageTiming},
      {"renderAsync", RenderAsync}, {"multiCreate", MultiCreate}, {"multiDestroy", MultiDestroy}, {"multiTarget", MultiTarget},
      {"multiCamera", MultiCamera}, {"multiTrace", MultiTrace}, {"multiRender", MultiRender}, {"multiRenderAsync", MultiRenderAsync},
      {"multiClear", MultiClear}, {"multiSync", MultiSync}, {"multiSetExchange", MultiSetExchange}, {"multiGetExchange", MultiGetExchange}, {"multiLastStageMs", MultiLastStageMs}, {"multiReadRadiance", MultiReadRadiance}, {"multiDraw", MultiDraw},
      {"enableCounters", EnableCounters}, {"counters", GetCounters}, {"builderCreate", BuilderCreate}, {"builderDestroy", BuilderDestroy},
      {"builderParseObj", BuilderParseObj}, {"builderCommit", BuilderCommit}, {"builderNormalize", BuilderNormalize},
      {"builderBuild", BuilderBuild}, {"builderAutofocus", BuilderAutofocus}, {"envBins", EnvBins},
      {"randBaseNext", RandBaseNext}, {"deviceCount", DeviceCount}, {"deviceMemory", DeviceMemory}, {"abiVersion", AbiVersion}};
  for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); ++i) {
    napi_value f;
    if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
    if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
  }
  return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
