"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (fspt_amd) never imports this.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FSPT_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # (FSPT_ORACLE_LIB: a sanitizer build, tools/sanitize_cpu.sh)
_F = C.POINTER(C.c_float)


class OScene(C.Structure):
    _fields_ = [
        ("bvh", _F), ("n_nodes", C.c_uint32),
        ("tri", _F), ("n_tris", C.c_uint32),
        ("mat", _F), ("norm", _F), ("uv", _F),
        ("atlas", C.POINTER(C.c_uint8)), ("atlas_res", C.c_uint32), ("atlas_layers", C.c_uint32),
        ("env", C.POINTER(C.c_uint8)), ("env_w", C.c_uint32), ("env_h", C.c_uint32),
        ("bins", C.POINTER(C.c_uint32)), ("n_bins", C.c_uint32),
        ("leaf_size", C.c_uint32),
    ]


class OCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "steps", "leaves", "shades", "env_lookups")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class OFirstHit(C.Structure):
    _fields_ = [("t", C.c_float), ("index", C.c_int32), ("origin", C.c_float * 3), ("bary", C.c_float * 3),
                ("uv", C.c_float * 2), ("diffuse", C.c_float * 3), ("emissive", C.c_float * 3),
                ("mr", C.c_float * 2), ("tex_normal", C.c_float * 3), ("macro_normal", C.c_float * 3),
                ("bary_normal", C.c_float * 3)]


FIRST_HIT_DTYPE = np.dtype([("t", "f4"), ("index", "i4"), ("origin", "f4", 3), ("bary", "f4", 3), ("uv", "f4", 2),
                            ("diffuse", "f4", 3), ("emissive", "f4", 3), ("mr", "f4", 2), ("tex_normal", "f4", 3),
                            ("macro_normal", "f4", 3), ("bary_normal", "f4", 3)])
assert FIRST_HIT_DTYPE.itemsize == C.sizeof(OFirstHit)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `make -C oracle`")
        l = C.CDLL(LIB_PATH)
        l.oracle_rand_base_next.restype = C.c_float
        l.oracle_rand_base_next.argtypes = [C.POINTER(C.c_uint64)]
        l.oracle_has_fma.restype = C.c_int
        if not l.oracle_has_fma():
            raise RuntimeError("oracle needs a CPU with FMA (built with -mfma)")
        _lib = l
    return _lib


def _fp(a):
    return a.ctypes.data_as(_F)


def oscene(arrays):
    """OScene viewing a fspt_amd.scene.SceneArrays-like object (keep it alive)."""
    s = OScene()
    s.bvh = _fp(arrays.bvh); s.n_nodes = arrays.bvh.size // 9
    s.tri = _fp(arrays.tri); s.n_tris = arrays.tri.size // 9
    s.mat = _fp(arrays.mat); s.norm = _fp(arrays.norm); s.uv = _fp(arrays.uv)
    s.atlas = arrays.atlas.ctypes.data_as(C.POINTER(C.c_uint8))
    s.atlas_res = arrays.atlas_res; s.atlas_layers = arrays.atlas_layers
    if arrays.env is not None:
        s.env = arrays.env.ctypes.data_as(C.POINTER(C.c_uint8)); s.env_w = arrays.env_w; s.env_h = arrays.env_h
    else:
        s.env = None; s.env_w = 0; s.env_h = 0
    s.bins = arrays.bins.ctypes.data_as(C.POINTER(C.c_uint32)); s.n_bins = arrays.bins.size // 4
    s.leaf_size = arrays.leaf_size
    return s


def camera(W, H, P, I, fov_scale, lens, rand_base):
    pos = np.zeros((H, W, 4), np.float32); d = np.zeros((H, W, 4), np.float32)
    lib().oracle_camera(C.c_uint32(W), C.c_uint32(H), (C.c_float * 3)(*P), (C.c_float * 3)(*I), C.c_float(fov_scale),
                        (C.c_float * 2)(*lens), C.c_float(rand_base), _fp(pos), _fp(d))
    return pos, d


def camera_probe(W, H, P, I, fov_scale, lens, rec):
    """camera.fs main with the GLSL's own rnd() values (rec: [H, W, 4], call order) replayed."""
    rec = np.ascontiguousarray(rec, np.float32)
    assert rec.shape == (H, W, 4)
    pos = np.zeros((H, W, 4), np.float32); d = np.zeros((H, W, 4), np.float32)
    lib().oracle_camera_probe(C.c_uint32(W), C.c_uint32(H), (C.c_float * 3)(*P), (C.c_float * 3)(*I), C.c_float(fov_scale),
                              (C.c_float * 2)(*lens), _fp(rec), _fp(pos), _fp(d))
    return pos, d


def trace(arrays, W, H, pos, d, tick, rand_base, env_theta, num_bounces, accum, counters=None, first_hits=False,
          shard=0, n_shards=1, tile=32):
    s = oscene(arrays)
    pos = np.ascontiguousarray(pos, np.float32); d = np.ascontiguousarray(d, np.float32)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous
    fh = np.zeros(W * H, FIRST_HIT_DTYPE) if first_hits else None
    lib().oracle_trace(C.byref(s), C.c_uint32(W), C.c_uint32(H), _fp(pos), _fp(d), C.c_uint32(tick),
                       C.c_float(rand_base), C.c_float(env_theta), C.c_uint32(num_bounces), _fp(accum),
                       C.byref(counters) if counters is not None else None,
                       fh.ctypes.data_as(C.c_void_p) if fh is not None else None,
                       C.c_uint32(shard), C.c_uint32(n_shards), C.c_uint32(tile))
    return fh


def trace_test(arrays, W, H, pos, d, tick, accum, shard=0, n_shards=1, tile=32):
    """bvh_test.fs main (the reference's mode=test draw)."""
    s = oscene(arrays)
    pos = np.ascontiguousarray(pos, np.float32); d = np.ascontiguousarray(d, np.float32)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous
    lib().oracle_trace_test(C.byref(s), C.c_uint32(W), C.c_uint32(H), _fp(pos), _fp(d), C.c_uint32(tick), _fp(accum),
                            C.c_uint32(shard), C.c_uint32(n_shards), C.c_uint32(tile))


def render(arrays, W, H, P, I, fov_scale, lens, env_theta, num_bounces, first_tick, n_ticks, seed, accum,
           counters=None, shard=0, n_shards=1, tile=32):
    s = oscene(arrays)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous
    lib().oracle_render(C.byref(s), C.c_uint32(W), C.c_uint32(H), (C.c_float * 3)(*P), (C.c_float * 3)(*I),
                        C.c_float(fov_scale), (C.c_float * 2)(*lens), C.c_float(env_theta), C.c_uint32(num_bounces),
                        C.c_uint32(first_tick), C.c_uint32(n_ticks), C.c_uint64(seed), _fp(accum),
                        C.byref(counters) if counters is not None else None,
                        C.c_uint32(shard), C.c_uint32(n_shards), C.c_uint32(tile))


def intersect(arrays, rays):
    s = oscene(arrays)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = rays.shape[0]
    t = np.zeros(n, np.float32); idx = np.zeros(n, np.int32)
    steps = np.zeros(n, np.uint32); leaves = np.zeros(n, np.uint32)
    lib().oracle_intersect(C.byref(s), _fp(rays), C.c_uint32(n), _fp(t), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                           steps.ctypes.data_as(C.POINTER(C.c_uint32)), leaves.ctypes.data_as(C.POINTER(C.c_uint32)))
    return t, idx, steps, leaves


def math_eval(op, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    out = np.zeros_like(a)
    bb = np.ascontiguousarray(b, np.float32) if b is not None else None
    lib().oracle_math_eval(C.c_int(op), _fp(a), _fp(bb) if bb is not None else None, C.c_uint32(a.size), _fp(out))
    return out


def rand_base_stream(seed, n):
    st = C.c_uint64(seed)
    return [float(lib().oracle_rand_base_next(C.byref(st))) for _ in range(n)]


def brdf_probe(arrays, which, inputs):
    s = oscene(arrays)
    inputs = np.ascontiguousarray(inputs, np.float32).reshape(-1, 8)
    out = np.zeros((inputs.shape[0], 4), np.float32)
    lib().oracle_brdf_probe(C.byref(s), C.c_int(which), _fp(inputs), C.c_uint32(inputs.shape[0]), _fp(out))
    return out


BOUNCE_FLOATS = 44


def sampler_probe(arrays, which, inputs, rec, env_theta=0.0):
    """which 0 sampleMicrofacet / 1 sampleLambert / 2 sampleEnv with the recorded rnd() values `rec` (n x k) replayed."""
    s = oscene(arrays)
    inputs = np.ascontiguousarray(inputs, np.float32).reshape(-1, 4)
    rec = np.ascontiguousarray(rec, np.float32).reshape(inputs.shape[0], -1)
    out = np.zeros((inputs.shape[0], 4), np.float32)
    lib().oracle_sampler_probe(C.byref(s), C.c_int(which), _fp(inputs), _fp(rec), C.c_uint32(rec.shape[1]),
                               C.c_float(env_theta), C.c_uint32(inputs.shape[0]), _fp(out))
    return out


def bounce_probe(arrays, rays, t, index, rand_base, env_theta, rec=None, tex=None):
    """One bounce-loop iteration (tracer.fs:447-499) per (ray, hit); rec = n x 8 recorded rnd() values or None,
    tex = n x 12 recorded texture() results (diffuse.rgb, emissive.rgb, mr.rg, normal.rgb, pad) or None."""
    s = oscene(arrays)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = rays.shape[0]
    t = np.ascontiguousarray(t, np.float32).reshape(n); index = np.ascontiguousarray(index, np.int32).reshape(n)
    if rec is not None:
        rec = np.ascontiguousarray(rec, np.float32).reshape(n, 8)
    if tex is not None:
        tex = np.ascontiguousarray(tex, np.float32).reshape(n, 12)
    out = np.zeros((n, BOUNCE_FLOATS), np.float32)
    lib().oracle_bounce_probe(C.byref(s), _fp(rays), _fp(t), index.ctypes.data_as(C.POINTER(C.c_int32)),
                              C.c_float(rand_base), C.c_float(env_theta), _fp(rec) if rec is not None else None,
                              _fp(tex) if tex is not None else None, C.c_uint32(n), _fp(out))
    return out


def path_replay(arrays, pos, d, rec, rec_count, rand_base, env_theta, num_bounces, env_rec=None, tex_rec=None):
    """Stage D6: the whole path (tracer.fs main, 436-518) for rays pos / d ([n, 4] each) with the recorded rnd() values
    of the reference GLSL replayed (rec [n, stride], rec_count [n] of them valid) and, optionally, what its envSample
    returned for the path's k-th environment lookup (env_rec [n, k, 3]) and the four texture() results of its k-th loop
    iteration (tex_rec [n, k, 12]: diffuse.rgb, emissive.rgb, mr.rg, normal.rgb, pad).  Returns (clamped colour [n, 3], rnd() calls made
    [n], hash of the intersectScene hit indices [n], intersectScene calls [n], environment lookups made [n])."""
    s = oscene(arrays)
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 4)
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 4)
    n = pos.shape[0]
    rec = np.ascontiguousarray(rec, np.float32).reshape(n, -1)
    cnt = np.ascontiguousarray(rec_count, np.uint32).reshape(n)
    if env_rec is not None:
        env_rec = np.ascontiguousarray(env_rec, np.float32).reshape(n, -1, 3)
    if tex_rec is not None:
        tex_rec = np.ascontiguousarray(tex_rec, np.float32).reshape(n, -1, 12)
    col = np.zeros((n, 3), np.float32); used = np.zeros(n, np.uint32); sig = np.zeros((n, 2), np.uint32)
    env_used = np.zeros(n, np.uint32)
    u32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    lib().oracle_path_replay(C.byref(s), _fp(pos), _fp(d), C.c_uint32(n), _fp(rec), C.c_uint32(rec.shape[1]), u32(cnt),
                             C.c_float(rand_base), C.c_float(env_theta), C.c_uint32(num_bounces),
                             _fp(env_rec) if env_rec is not None else None, C.c_uint32(env_rec.shape[1] if env_rec is not None else 0),
                             _fp(tex_rec) if tex_rec is not None else None, C.c_uint32(tex_rec.shape[1] if tex_rec is not None else 0),
                             _fp(col), u32(used), u32(sig), u32(env_used))
    return col, used, sig[:, 0], sig[:, 1], env_used


def rnd_sequence(seeds, k):
    seeds = np.ascontiguousarray(seeds, np.float32).reshape(-1)
    out = np.zeros((seeds.size, k), np.float32)
    lib().oracle_rnd_sequence(_fp(seeds), C.c_uint32(seeds.size), C.c_uint32(k), _fp(out))
    return out


def draw(accum, exposure=1.0, saturation=1.0, denoise=False, max_sigma=3.0, scale=1.0):
    accum = np.ascontiguousarray(accum, np.float32)
    H, W = accum.shape[:2]
    out = np.zeros((H, W, 4), np.uint8)
    lib().oracle_draw_scaled(_fp(accum), C.c_uint32(W), C.c_uint32(H), C.c_float(exposure), C.c_float(saturation),
                             C.c_int(1 if denoise else 0), C.c_float(max_sigma), C.c_float(scale),
                             out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out
