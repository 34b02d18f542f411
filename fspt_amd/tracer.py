"""Host-side mirror of the reference's frame driver (main.js:741-857) on top of
libfspt: same method names, same argument meaning, same call order.

    pt = PathTracer(scene_arrays, width, height)
    pt.eye, pt.dir, pt.fovScale, pt.lensFeatures, pt.envTheta   # main.js:67-74
    pt.drawCamera(randBase); pt.drawTracer(i, randBase)          # one sample
    pt.tick()                                                    # main.js:838-857
    pt.clear()                                                   # main.js:826-836
    pt.readRadiance() -> float32 [H, W, 4], row 0 = bottom       # what draw.fs:87 reads

All compute happens in the HIP kernels; a missing library or device raises
FsptError (there is no CPU path here).
"""
import ctypes as C

import numpy as np

from . import _lib as L


def set_texture_interleave_budget(nbytes):
    """Memory later Scene()s may spend on interleaved material textures (fspt_set_texture_interleave_budget)."""
    L.check(L.lib().fspt_set_texture_interleave_budget(int(nbytes)))

def device_memory(device=0):
    """(free, total) bytes of a device as the HIP runtime reports them (fspt_device_memory): what a host sizes its batches
    against, and how the tests check that dropped tracers give their memory back."""
    f, t = C.c_uint64(), C.c_uint64()
    L.check(L.lib().fspt_device_memory(int(device), C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


PIPELINES = {"megakernel": 0, "wavefront": 1, "stream": 2}  # fspt_target_set_pipeline codes


class Scene:
    """Device-resident scene (initBVH's texture uploads, main.js:408-437,548-560)."""

    def __init__(self, arrays, device=0):
        self.arrays = arrays
        self.device = device
        self._h = C.c_void_p()
        desc = arrays.desc()
        L.check(L.lib().fspt_scene_create(C.byref(desc), device, C.byref(self._h)))

    @property
    def depth(self):
        d = C.c_uint32()
        L.check(L.lib().fspt_scene_depth(self._h, C.byref(d)))
        return d.value

    def intersect(self, rays, two_level=False):
        """intersectScene (tracer.fs:366-404) for rays float32 [n, 6] -> t, index, steps, leaves.  two_level: walk the
        128-byte two-level nodes (include/fspt_tuning.h: fspt_target_set_node_form) - same results, same counts."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        t = np.zeros(n, np.float32); idx = np.zeros(n, np.int32)
        steps = np.zeros(n, np.uint32); leaves = np.zeros(n, np.uint32)
        L.check(L.lib().fspt_intersect_form(self._h, 1 if two_level else 0, L.fptr(rays), n, L.fptr(t),
                                            idx.ctypes.data_as(C.POINTER(C.c_int32)), L.u32ptr(steps), L.u32ptr(leaves)))
        return t, idx, steps, leaves

    def two_level_nodes(self):
        """(the scene has two-level nodes, their bytes): fspt_scene_create builds them when every box of the tree is the
        exact union of its children's boxes (every tree bvh.js builds)."""
        yes = C.c_int(); b = C.c_uint64()
        L.check(L.lib().fspt_scene_two_level_nodes(self._h, C.byref(yes), C.byref(b)))
        return bool(yes.value), int(b.value)

    def close(self):
        if self._h:
            L.lib().fspt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PathTracer:
    NUM_BOUNCES = 4  # tracer.fs:9

    def __init__(self, scene, width, height, device=0, num_bounces=None):
        self.scene = scene if isinstance(scene, Scene) else Scene(scene, device)
        self.resolution = (int(width), int(height))
        self._t = C.c_void_p()
        L.check(L.lib().fspt_target_create(self.scene._h, self.resolution[0], self.resolution[1], C.byref(self._t)))
        # main.js:67-74 defaults
        self.fovScale = 0.5
        self.envTheta = 0.0
        self.dir = [0.0, 0.0, -1.0]
        self.eye = [0.0, 0.0, 2.0]
        self.lensFeatures = [1.0 - 1.0 / 2.0, 0.02]
        self.num_bounces = self.NUM_BOUNCES if num_bounces is None else int(num_bounces)
        self.pingpong = 0
        self._rng = C.c_uint64(1)
        self._keep = None

    # ---- configuration -----------------------------------------------------
    def set_camera(self, P, I, fov_scale=0.5, env_theta=0.0, focal_depth=2.0, aperture=0.02, **_):
        self.eye = [float(x) for x in P]
        self.dir = [float(x) for x in I]
        self.fovScale = float(fov_scale)
        self.envTheta = float(env_theta)
        self.lensFeatures = [1.0 - 1.0 / float(focal_depth), float(aperture)]

    def seed(self, s):
        """Seed of the host PRNG that replaces Math.random()*10000 (main.js:748,777)."""
        if int(s) == 0:
            raise ValueError("xorshift seed must be non-zero")
        self._rng = C.c_uint64(int(s))

    def next_rand_base(self):
        return float(L.lib().fspt_rand_base_next(C.byref(self._rng)))

    def set_shard(self, shard, n_shards, tile=32):
        L.check(L.lib().fspt_target_set_shard(self._t, shard, n_shards, tile))

    def set_viewport(self, w=0, h=0):
        """gl.viewport(0, 0, w, h) of drawCamera / drawTracer (main.js:744,761); 0, 0 = the whole target.  The
        reference uses resolution * 0.25 while the camera moves (resScale, main.js:840)."""
        L.check(L.lib().fspt_target_set_viewport(self._t, int(w), int(h)))

    def bind_accumulator(self, device_ptr, keep=None):
        """Accumulate into caller-owned device memory (e.g. a torch tensor) so a
        collective can run on it in place; `keep` is held to keep it alive."""
        self._keep = keep
        L.check(L.lib().fspt_target_bind_accumulator(self._t, C.c_void_p(device_ptr)))
        self.bound_ptr = int(device_ptr)  # what pack_tiles / unpack_tiles read and write (distributed.TileGather checks it)

    # ---- the two ends of a one-process-per-GPU read-out exchange (include/fspt.h) -----------------------------
    def shard_slots(self, shard, n_shards):
        """Entries of shard `shard`'s packed pixel array: owned tiles x tile^2."""
        n = C.c_uint64()
        L.check(L.lib().fspt_target_shard_slots(self._t, int(shard), int(n_shards), C.byref(n)))
        return int(n.value)

    def pack_tiles(self, packed_ptr, channels=4):
        """This target's own pixels -> packed[slots][channels] (device pointer).  Blocking."""
        L.check(L.lib().fspt_target_pack_tiles(self._t, C.c_void_p(packed_ptr), int(channels)))

    def unpack_tiles(self, packed_ptr, shard, n_shards, channels=4):
        """Packed pixels of shard `shard` of `n_shards` -> the accumulator (channels 3: alpha = 1).  Blocking."""
        L.check(L.lib().fspt_target_unpack_tiles(self._t, C.c_void_p(packed_ptr), int(shard), int(n_shards), int(channels)))

    def set_pipeline(self, pipeline, batch_ticks=0):
        """'wavefront' (batches), 'stream' (fixed pool of live paths) or 'megakernel'; results are bit-identical
        (include/fspt_tuning.h)."""
        code = PIPELINES.get(pipeline, pipeline)
        L.check(L.lib().fspt_target_set_pipeline(self._t, int(code), int(batch_ticks)))

    def set_pool(self, paths=0, drain=-1, max_iterations=0, overlap=-1):
        """Stream scheduler: live paths per state set (0 = default), drain iterations (-1 = default), iteration cap
        (0 = none; test hook), second HIP stream for plan / primary / resolve (-1 = default).  include/fspt_tuning.h:
        fspt_target_set_pool."""
        L.check(L.lib().fspt_target_set_pool(self._t, int(paths), int(drain), int(max_iterations), int(overlap)))

    def set_primary_form(self, form=0):
        """k_wf_primary's traversal phase: 1 one ray per lane, 2 per-lane refill, 0 (default) measured and chosen by the
        library per batch size (include/fspt_tuning.h)."""
        L.check(L.lib().fspt_target_set_primary_form(self._t, int(form)))

    def primary_form(self, batch_ticks):
        """(form the next batch of that size uses, [ms per sample of form 1, form 2] measured so far or -1)."""
        f = C.c_int()
        ms = (C.c_double * 2)()
        L.check(L.lib().fspt_target_get_primary_form(self._t, int(batch_ticks), C.byref(f), ms))
        return int(f.value), [float(ms[0]), float(ms[1])]

    def set_node_form(self, primary=-1, trace=-1, tail=-1, trace_below=-1):
        """Node form per kernel class: -1 the library's choice, 0 the 64-byte nodes, 1 the two-level nodes (two traversal
        steps per memory round trip; bit-identical results); tail also 2 = adaptive (two-level once a wave's list is used up).  trace_below: the library's choice for a trace launch is
        two-level when it expects fewer paths than this (include/fspt_tuning.h)."""
        L.check(L.lib().fspt_target_set_node_form(self._t, int(primary), int(trace), int(tail), int(trace_below)))

    def set_trace_budget(self, steps):
        """Traversal steps a starved trace wave walks on before it suspends its rays (0 = never; include/fspt_tuning.h)."""
        L.check(L.lib().fspt_target_set_trace_budget(self._t, int(steps)))

    def prepare(self):
        """Allocate the pipeline's path-state buffers now (not lazily inside the first render)."""
        L.check(L.lib().fspt_target_prepare(self._t))

    def set_memory_limit(self, nbytes):
        """Cap the wavefront path state of this target (bytes, 0 = none); a batch that does not fit is halved."""
        L.check(L.lib().fspt_target_set_memory_limit(self._t, int(nbytes)))

    def path_state_bytes(self):
        """(bytes of path state currently allocated, batch size in use)."""
        b = C.c_uint64(); n = C.c_uint32()
        L.check(L.lib().fspt_target_path_state_bytes(self._t, C.byref(b), C.byref(n)))
        return b.value, n.value

    def set_stage_timing(self, on=True):
        """HIP event pairs around every launch (what last_stage_ms reads); off: ~1.3 % faster 20-tick regions, zeros there."""
        L.check(L.lib().fspt_target_set_stage_timing(self._t, 1 if on else 0))

    def last_stage_ms(self):
        ms = (C.c_float * 5)(); n = (C.c_uint32 * 5)()
        L.check(L.lib().fspt_last_stage_ms(self._t, ms, n))
        return {k: (ms[i], n[i]) for i, k in enumerate(("primary", "trace", "logic", "resolve", "tail"))}

    def set_deferred(self, on=True):
        """Two-call ticks (drawCamera + drawTracer) are recorded and run in batches at the next read-out (default);
        False: every drawTracer executes at once."""
        L.check(L.lib().fspt_target_set_deferred(self._t, 1 if on else 0))

    def live_paths(self, n_rounds=12):
        """Fraction of the batch's samples still alive after round r (index r; r = 1 is the primary launch)."""
        f = (C.c_double * n_rounds)()
        L.check(L.lib().fspt_target_live_paths(self._t, f, n_rounds))
        return list(f)

    def set_tail(self, round=-1):
        """-1: adaptive (default), 0: never, r >= 1: the tail kernel takes over after wavefront round r."""
        L.check(L.lib().fspt_target_set_tail(self._t, int(round)))

    def enable_counters(self, on=True):
        """0 / False: off.  1 / True: count the reference algorithm's work (equals the oracle's counters).
        2: count what the production kernels really do (NEE shadow rays stop at the first hit)."""
        L.check(L.lib().fspt_enable_counters(self._t, int(on)))

    # ---- the reference's draw calls ------------------------------------------
    def drawCamera(self, randBase):
        P = (C.c_float * 3)(*self.eye); I = (C.c_float * 3)(*self.dir); lens = (C.c_float * 2)(*self.lensFeatures)
        L.check(L.lib().fspt_camera(self._t, P, I, self.fovScale, lens, float(randBase)))

    def drawTracer(self, i, randBase):
        L.check(L.lib().fspt_trace(self._t, int(i), float(randBase), self.envTheta, self.num_bounces))

    def drawTracerTest(self, i):
        """drawTracer with bvh_test.fs (`mode=test`, main.js:879-883): traversal-step heat map."""
        L.check(L.lib().fspt_trace_test(self._t, int(i)))

    def tick(self):
        """One iteration of main.js:838-857 (camera draw, trace draw, pingpong++)."""
        self.drawCamera(self.next_rand_base())
        self.drawTracer(self.pingpong, self.next_rand_base())
        self.pingpong += 1

    def render(self, n_ticks):
        """n_ticks fused ticks (ray generation inside the path kernel), same
        randBase stream and results as n_ticks x tick()."""
        cp = L.CameraParams()
        cp.P = (C.c_float * 3)(*self.eye); cp.I = (C.c_float * 3)(*self.dir)
        cp.fov_scale = self.fovScale; cp.lens = (C.c_float * 2)(*self.lensFeatures)
        cp.env_theta = self.envTheta; cp.num_bounces = self.num_bounces
        L.check(L.lib().fspt_render(self._t, C.byref(cp), self.pingpong, int(n_ticks), self._rng.value))
        for _ in range(2 * int(n_ticks)):  # advance the host stream like the kernel did
            self.next_rand_base()
        self.pingpong += int(n_ticks)

    def clear(self):
        L.check(L.lib().fspt_clear(self._t))
        self.pingpong = 0
        L.check(L.lib().fspt_counters_reset(self._t))

    def sync(self):
        L.check(L.lib().fspt_sync(self._t))

    # ---- read-back -------------------------------------------------------------
    def setRays(self, pos, dir):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1)
        dir = np.ascontiguousarray(dir, dtype=np.float32).reshape(-1)
        n = self.resolution[0] * self.resolution[1] * 4
        if pos.size != n or dir.size != n:
            raise ValueError("ray buffers must be W*H*4 floats")
        L.check(L.lib().fspt_set_rays(self._t, L.fptr(pos), L.fptr(dir)))

    def readRays(self):
        W, H = self.resolution
        pos = np.zeros((H, W, 4), np.float32); d = np.zeros((H, W, 4), np.float32)
        L.check(L.lib().fspt_read_rays(self._t, L.fptr(pos), L.fptr(d)))
        return pos, d

    def readRadiance(self):
        W, H = self.resolution
        out = np.zeros((H, W, 4), np.float32)
        L.check(L.lib().fspt_read_radiance(self._t, L.fptr(out)))
        return out

    def draw(self, exposure=1.0, saturation=1.0, denoise=False, max_sigma=3.0, scale=1.0):
        """drawQuad (main.js:809-824) / draw.fs: tonemapped RGBA8 [H, W, 4], row 0 = bottom.  scale = draw.fs's
        `scale` uniform (resScale: 0.25 while the camera moves, main.js:819,840)."""
        W, H = self.resolution
        out = np.zeros((H, W, 4), np.uint8)
        L.check(L.lib().fspt_draw_scaled(self._t, float(exposure), float(saturation), 1 if denoise else 0,
                                         float(max_sigma), float(scale), L.u8ptr(out)))
        return out

    def counters(self):
        c = L.Counters()
        L.check(L.lib().fspt_get_counters(self._t, C.byref(c)))
        return c.as_dict()

    def last_kernel_ms(self):
        ms = C.c_float(); n = C.c_uint32()
        L.check(L.lib().fspt_last_kernel_ms(self._t, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        """Destroy the target.  Recorded (deferred) ticks are executed first: fspt_target_destroy itself drops them - it
        never writes to a caller-owned accumulator, which may be gone by then - but here `_keep` still holds the bound
        buffer, so a host that binds a tensor, ticks and closes finds every tick in its tensor."""
        if self._t:
            L.lib().fspt_sync(self._t)  # (an error here must not keep the target alive: destroy follows regardless)
            L.lib().fspt_target_destroy(self._t)
            self._t = C.c_void_p()
            self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiPathTracer:
    """PathTracer's frame driver over several GPUs from ONE host thread (include/fspt.h: fspt_multi_*): every device
    traces every len(devices)-th 32x32 tile of the same frame, nothing moves between devices while rendering, and
    readRadiance() / draw() gather the tiles onto devices[0] with peer-to-peer copies.  Same attribute and method
    names as PathTracer; the result is bit-identical to a single-GPU render."""
    NUM_BOUNCES = PathTracer.NUM_BOUNCES

    def __init__(self, arrays, width, height, devices=(0,), num_bounces=None):
        self.arrays = arrays
        self.devices = [int(d) for d in devices]
        self.resolution = (int(width), int(height))
        self._m = C.c_void_p()
        desc = arrays.desc()
        devs = (C.c_int * len(self.devices))(*self.devices)
        L.check(L.lib().fspt_multi_create(C.byref(desc), devs, len(self.devices), self.resolution[0], self.resolution[1],
                                          C.byref(self._m)))
        self.fovScale = 0.5
        self.envTheta = 0.0
        self.dir = [0.0, 0.0, -1.0]
        self.eye = [0.0, 0.0, 2.0]
        self.lensFeatures = [1.0 - 1.0 / 2.0, 0.02]
        self.num_bounces = self.NUM_BOUNCES if num_bounces is None else int(num_bounces)
        self.pingpong = 0
        self._rng = C.c_uint64(1)

    set_camera = PathTracer.set_camera
    seed = PathTracer.seed
    next_rand_base = PathTracer.next_rand_base

    def _targets(self):
        for i in range(len(self.devices)):
            t = C.c_void_p()
            L.check(L.lib().fspt_multi_target(self._m, i, C.byref(t)))
            yield t

    def set_pipeline(self, pipeline, batch_ticks=0):
        code = PIPELINES.get(pipeline, pipeline)
        for t in self._targets():
            L.check(L.lib().fspt_target_set_pipeline(t, int(code), int(batch_ticks)))

    def set_tail(self, round=-1):
        for t in self._targets():
            L.check(L.lib().fspt_target_set_tail(t, int(round)))

    def prepare(self):
        for t in self._targets():
            L.check(L.lib().fspt_target_prepare(t))

    def drawCamera(self, randBase):
        P = (C.c_float * 3)(*self.eye); I = (C.c_float * 3)(*self.dir); lens = (C.c_float * 2)(*self.lensFeatures)
        L.check(L.lib().fspt_multi_camera(self._m, P, I, self.fovScale, lens, float(randBase)))

    def drawTracer(self, i, randBase):
        L.check(L.lib().fspt_multi_trace(self._m, int(i), float(randBase), self.envTheta, self.num_bounces))

    tick = PathTracer.tick

    def render(self, n_ticks):
        cp = L.CameraParams()
        cp.P = (C.c_float * 3)(*self.eye); cp.I = (C.c_float * 3)(*self.dir)
        cp.fov_scale = self.fovScale; cp.lens = (C.c_float * 2)(*self.lensFeatures)
        cp.env_theta = self.envTheta; cp.num_bounces = self.num_bounces
        L.check(L.lib().fspt_multi_render(self._m, C.byref(cp), self.pingpong, int(n_ticks), self._rng.value))
        for _ in range(2 * int(n_ticks)):
            self.next_rand_base()
        self.pingpong += int(n_ticks)

    def clear(self):
        L.check(L.lib().fspt_multi_clear(self._m))
        self.pingpong = 0

    def sync(self):
        L.check(L.lib().fspt_multi_sync(self._m))

    def readRadiance(self):
        W, H = self.resolution
        out = np.zeros((H, W, 4), np.float32)
        L.check(L.lib().fspt_multi_read_radiance(self._m, L.fptr(out)))
        return out

    def draw(self, exposure=1.0, saturation=1.0, denoise=False, max_sigma=3.0):
        W, H = self.resolution
        out = np.zeros((H, W, 4), np.uint8)
        L.check(L.lib().fspt_multi_draw(self._m, float(exposure), float(saturation), 1 if denoise else 0, float(max_sigma),
                                        L.u8ptr(out)))
        return out

    def peer_access(self, i):
        """How target i's tiles reach devices[0]: bit 0 = its device can write devices[0]'s memory directly (the
        direction the gather copy runs), bit 1 = the reverse; 0 = staged through the host."""
        m = C.c_int()
        L.check(L.lib().fspt_multi_peer_access(self._m, int(i), C.byref(m)))
        return m.value

    EXCHANGES = {"peer": 0, "rccl_gather": 1, "rccl_reduce": 2}

    def set_exchange(self, mode):
        """Read-out exchange: 'peer' (hipMemcpyPeerAsync of packed tiles, default), 'rccl_gather' (ncclSend / ncclRecv of
        the same tiles) or 'rccl_reduce' (ncclReduce(SUM) of own-tiles-only frames); RCCL needs distinct devices."""
        L.check(L.lib().fspt_multi_set_exchange(self._m, int(self.EXCHANGES.get(mode, mode))))

    def exchange(self):
        """(mode code, RCCL version or 0 when RCCL is not loaded)."""
        m = C.c_int(); v = C.c_int()
        L.check(L.lib().fspt_multi_get_exchange(self._m, C.byref(m), C.byref(v)))
        return m.value, v.value

    def stage_ms(self):
        """Per device [render, pack, transfer, scatter] milliseconds of the most recent render + read-out
        (fspt_multi_last_stage_ms; -1 = the stage did not run on that device).  Blocking."""
        import numpy as np
        n = len(self.devices)
        out = np.zeros((n, 4), np.float32)
        L.check(L.lib().fspt_multi_last_stage_ms(self._m, L.fptr(out), n))
        return out

    def last_gather_bytes(self):
        b = C.c_uint64()
        L.check(L.lib().fspt_multi_last_gather_bytes(self._m, C.byref(b)))
        return b.value

    def close(self):
        if self._m:
            L.lib().fspt_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bytes_per_sample(counters):
    """Algorithmic bytes per sample on the REFERENCE layout:
    60 B per traversal step (12 B header + 2x24 B child boxes, tracer.fs:374-378),
    144 B per leaf visit (4 x 36 B, tracer.fs:355-364), 280 B per shading event
    (tracer.fs:447-460), 16 B per environment lookup (tracer.fs:410-419), 64 B of
    ray + accumulator traffic per sample (tracer.fs:439,516-517)."""
    s = max(1, counters["samples"])
    return (60.0 * counters["steps"] + 144.0 * counters["leaves"] + 280.0 * counters["shades"]
            + 16.0 * counters["env_lookups"]) / s + 64.0
