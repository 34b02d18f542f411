"""ctypes binding of libfspt.so (include/fspt.h, include/fspt_tuning.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).
There is no fallback: if the shared object is missing this module raises, and
every device entry point raises ``FsptError`` when no HIP device is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FSPT_LIB") or os.path.join(_HERE, "libfspt.so")  # FSPT_LIB: A/B builds of the same ABI
ABI_VERSION = 4  # include/fspt.h FSPT_ABI_VERSION this binding was written against (tests/test_abi.py pins the two)


class FsptError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfspt error {code}: {msg}")
        self.code = code


class SceneDesc(C.Structure):
    _fields_ = [
        ("bvh", C.POINTER(C.c_float)), ("n_nodes", C.c_uint32),
        ("tri", C.POINTER(C.c_float)), ("n_tris", C.c_uint32),
        ("mat", C.POINTER(C.c_float)),
        ("norm", C.POINTER(C.c_float)),
        ("uv", C.POINTER(C.c_float)),
        ("atlas", C.POINTER(C.c_uint8)), ("atlas_res", C.c_uint32), ("atlas_layers", C.c_uint32),
        ("env", C.POINTER(C.c_uint8)), ("env_w", C.c_uint32), ("env_h", C.c_uint32),
        ("bins", C.POINTER(C.c_uint32)), ("n_bins", C.c_uint32),
        ("leaf_size", C.c_uint32),
    ]


class CameraParams(C.Structure):
    _fields_ = [
        ("P", C.c_float * 3), ("I", C.c_float * 3), ("fov_scale", C.c_float),
        ("lens", C.c_float * 2), ("env_theta", C.c_float), ("num_bounces", C.c_uint32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "steps", "leaves", "shades", "env_lookups")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PropDesc(C.Structure):
    _fields_ = [
        ("rotate", C.POINTER(C.c_double)), ("n_rotate", C.c_uint32),
        ("scale", C.c_double), ("translate", C.c_double * 3),
        ("normals_mode", C.c_uint32),
        ("diffuse_layer", C.c_double), ("emissive_layer", C.c_double),
        ("normal_layer", C.c_double), ("mr_layer", C.c_double),
        ("emittance", C.c_double * 3),
        ("ior", C.c_double), ("dielectric", C.c_double),
    ]


class WorldTransform(C.Structure):
    _fields_ = [
        ("rotate", C.POINTER(C.c_double)), ("n_rotate", C.c_uint32), ("has_rotate", C.c_uint32),
        ("translate", C.c_double * 3), ("has_translate", C.c_uint32),
    ]


class GroupMaterial(C.Structure):
    _fields_ = [
        ("diffuse_layer", C.c_double), ("emissive_layer", C.c_double),
        ("normal_layer", C.c_double), ("mr_layer", C.c_double),
        ("emittance", C.c_double * 3),
        ("ior", C.c_double), ("dielectric", C.c_double),
    ]


# every symbol include/fspt.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
_F = C.POINTER(C.c_float)
_U32 = C.POINTER(C.c_uint32)
SIGNATURES = {
    "fspt_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(_VP)]),
    "fspt_scene_destroy": (C.c_int, [_VP]),
    "fspt_set_texture_interleave_budget": (C.c_int, [C.c_uint64]),
    "fspt_scene_depth": (C.c_int, [_VP, _U32]),
    "fspt_target_create": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    "fspt_target_destroy": (C.c_int, [_VP]),
    "fspt_target_set_viewport": (C.c_int, [_VP, C.c_uint32, C.c_uint32]),
    "fspt_target_set_shard": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint32]),
    "fspt_target_bind_accumulator": (C.c_int, [_VP, _VP]),
    "fspt_target_accumulator": (C.c_int, [_VP, C.POINTER(_VP)]),
    "fspt_target_size": (C.c_int, [_VP, _U32, _U32]),
    "fspt_camera": (C.c_int, [_VP, _F, _F, C.c_float, _F, C.c_float]),
    "fspt_set_rays": (C.c_int, [_VP, _F, _F]),
    "fspt_read_rays": (C.c_int, [_VP, _F, _F]),
    "fspt_trace": (C.c_int, [_VP, C.c_uint32, C.c_float, C.c_float, C.c_uint32]),
    "fspt_trace_test": (C.c_int, [_VP, C.c_uint32]),
    "fspt_render": (C.c_int, [_VP, C.POINTER(CameraParams), C.c_uint32, C.c_uint32, C.c_uint64]),
    "fspt_rand_base_next": (C.c_float, [C.POINTER(C.c_uint64)]),
    "fspt_clear": (C.c_int, [_VP]),
    "fspt_sync": (C.c_int, [_VP]),
    "fspt_read_radiance": (C.c_int, [_VP, _F]),
    "fspt_draw": (C.c_int, [_VP, C.c_float, C.c_float, C.c_int, C.c_float, C.POINTER(C.c_uint8)]),
    "fspt_draw_scaled": (C.c_int, [_VP, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint8)]),
    "fspt_intersect": (C.c_int, [_VP, _F, C.c_uint32, _F, C.POINTER(C.c_int32), _U32, _U32]),
    "fspt_intersect_form": (C.c_int, [_VP, C.c_int, _F, C.c_uint32, _F, C.POINTER(C.c_int32), _U32, _U32]),
    "fspt_scene_two_level_nodes": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "fspt_target_set_node_form": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_int64]),
    "fspt_enable_counters": (C.c_int, [_VP, C.c_int]),
    "fspt_get_counters": (C.c_int, [_VP, C.POINTER(Counters)]),
    "fspt_counters_reset": (C.c_int, [_VP]),
    "fspt_get_trace_lds_steps": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "fspt_math_eval": (C.c_int, [C.c_int, C.c_int, _F, _F, C.c_uint32, _F]),
    "fspt_last_kernel_ms": (C.c_int, [_VP, _F, _U32]),
    "fspt_target_set_pipeline": (C.c_int, [_VP, C.c_int, C.c_uint32]),
    "fspt_target_set_primary_form": (C.c_int, [_VP, C.c_int]),
    "fspt_target_get_primary_form": (C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "fspt_target_set_pool": (C.c_int, [_VP, C.c_uint32, C.c_int, C.c_uint32, C.c_int]),
    "fspt_target_set_trace_budget": (C.c_int, [_VP, C.c_uint32]),
    "fspt_last_stage_ms": (C.c_int, [_VP, _F, _U32]),
    "fspt_target_set_stage_timing": (C.c_int, [_VP, C.c_int]),
    "fspt_target_prepare": (C.c_int, [_VP]),
    "fspt_target_set_tail": (C.c_int, [_VP, C.c_int]),
    "fspt_target_set_deferred": (C.c_int, [_VP, C.c_int]),
    "fspt_target_live_paths": (C.c_int, [_VP, C.POINTER(C.c_double), C.c_uint32]),
    "fspt_target_set_memory_limit": (C.c_int, [_VP, C.c_uint64]),
    "fspt_target_path_state_bytes": (C.c_int, [_VP, C.POINTER(C.c_uint64), _U32]),
    "fspt_multi_create": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    "fspt_multi_destroy": (C.c_int, [_VP]),
    "fspt_multi_target": (C.c_int, [_VP, C.c_uint32, C.POINTER(_VP)]),
    "fspt_multi_camera": (C.c_int, [_VP, _F, _F, C.c_float, _F, C.c_float]),
    "fspt_multi_trace": (C.c_int, [_VP, C.c_uint32, C.c_float, C.c_float, C.c_uint32]),
    "fspt_multi_render": (C.c_int, [_VP, C.POINTER(CameraParams), C.c_uint32, C.c_uint32, C.c_uint64]),
    "fspt_multi_clear": (C.c_int, [_VP]),
    "fspt_multi_sync": (C.c_int, [_VP]),
    "fspt_multi_read_radiance": (C.c_int, [_VP, _F]),
    "fspt_multi_draw": (C.c_int, [_VP, C.c_float, C.c_float, C.c_int, C.c_float, C.POINTER(C.c_uint8)]),
    "fspt_multi_last_gather_bytes": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "fspt_multi_size": (C.c_int, [_VP, _U32, _U32]),
    "fspt_multi_peer_access": (C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_int)]),
    "fspt_multi_set_exchange": (C.c_int, [_VP, C.c_int]),
    "fspt_multi_get_exchange": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fspt_target_shard_slots": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "fspt_target_pack_tiles": (C.c_int, [_VP, _VP, C.c_uint32]),
    "fspt_target_unpack_tiles": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.c_uint32]),
    "fspt_builder_create": (C.c_int, [C.POINTER(_VP)]),
    "fspt_builder_destroy": (C.c_int, [_VP]),
    "fspt_builder_add_obj": (C.c_int, [_VP, C.c_char_p, C.c_size_t, C.POINTER(PropDesc)]),
    "fspt_builder_parse_obj": (C.c_int, [_VP, C.c_char_p, C.c_size_t, C.POINTER(PropDesc), C.POINTER(WorldTransform),
                                         C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, _U32]),
    "fspt_builder_group_info": (C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_char_p), _U32, C.POINTER(C.c_int32)]),
    "fspt_builder_mtllib_name": (C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_char_p)]),
    "fspt_builder_commit_obj": (C.c_int, [_VP, C.POINTER(GroupMaterial), C.c_uint32]),
    "fspt_builder_normalize": (C.c_int, [_VP, C.c_double]),
    "fspt_builder_build": (C.c_int, [_VP, C.c_uint32]),
    "fspt_builder_counts": (C.c_int, [_VP, _U32, _U32, _U32]),
    "fspt_builder_autofocus": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fspt_builder_get": (C.c_int, [_VP, _F, _F, _F, _F, _F]),
    "fspt_env_bins": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, _U32, C.c_uint32, _U32]),
    "fspt_multi_last_stage_ms": (C.c_int, [_VP, _F, C.c_uint32]),
    "fspt_device_memory": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "fspt_last_error": (C.c_char_p, []),
    "fspt_abi_version": (C.c_int, []),
    "fspt_device_count": (C.c_int, []),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 (same soname as /opt/rocm's, but
    libtorch_hip asks for it by file name): if libfspt pulled in /opt/rocm's copy first, a later `import torch`
    would load a second runtime and find "No HIP GPUs".  Loading torch's copy first (without importing torch)
    makes both bind to the same one, whatever the import order; bound accumulators (torch tensors) then live
    in the runtime that launches the kernels."""
    if os.environ.get("FSPT_OWN_HIP_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # no torch: libfspt uses the system runtime


def lib():
    """Load libfspt.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FsptError(-100, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _share_hip_runtime_with_torch()
        l = C.CDLL(LIB_PATH)
        try:
            l.fspt_abi_version.restype = C.c_int
            have = l.fspt_abi_version()
        except AttributeError:
            have = None
        if have != ABI_VERSION:
            raise FsptError(-101, f"{LIB_PATH} has ABI version {have}, this binding needs {ABI_VERSION}: stale build - "
                                  "run `python -c 'import __graft_entry__ as g; g.build()'`")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise FsptError(rc, lib().fspt_last_error().decode("utf-8", "replace"))


def fptr(a):
    return a.ctypes.data_as(_F)


def u32ptr(a):
    return a.ctypes.data_as(_U32)


def u8ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))
