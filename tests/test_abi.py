"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/fspt.h declares, and refuses to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from fspt_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("fspt.h", "fspt_multi.h", "fspt_tuning.h")  # the drop-in boundary; scheduling knobs + measurement hooks


def header_functions(names=HEADERS):
    out = set()
    for name in names:
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(fspt_[a-z_0-9]+)\s*\(", text))
    return sorted(out)


def test_boundary_header_stays_the_boundary():
    """include/fspt.h is what replaces the reference's WebGL calls + multi-GPU + the scene builder; every
    scheduling knob lives in fspt_tuning.h."""
    boundary = header_functions(("fspt.h",))
    assert not [n for n in boundary if re.search(r"set_(pipeline|pool|tail|trace_budget|deferred|memory_limit)|last_stage|live_paths", n)]
    assert len(open(os.path.join(ROOT, "include", "fspt.h")).read().split("\n")) <= 300


def test_header_symbols_exported():
    lib = C.CDLL(L.LIB_PATH)
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"


def test_binding_covers_header():
    assert sorted(L.SIGNATURES) == header_functions()


def test_abi_version():
    """header, library and both bindings agree (a stale libfspt.so is refused with a clear message at load, not with an
    AttributeError on the first new entry point)"""
    import re
    hdr = int(re.search(r"#define FSPT_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "fspt.h")).read()).group(1))
    js = int(re.search(r"const ABI_VERSION = (\d+)", open(os.path.join(ROOT, "fspt_amd", "js", "fspt.js")).read()).group(1))
    assert L.lib().fspt_abi_version() == hdr == L.ABI_VERSION == js == 4


def test_rand_base_stream_range():
    st = C.c_uint64(1)
    vals = [L.lib().fspt_rand_base_next(C.byref(st)) for _ in range(1000)]
    assert all(0.0 <= v < 10000.0 for v in vals)
    assert len(set(vals)) > 990


def test_no_cpu_fallback(small_scene):
    """Without a HIP device the product path must fail loudly, not compute."""
    lib = L.lib()
    if lib.fspt_device_count() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    d = small_scene.desc()
    rc = lib.fspt_scene_create(C.byref(d), 0, C.byref(h))
    assert rc == -2  # FSPT_E_NO_DEVICE
    assert b"no CPU fallback" in lib.fspt_last_error()
    a = np.zeros(4, np.float32)
    assert lib.fspt_math_eval(0, 0, L.fptr(a), None, 4, L.fptr(a)) == -2
    # the multi-device target: same, and its half-built state is torn down without a crash
    m = C.c_void_p()
    devs = (C.c_int * 2)(0, 1)
    assert lib.fspt_multi_create(C.byref(d), devs, 2, 64, 48, C.byref(m)) == -2 and not m.value
    assert lib.fspt_multi_create(C.byref(d), devs, 0, 64, 48, C.byref(m)) == -1  # no devices listed
    # round 6's entry points: the memory query needs a device too; the stage-time query refuses a NULL handle
    f, t = C.c_uint64(), C.c_uint64()
    assert lib.fspt_device_memory(0, C.byref(f), C.byref(t)) == -2
    assert lib.fspt_device_memory(0, None, C.byref(t)) == -1
    ms = np.zeros(8, np.float32)
    assert lib.fspt_multi_last_stage_ms(None, L.fptr(ms), 2) == -1


def test_scene_validation_errors(small_scene):
    lib = L.lib()
    h = C.c_void_p()
    assert lib.fspt_scene_create(None, 0, C.byref(h)) == -1
    d = small_scene.desc()
    d.n_bins = 0
    assert lib.fspt_scene_create(C.byref(d), 0, C.byref(h)) == -1
    # a child index that violates pre-order must be rejected before touching the device
    bad = small_scene.bvh.copy()
    bad.view(np.int32)[0] = 0  # root.left = root
    d = small_scene.desc()
    d.bvh = L.fptr(bad)
    assert lib.fspt_scene_create(C.byref(d), 0, C.byref(h)) == -1
    assert b"pre-order" in lib.fspt_last_error()


def test_builder_errors():
    lib = L.lib()
    b = C.c_void_p()
    assert lib.fspt_builder_create(C.byref(b)) == 0
    assert lib.fspt_builder_build(b, 4) == -1  # no triangles
    assert lib.fspt_builder_counts(b, None, None, None) == -6
    pd = L.PropDesc(); pd.scale = 1.0
    txt = b"v 0 0 0\nv 1 0 0\nf 1 2 5\n"
    assert lib.fspt_builder_add_obj(b, txt, len(txt), C.byref(pd)) == -5
    lib.fspt_builder_destroy(b)
