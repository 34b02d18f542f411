"""The N-API addon's handle logic (fspt_amd/csrc/fspt_napi.c "handles"), on the CPU: the addon is built here against
tests/napi_mock/libfspt_mock.c - a stand-in for libfspt.so whose fspt_render only sleeps - so that what the ADDON does
can be tested without a device: while a renderAsync job is in flight every other call on its target throws
Error('render in flight') (a data race / use-after-free before), handles know their kind, destroyed
handles are refused, and a tracer that is dropped without close() is cleaned up by the externals' finalizers (scene after
its targets).  The same properties on the real library and a GPU: tests/test_node_host.py."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists("/usr/include/node/node_api.h"),
                                reason="node or the Node headers are missing")


@pytest.fixture(scope="module")
def report(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("napi_mock"))
    inc = os.path.join(ROOT, "include")
    mock = os.path.join(ROOT, "tests", "napi_mock")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-I" + inc, "-o", os.path.join(d, "libfspt.so"),
                           os.path.join(mock, "libfspt_mock.c"), os.path.join(mock, "libfspt_mock_stubs.c")])
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-I/usr/include/node", "-I" + inc, "-DNODE_GYP_MODULE_NAME=fspt_napi",
                           "-o", os.path.join(d, "fspt_napi.node"), os.path.join(ROOT, "fspt_amd", "csrc", "fspt_napi.c"),
                           "-L" + d, "-lfspt", "-Wl,-rpath," + d])
    out = os.path.join(d, "out.json")
    subprocess.check_call(["node", "--expose-gc", os.path.join(ROOT, "tests", "napi_handles_check.js"), d, out], timeout=120)
    return json.load(open(out))


def test_calls_during_render_async_throw(report):
    for name, msg in report["during"].items():
        assert msg == "Error: render in flight", (name, msg)
    # ... and the target is the caller's again in the promise's very first reaction
    assert report["first_reaction"] is None
    assert report["renders_seen"] == 1.0  # the one job ran, the refused calls did not


def test_handle_kinds_and_destroyed_handles(report):
    k = report["kinds"]
    assert k["scene_as_target"] == "TypeError: fspt_napi: expected a target handle"
    assert k["target_as_scene"] == "TypeError: fspt_napi: expected a scene handle"
    assert k["number_as_target"] == "TypeError: fspt_napi: expected a target handle"
    assert "still has targets" in k["scene_with_targets"]
    assert k["destroyed_target"] == "Error: fspt_napi: the target handle was destroyed"
    assert k["double_destroy"] == "Error: fspt_napi: the target handle was destroyed"
    assert report["live_after_explicit_destroy"] == 0


def test_dropped_handles_are_finalized(report):
    assert report["live_before_gc"] == 50 * 4  # 50 x (scene + 2 targets + builder), nothing destroyed by hand
    assert report["live_after_gc"] == 0        # (the mock aborts if a scene goes before one of its targets)
    assert report["live_during_dropped_job"] == 2  # the job holds its target, the target its scene
    assert report["live_after_dropped_job"] == 0


def test_multi_handles(report):
    d = report["multi_during"]
    assert d["multiReadRadiance"] == d["its_target"] == d["multiDestroy"] == "Error: render in flight"
    a = report["multi_after"]
    assert a["its_target"] is None
    assert "destroyed with the multi" in a["destroy_its_target"]
    assert a["stage_ms"] == [-1.0] * 8
    assert a["its_target_after_destroy"] == "Error: fspt_napi: the target handle was destroyed"
    assert report["live_at_end"] == 0
