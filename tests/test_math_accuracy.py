"""Oracle math vs IEEE float64 (not the HIP path): the CPU half of the accuracy contract in tests/mathref.py.
The GPU half (test_parity_gpu.py::test_math_accuracy_vs_float64) checks the device results against the same bounds
directly, so a defect shared by the twin headers oracle_math.h / fspt_math.hpp cannot hide behind their bit equality."""
import numpy as np
import pytest

import mathref
import oracle as O


@pytest.mark.parametrize("name", sorted(mathref.BOUNDS))
def test_oracle_math_accuracy(name):
    a, b = mathref.inputs(name)
    got = O.math_eval(mathref.OPS[name], a, b)
    mathref.check(name, got, a, b)


def test_rnd_is_fract_of_scaled_sin():
    """rnd (tracer.fs:181): fract(sin(seed += 0.2113...) * 43758.5453) built from the checked primitives; in [0, 1)."""
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 3e6, 1 << 14).astype(np.float32)
    r = O.math_eval(mathref.OPS["rnd"], a, None)
    assert (r >= 0).all() and (r < 1).all()
    s = O.math_eval(mathref.OPS["sin"], (a + np.float32(0.211324865405187)).astype(np.float32), None)
    x = (s * np.float32(43758.5453123)).astype(np.float32)
    assert np.array_equal(r, (x - np.floor(x)).astype(np.float32))
