"""Accuracy contract of the fspt-math primitives against IEEE float64 (numpy), independent of the oracle.

oracle/oracle_math.h and fspt_amd/csrc/fspt_math.hpp implement the same written spec, so their bitwise
equality (tests/test_parity_gpu.py::test_math_bitwise) cannot reveal an error they share.  These bounds can: every
primitive is compared with numpy's float64 result on the argument ranges the path tracer uses, in units in the last
place of the float32 result (ULP) or absolutely where the function crosses zero."""
import numpy as np

OPS = {"sin": 0, "cos": 1, "atan2": 2, "asin": 3, "exp2": 4, "div": 5, "sqrt": 6, "rnd": 7, "fract": 8, "log2": 9, "pow": 10}


def inputs(name, n=1 << 16, seed=0):
    rng = np.random.default_rng(seed)
    if name in ("sin", "cos"):
        # tracer.fs's rnd() calls sin() of the running seed: arguments reach a few million
        a = np.concatenate([rng.uniform(-10, 10, n // 4), rng.uniform(-3e4, 3e4, n // 4), rng.uniform(-3e6, 3e6, n // 4),
                            rng.normal(size=n // 4) * 1e-3])
        return a.astype(np.float32), None
    if name == "atan2":  # envSample: atan(dir.z, dir.x) of unit vectors, plus axis-aligned and tiny components
        a = np.concatenate([rng.normal(size=n - 8), [0, 0, 1, -1, 1e-30, -1e-30, 1, -1]])
        b = np.concatenate([rng.normal(size=n - 8), [1, -1, 0, 0, 1, 1, 1e-30, -1e-30]])
        return a.astype(np.float32), b.astype(np.float32)
    if name == "asin":
        return np.concatenate([rng.uniform(-1, 1, n - 4), [-1, 1, 0, 1e-20]]).astype(np.float32), None
    if name == "exp2":  # envColor: 2^(a*255 - 128); draw: pow via exp2
        return rng.uniform(-126, 127, n).astype(np.float32), None
    if name == "log2":
        return np.concatenate([rng.uniform(1e-6, 4, n // 2), 10 ** rng.uniform(-37, 38, n // 2)]).astype(np.float32), None
    if name == "pow":   # draw.fs gamma: pow(x in [0,1], 0.4545)
        return rng.uniform(1e-6, 1, n).astype(np.float32), np.full(n, 0.454545, np.float32)
    if name == "sqrt":
        return np.abs(rng.normal(size=n) * 10 ** rng.uniform(-18, 18, n)).astype(np.float32), None
    if name == "div":
        return ((rng.normal(size=n) * 10 ** rng.uniform(-6, 6, n)).astype(np.float32),
                (rng.normal(size=n) * 10 ** rng.uniform(-6, 6, n)).astype(np.float32))
    raise KeyError(name)


def exact(name, a, b):
    a64 = a.astype(np.float64)
    b64 = b.astype(np.float64) if b is not None else None
    return {"sin": np.sin, "cos": np.cos, "asin": np.arcsin, "exp2": np.exp2, "log2": np.log2, "sqrt": np.sqrt}[name](a64) \
        if b is None else {"atan2": np.arctan2, "pow": np.power, "div": np.divide}[name](a64, b64)


# name -> (max ULP error, max absolute error); a result passes if it is within EITHER bound (ULPs are meaningless next to
# a zero crossing: sin(3e6) with |result| ~ 1e-4 is accurate to 1e-7 absolutely, not to an ULP).
BOUNDS = {  # measured on 65 536 inputs each: sin 1.41 / cos 1.47 / atan2 2.90 / asin 2.24 / exp2 0.90 / log2 1.08 / pow 4.95 ULP
    "sin": (2.0, 1.2e-7), "cos": (2.0, 1.2e-7),  # float64 range reduction + Cephes sinf/cosf kernels
    "atan2": (3.5, 3.0e-7), "asin": (3.0, 2.0e-7),
    "exp2": (1.0, 0.0), "log2": (1.5, 0.0),
    "pow": (6.0, 1.0e-7),                          # exp2(y * log2(x)): feeds draw.fs's 8-bit gamma only
    "sqrt": (0.5, 0.0), "div": (0.5, 0.0),        # IEEE correctly rounded
}


def check(name, got, a, b):
    """Assert `got` (float32 results for inputs a, b) meets BOUNDS[name]; returns (max ulp, max abs) over the inputs
    that needed the other bound."""
    want = exact(name, a, b)
    w32 = want.astype(np.float32)
    ulp = np.spacing(np.maximum(np.abs(w32), np.float32(np.finfo(np.float32).tiny))).astype(np.float64)
    err = np.abs(got.astype(np.float64) - want)
    max_ulp, max_abs = BOUNDS[name]
    ok = (err <= max_ulp * ulp) | (err <= max_abs)
    assert ok.all(), (name, int((~ok).sum()), float((err / ulp)[~ok].max()), float(err[~ok].max()), a[~ok][:4], got[~ok][:4])
    return float((err / ulp)[err > max_abs].max(initial=0.0)), float(err[err > max_ulp * ulp].max(initial=0.0))
