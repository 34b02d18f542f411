"""Host-side checks of the oracle restatement itself (no GPU)."""
import numpy as np

import oracle as O
from conftest import random_rays


def test_oracle_deterministic(small_scene, camera):
    W, H = 48, 32
    a = np.zeros((H, W, 4), np.float32); b = np.zeros((H, W, 4), np.float32)
    for acc in (a, b):
        O.render(small_scene, W, H, camera["P"], camera["I"], camera["fov_scale"], camera["lens"],
                 camera["env_theta"], 4, 0, 3, 7, acc)
    assert np.array_equal(a, b)
    assert np.isfinite(a).all() and (a[..., 3] == 1).all()
    assert (a[..., :3] >= 0).all() and (a[..., :3] <= 1024).all()


def test_oracle_running_mean(small_scene, camera):
    """tracer.fs:515-517: n ticks of running mean == mean of the n single-tick images (to rounding)."""
    W, H = 32, 24
    rbs = O.rand_base_stream(5, 8)
    acc = np.zeros((H, W, 4), np.float32)
    singles = []
    for k in range(4):
        pos, d = O.camera(W, H, camera["P"], camera["I"], camera["fov_scale"], camera["lens"], rbs[2 * k])
        O.trace(small_scene, W, H, pos, d, k, rbs[2 * k + 1], camera["env_theta"], 4, acc)
        one = np.zeros((H, W, 4), np.float32)
        O.trace(small_scene, W, H, pos, d, 0, rbs[2 * k + 1], camera["env_theta"], 4, one)
        singles.append(one)
    np.testing.assert_allclose(acc[..., :3], np.mean(singles, 0)[..., :3], rtol=1e-5, atol=1e-6)


def test_oracle_intersect_vs_bruteforce(small_scene):
    """intersectScene == brute-force Moller-Trumbore over all triangles (float64 check)."""
    rays = random_rays(small_scene, 400, seed=3)
    t, idx, steps, leaves = O.intersect(small_scene, rays)
    tri = small_scene.tri.reshape(-1, 3, 3).astype(np.float64)
    o = rays[:, :3].astype(np.float64); d = rays[:, 3:].astype(np.float64)
    e1 = tri[:, 1] - tri[:, 0]; e2 = tri[:, 2] - tri[:, 0]
    hits = 0
    for i in range(len(rays)):
        p = np.cross(d[i], e2); det = (e1 * p).sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = o[i] - tri[:, 0]
            u = (tv * p).sum(1) * inv
            q = np.cross(tv, e1)
            v = (q * d[i]).sum(1) * inv
            dist = (e2 * q).sum(1) * inv
        ok = (np.abs(det) >= 1e-6) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (dist > 1e-6)
        if ok.any():
            best = dist[ok].min()
            if idx[i] >= 0:
                assert abs(t[i] - best) <= 1e-4 * max(1.0, best)
                hits += 1
            else:  # fp32 vs fp64 edge graze
                assert best > 0
        else:
            assert idx[i] == -1 and t[i] == np.float32(100000.0)
    assert hits > 50
    assert (steps >= 1).all() and (leaves <= steps).all()


def test_shards_partition_frame(small_scene, camera):
    """tile-sharded renders sum to the single render, bit for bit."""
    W, H = 80, 48
    full = np.zeros((H, W, 4), np.float32)
    args = (small_scene, W, H, camera["P"], camera["I"], camera["fov_scale"], camera["lens"], camera["env_theta"], 4, 0, 2, 3)
    O.render(*args, full)
    total = np.zeros_like(full)
    for s in range(3):
        part = np.zeros_like(full)
        O.render(*args, part, shard=s, n_shards=3, tile=16)
        total += part
    assert np.array_equal(total, full)


def test_oracle_root_leaf_and_black_env():
    """2-triangle scene: root is a leaf; rays that miss see the black default environment."""
    from fspt_amd import scene as S
    props = [{"path": "q.obj", "scale": 3, "rotate": [], "translate": [0, -0.5, 0], "emittance": [0, 0, 0],
              "diffuse": [0.8, 0.7, 0.6], "normals": "flat"}]
    arrays = S.build_scene(props, {"q.obj": S.QUAD_OBJ})
    assert arrays.n_nodes == 1
    rays = np.array([[0, 1, 0, 0, -1, 0], [0, 1, 0, 0, 1, 0], [10, 1, 0, 0, -1, 0]], np.float32)
    t, idx, steps, leaves = O.intersect(arrays, rays)
    assert idx[0] >= 0 and abs(t[0] - 1.5) < 1e-5 and idx[1] == -1 and idx[2] == -1
    assert (steps == 1).all() and (leaves == 1).all()
    acc = np.zeros((8, 8, 4), np.float32)
    O.render(arrays, 8, 8, [0.2, 1.0, 2.5], [0.0, 0.45, 0.9], 0.5, [0.5, 0.02], 0.0, 4, 0, 1, 3, acc)
    assert not acc[..., :3].any() and (acc[..., 3] == 1).all()  # looking away: only black environment
