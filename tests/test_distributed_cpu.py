"""N>1 path on CPU: world_size-2 gloo.  Each rank traces its tile shard (the oracle stands in
for the GPU kernels here, as the checker-side renderer), the full-size buffers are sum-reduced
to rank 0 exactly as bench.py does over RCCL, and the result must equal the unsharded render."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fspt_amd import distributed as D

W, H, TICKS, SEED = 100, 70, 2, 17


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import oracle as O
    from fspt_amd import scene as S
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D.init_process_group(backend="gloo")
    arrays = S.bunny_scene(n=8, env_size=(64, 32))
    cam = S.BUNNY_CAMERA
    lens = S.lens_features(cam["focal_depth"], cam["aperture"])
    acc = np.zeros((H, W, 4), np.float32)
    O.render(arrays, W, H, cam["P"], cam["I"], cam["fov_scale"], lens, cam["env_theta"], 4, 0, TICKS, SEED, acc,
             shard=rank, n_shards=world, tile=D.TILE)
    # a rank only ever touches its own tiles
    mask = D.owner_mask(rank, world, W, H)
    assert not acc[~mask].any() and (acc[..., 3][mask] == 1).all()
    t = torch.from_numpy(acc.copy())
    D.reduce_radiance(t, dst=0)
    t2 = torch.from_numpy(acc.copy())
    D.TileGather(rank, world, W, H, torch.device("cpu")).exchange(t2)
    if rank == 0:
        np.save(os.path.join(tmp, "reduced.npy"), t.numpy())
        np.save(os.path.join(tmp, "gathered.npy"), t2.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_shard_and_reduce(tmp_path):
    import oracle as O
    from fspt_amd import scene as S
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "reduced.npy"))
    arrays = S.bunny_scene(n=8, env_size=(64, 32))
    cam = S.BUNNY_CAMERA
    lens = S.lens_features(cam["focal_depth"], cam["aperture"])
    want = np.zeros((H, W, 4), np.float32)
    O.render(arrays, W, H, cam["P"], cam["I"], cam["fov_scale"], lens, cam["env_theta"], 4, 0, TICKS, SEED, want)
    assert np.array_equal(got, want)
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "gathered.npy")), want)  # tile-gather exchange


def test_tile_ownership_partitions_frame():
    for world in (1, 2, 3, 8):
        total = np.zeros((H, W), np.int32)
        for r in range(world):
            total += D.owner_mask(r, world, W, H)
        assert (total == 1).all()
        ids = sorted(t for r in range(world) for t in D.owned_tiles(r, world, W, H))
        assert ids == list(range(((W + 31) // 32) * ((H + 31) // 32)))


def test_weak_frame_keeps_per_gpu_pixels():
    for n in (1, 2, 4, 8):
        w, h = D.weak_frame(n, 1920, 1080)
        assert w * h == n * 1920 * 1080
