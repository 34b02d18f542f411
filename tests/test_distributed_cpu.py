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
    t3 = torch.from_numpy(acc.copy())
    g3 = D.TileGather(rank, world, W, H, torch.device("cpu"), channels=3)  # RGB only, alpha set by rank 0
    g3.exchange(t3)
    g3.exchange(t3)  # a second read-out of the same target
    # the accumulator is cleared (fspt_clear zeroes alpha too), re-rendered and read out again: alpha must come back
    t4 = torch.from_numpy(acc.copy())
    g3.exchange(t4)
    t4.zero_()
    t4 += torch.from_numpy(acc)
    g3.exchange(t4)
    # repeated sum-reduces of the same accumulator (bench.py --exchange reduce --reps N): rank 0 must not add the other
    # ranks' pixels of the previous read-out again
    t5 = torch.from_numpy(acc.copy())
    own = torch.from_numpy(mask)
    for _ in range(3):
        D.reduce_radiance(t5, dst=0, own_mask=own)
    # ... and a non-finite value left on a foreign pixel must not survive the zeroing (0 * inf = NaN would): ADVICE r3
    t6 = torch.from_numpy(acc.copy())
    foreign = ~own
    D.reduce_radiance(t6, dst=0, foreign_mask=foreign)
    fy, fx = np.argwhere(~mask)[0]
    t6[fy, fx, 0] = float("inf"); t6[fy, fx, 1] = float("nan")
    D.reduce_radiance(t6, dst=0, foreign_mask=foreign)
    if rank == 0:
        np.save(os.path.join(tmp, "cleared_rgb.npy"), t4.numpy())
        np.save(os.path.join(tmp, "reduced3.npy"), t5.numpy())
        np.save(os.path.join(tmp, "reduced_nonfinite.npy"), t6.numpy())
        np.save(os.path.join(tmp, "reduced.npy"), t.numpy())
        np.save(os.path.join(tmp, "gathered.npy"), t2.numpy())
        np.save(os.path.join(tmp, "gathered_rgb.npy"), t3.numpy())
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
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "gathered_rgb.npy")), want)  # ... shipping RGB only
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "cleared_rgb.npy")), want)   # ... after a clear
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "reduced3.npy")), want)      # three reduces in a row
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "reduced_nonfinite.npy")), want)  # inf / NaN on a foreign pixel


def test_watchdog_ends_a_hung_rank():
    """A rank stuck in a collective whose peer never arrives must END, non-zero, and say where (the
    first multi-GPU run on hardware must not be able to hang silently).  One rank of a 2-rank gloo group whose peer
    never starts: init_process_group's own timeout or the watchdog behind it fires."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from fspt_amd import distributed as D\n"
            "with D.Watchdog(1.5, 'a collective whose peer never arrives', 0):\n"
            "    import time; time.sleep(60)\n" % root)
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and time.time() - t0 < 45
    assert "TIMEOUT after 2 s in: a collective whose peer never arrives" in p.stderr
    # and a block that finishes in time is left alone
    with D.Watchdog(5.0, "nothing", 0):
        pass


def test_tile_ownership_partitions_frame():
    for world in (1, 2, 3, 8):
        total = np.zeros((H, W), np.int32)
        for r in range(world):
            total += D.owner_mask(r, world, W, H)
        assert (total == 1).all()
        ids = sorted(t for r in range(world) for t in D.owned_tiles(r, world, W, H))
        assert ids == list(range(((W + 31) // 32) * ((H + 31) // 32)))


def test_weak_frame_keeps_per_gpu_pixels():
    """Weak scaling keeps the picture (aspect ratio) and the per-GPU pixel count: sqrt(N) times the linear resolution."""
    for n in (1, 2, 4, 8):
        w, h = D.weak_frame(n, 1920, 1080)
        assert abs(w * h - n * 1920 * 1080) <= 0.003 * n * 1920 * 1080, (n, w, h)
        assert abs(w / h - 1920 / 1080) < 2e-3
    assert D.weak_frame(1, 1920, 1080) == (1920, 1080) and D.weak_frame(4, 1920, 1080) == (3840, 2160)


def _run_bench(argv, env_extra=None, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, [json.loads(l) for l in lines]


@pytest.mark.parametrize("exchange", ["gather", "reduce"])
def test_bench_self_launches_n_ranks(exchange):
    """`python bench.py --gpus 2` (no torchrun) used to run ONE rank and print n_gpus 1.  It now starts
    its own 2 rank processes; --dry-run exercises exactly that launcher, the rendezvous and the read-out exchange on
    CPU over gloo (no rendering).  One JSON line, n_gpus == the ranks the process group saw."""
    p, out = _run_bench(["--gpus", "2", "--dry-run", "--exchange", exchange, "--config", "c4"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(out) == 1
    j = out[0]
    assert j["n_gpus"] == 2 and j["world_size_seen"] == 2 and j["self_launched"] and j["exchange_ok"]
    assert j["scaling"] == "strong" and j["frame"] == [3840 // 16, 2160 // 16]  # C4: the frame does not grow with N


def test_bench_under_torchrun_and_rank_mismatch():
    """The driver's form (python -m torch.distributed.run ... bench.py --gpus N): bench.py is one rank and does not
    launch anything; a --gpus that disagrees with the world size fails loudly instead of running fewer GPUs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    import json
    out = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(out) == 1 and out[0]["n_gpus"] == 2 and out[0]["world_size_seen"] == 2 and not out[0]["self_launched"]
    assert out[0]["scaling"] == "weak" and out[0]["frame"] == [D.weak_frame(2, 1920, 1080)[0] // 16, D.weak_frame(2, 1920, 1080)[1] // 16]
    # WORLD_SIZE says 2 ranks, --gpus says 4
    p, out = _run_bench(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in p.stderr and not out
