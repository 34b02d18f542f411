"""Multi-GPU host logic: one process per GPU, the frame cut into
tile x tile pixel tiles dealt round-robin to the ranks, NO collective on the data
path; ONE sum-reduce of the zero-initialised full-size RGBA32F radiance buffer to
rank 0 at read-out (RCCL over xGMI: backend "nccl"; gloo on CPU for tests).

Per-pixel sample accumulation depends only on the pixel (tracer.fs:516-517) and the
RNG only on pixel coordinates + randBase (camera.fs:38, tracer.fs:458), so the
reduced buffer is bit-identical to a single-GPU render.
"""
import os

TILE = 32


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None, device=None):
    """torch.distributed rendezvous from the torchrun environment (MASTER_ADDR etc.)."""
    import torch.distributed as dist
    rank, _, world = env_rank()
    if world <= 1:
        return None
    if backend is None:
        import torch
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def weak_frame(n_gpus, width, height):
    """Frame size that keeps the per-GPU pixel count fixed as n_gpus grows."""
    fac = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}.get(n_gpus, (n_gpus, 1))
    return width * fac[0], height * fac[1]


def owned_tiles(rank, world, width, height, tile=TILE):
    """Tile ids (row-major over the tile grid) traced by `rank`: id % world == rank."""
    tiles_x, tiles_y = (width + tile - 1) // tile, (height + tile - 1) // tile
    return list(range(rank, tiles_x * tiles_y, world))


def owner_mask(rank, world, width, height, tile=TILE):
    """Boolean [H, W] mask of the pixels `rank` traces."""
    import numpy as np
    tiles_x = (width + tile - 1) // tile
    ys, xs = np.mgrid[0:height, 0:width]
    return ((ys // tile) * tiles_x + xs // tile) % world == rank


def reduce_radiance(accum, dst=0):
    """The one exchange step: sum the ranks' full-size buffers onto `dst` in place."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
