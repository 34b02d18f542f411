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


class Watchdog:
    """Hard deadline around a step that can hang for ever - the rendezvous, a collective whose peer has died.  When it
    fires the rank says on stderr what it was doing and the PROCESS exits with code 3 (os._exit: the main thread may be
    stuck inside RCCL, where no exception can reach it).  The launcher (torchrun / bench.py's self-launch) then sees a
    failed rank and takes the others down - a multi-GPU run cannot fail silently or hang until an outer timeout."""

    def __init__(self, seconds, what, rank=0):
        self.seconds, self.what, self.rank, self._t = float(seconds), what, rank, None

    def _fire(self):
        import sys
        sys.stderr.write(f"[rank {self.rank}] TIMEOUT after {self.seconds:.0f} s in: {self.what}\n")
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        import threading
        if self.seconds > 0:
            self._t = threading.Timer(self.seconds, self._fire)
            self._t.daemon = True
            self._t.start()
        return self

    def __exit__(self, *exc):
        if self._t is not None:
            self._t.cancel()
        return False


def init_process_group(backend=None, device=None, timeout_s=None):
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
    if timeout_s:
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
    with Watchdog(min(timeout_s, 300.0) + 30 if timeout_s else 0, f"init_process_group({backend}, world {world})", rank):
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def weak_frame(n_gpus, width, height):
    """Frame that keeps the per-GPU WORK fixed as n_gpus grows: the same picture (same aspect ratio, so the same share of
    sky, floor and mesh pixels) at sqrt(n_gpus) times the linear resolution; width a multiple of 8, pixel count within
    0.3 % of n_gpus * width * height (exact for 1, 4, 9 ...).  (Growing one side only - 3840x1080 for two GPUs - would
    widen the field of view: more sky, 1.6x cheaper pixels, and a 'scaling efficiency' above 1.)"""
    import math
    if n_gpus <= 1:
        return width, height
    s = math.sqrt(n_gpus)
    w = int(round(width * s / 8.0)) * 8
    h = int(round(w * height / width))
    return w, h


def owned_tiles(rank, world, width, height, tile=TILE):
    """Tile ids (row-major over the tile grid) traced by `rank`: id % world == rank."""
    tiles_x, tiles_y = (width + tile - 1) // tile, (height + tile - 1) // tile
    return list(range(rank, tiles_x * tiles_y, world))


def owner_mask(rank, world, width, height, tile=TILE):
    """Boolean [H, W] mask of the pixels `rank` traces (tile-level ownership, expanded to pixels)."""
    import numpy as np
    tiles_x, tiles_y = (width + tile - 1) // tile, (height + tile - 1) // tile
    g = (np.arange(tiles_y)[:, None] * tiles_x + np.arange(tiles_x)[None, :]) % world == rank
    return np.repeat(np.repeat(g, tile, axis=0), tile, axis=1)[:height, :width]


def owned_pixel_index(rank, world, width, height, tile=TILE, device=None):
    """Linear pixel indices (row-major, int64 tensor) of the pixels `rank` traces, in a fixed order."""
    import numpy as np
    import torch
    idx = np.flatnonzero(owner_mask(rank, world, width, height, tile).reshape(-1))
    return torch.from_numpy(idx).to(device) if device is not None else torch.from_numpy(idx)


class TileGather:
    """Read-out exchange that moves only what each rank owns (the cheaper alternative to reducing
    the whole frame): every rank packs its own pixels (1/world of the frame), ONE gather to `dst` into slices of one
    receive buffer, and `dst` scatters all pieces into its full-size buffer with ONE index_copy_ (rows a rank pads its
    send buffer with repeat its last pixel, so their duplicate writes carry identical values).  Index tensors are built
    once, outside any timed region.

    channels = 3 ships RGB only: tracer.fs:517 writes vec4(rgb, 1), so the alpha of every traced pixel is the constant
    1 - a quarter of the bytes on the links carried no information.  `dst` then sets the alpha of the gathered pixels
    to 1 itself, on every unpack (the bound accumulator may have been cleared in between: fspt_clear zeroes alpha too).
    Only valid when every owned pixel is traced (full-frame viewport); channels = 4 (default) ships the buffer as it is."""

    def __init__(self, rank, world, width, height, device, tile=TILE, dst=0, channels=4, tracer=None):
        import torch
        assert channels in (3, 4)
        self.rank, self.world, self.dst, self.C = rank, world, dst, channels
        self.tracer = tracer
        self.last_stage_ms = None
        if tracer is not None:
            # The library's own pack / unpack kernels (include/fspt_multi.h: fspt_target_pack_tiles / _unpack_tiles - the
            # k_tile_pack the single-process fspt_multi_* host uses): a shard's pixels in work-index order, one launch per
            # piece, no index tensors.  `tracer` must be sharded (rank, world, tile) and bound to the accumulator that is
            # exchanged.  (Without a tracer - CPU tensors, the gloo tests - the torch indexing path below does the same.)
            slots = [tracer.shard_slots(r, world) for r in range(world)]
            self.n_max = max(slots)
            self.send = torch.zeros((max(1, self.n_max), channels), dtype=torch.float32, device=device)
            self.recv = None
            if rank == dst:
                self.big = torch.zeros((world * max(1, self.n_max), channels), dtype=torch.float32, device=device)
                self.recv = list(self.big.split(max(1, self.n_max)))
                self.slots = slots
            # the zero fills above ran on torch's current stream, the pack / unpack kernels run on the library's own: order
            # them once, here (the holes of ragged tiles rely on the fill having happened before the first pack)
            if self.send.is_cuda:
                torch.cuda.current_stream(self.send.device).synchronize()
            return
        self.idx = owned_pixel_index(rank, world, width, height, tile, device)
        counts = [int(owner_mask(r, world, width, height, tile).sum()) for r in range(world)]
        self.n_max = max(counts)
        self.send = torch.zeros((self.n_max, channels), dtype=torch.float32, device=device)
        self.tmp = torch.zeros((self.n_max, 4), dtype=torch.float32, device=device) if channels != 4 else None
        self.recv, self.all_idx = None, None
        if rank == dst:
            self.big = torch.zeros((world * self.n_max, channels), dtype=torch.float32, device=device)
            self.recv = list(self.big.split(self.n_max))  # views: the gather lands in one buffer
            pieces = []
            for r in range(world):
                i = owned_pixel_index(r, world, width, height, tile, device)
                if i.numel() == 0:  # a rank without tiles (tiny frames): its rows are never written by index_copy_
                    i = self.idx[:1] if self.idx.numel() else torch.zeros(1, dtype=torch.int64, device=device)
                    pieces.append(i.expand(self.n_max).clone() if r == dst else None)
                    continue
                pad = i[-1:].expand(self.n_max - i.numel())
                pieces.append(torch.cat([i, pad]))
            self.rows = [r for r in range(world) if pieces[r] is not None and r != dst]
            self.all_idx = torch.cat([pieces[r] for r in self.rows]) if self.rows else None
            self.sel = (torch.cat([torch.arange(r * self.n_max, (r + 1) * self.n_max, device=device) for r in self.rows])
                        if self.rows else None)
            self.contiguous_rows = self.rows == list(range(self.rows[0], self.rows[0] + len(self.rows))) if self.rows else True

    def pack(self, accum):
        """Own pixels of accum ([H, W, 4]) -> self.send ([n_max, channels])."""
        import torch
        if self.tracer is not None:
            bound = getattr(self.tracer, "bound_ptr", None)
            if bound is not None and accum is not None and accum.data_ptr() != bound:
                raise ValueError("TileGather.pack: `accum` is not the accumulator the tracer is bound to "
                                 "(the library's pack kernel reads the bound one)")
            self.tracer.pack_tiles(self.send.data_ptr(), self.C)  # blocking
            return self.send
        flat = accum.view(-1, 4)
        n = self.idx.numel()
        if n:
            if self.C == 4:
                torch.index_select(flat, 0, self.idx, out=self.send[:n])
            else:
                torch.index_select(flat, 0, self.idx, out=self.tmp[:n])
                self.send[:n].copy_(self.tmp[:n, :self.C])
            if n < self.n_max:
                self.send[n:] = self.send[n - 1]
        return self.send

    def unpack(self, accum):
        """`dst` only: the other ranks' rows of self.big -> their pixels of accum."""
        if self.tracer is not None:
            import torch
            if accum.is_cuda:
                torch.cuda.current_stream(accum.device).synchronize()  # the gather ran on torch's stream, the kernels run on the library's
            for r in range(self.world):
                if r != self.dst and self.slots[r]:
                    self.tracer.unpack_tiles(self.recv[r].data_ptr(), r, self.world, self.C)
            return accum
        if self.all_idx is None:
            return accum
        flat = accum.view(-1, 4)
        if self.contiguous_rows:
            src = self.big[self.rows[0] * self.n_max:(self.rows[-1] + 1) * self.n_max]
        else:
            src = self.big.index_select(0, self.sel)
        if self.C == 4:
            flat.index_copy_(0, self.all_idx, src)
        else:
            flat[:, :self.C].index_copy_(0, self.all_idx, src)
            flat[:, 3].index_fill_(0, self.all_idx, 1.0)
        return accum

    def exchange(self, accum, stage_host=False):
        """accum: float32 [H, W, 4] on every rank (only own pixels non-zero); complete on `dst` afterwards.
        stage_host: the packed pieces travel through host memory (a backend without device-tensor gather: gloo with the
        ranks sharing one GPU, bench.py --share-gpu); packing and unpacking stay on the device."""
        import time
        import torch
        import torch.distributed as dist

        def done():  # the stage's work is finished on the device (host clock: the stages below are blocking by nature)
            if accum.is_cuda:
                torch.cuda.synchronize(accum.device)
            return time.perf_counter()
        t0 = time.perf_counter()
        self.pack(accum)
        t1 = done()
        t2 = t1
        if self.world > 1:
            if stage_host:
                send = self.send.cpu()
                recv = [torch.empty_like(send) for _ in range(self.world)] if self.rank == self.dst else None
                dist.gather(send, recv, dst=self.dst)
                if self.rank == self.dst:
                    self.big.copy_(torch.cat(recv))
            else:
                dist.gather(self.send, self.recv if self.rank == self.dst else None, dst=self.dst)
            t2 = done()
            if self.rank == self.dst:
                self.unpack(accum)
        # where this rank's share of the exchange went (bench.py prints it per rank: the first run on several physical
        # GPUs shows pack / collective / scatter cost and imbalance in one pass)
        self.last_stage_ms = {"pack": (t1 - t0) * 1e3, "collective": (t2 - t1) * 1e3, "unpack": (done() - t2) * 1e3}
        return accum


def reduce_radiance(accum, dst=0, own_mask=None, foreign_mask=None, stage_host=False):
    """The one exchange step as a sum-reduce (the RCCL reduce north_star names): the ranks' full-size buffers are summed
    onto `dst` in place.  Every rank's buffer must be zero outside its own tiles.  That holds for a fresh buffer, but
    after a reduce `dst` holds the other ranks' pixels too - a second reduce would add them again - so a caller that
    reduces repeatedly passes `foreign_mask` (bool [H, W], True where the pixel is NOT this rank's; computed once,
    outside the timed exchange) or `own_mask` (its complement): those pixels are set to zero first - masked_fill_, so
    that a non-finite value left there cannot survive as 0 * inf = NaN, and without a full-frame temporary."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if foreign_mask is None and own_mask is not None:
            foreign_mask = ~own_mask
        if foreign_mask is not None:
            accum.masked_fill_(foreign_mask.unsqueeze(-1), 0.0)
        if stage_host:  # (a backend without device-tensor reduce: see TileGather.exchange)
            h = accum.cpu()
            dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM)
            if dist.get_rank() == dst:
                accum.copy_(h)
        else:
            dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
