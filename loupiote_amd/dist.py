"""One-process-per-GPU plumbing: pixel-tile ownership and the radiance reduce.

The reference is single-GPU; this is new functionality required by the north
star: frames shard by pixel tile across the GPUs of a node and the accumulated radiance buffer
is summed to rank 0 with one collective per frame (RCCL over xGMI when the tensors live on
GPUs, gloo in the CPU tests).  No arithmetic on radiance happens here except that sum.
"""
import numpy as np

TILE_W, TILE_H = 32, 8


def tile_grid(width, height, tile_w=TILE_W, tile_h=TILE_H):
    return (width + tile_w - 1) // tile_w, (height + tile_h - 1) // tile_h


def owner_map(width, height, world_size, tile_w=TILE_W, tile_h=TILE_H):
    """rank that owns each pixel: tile id (row-major over the tile grid) mod world_size —
    the rule lpt_renderer_set_shard implements on the device (kernels.h slot_to_pixel)."""
    tiles_x, _ = tile_grid(width, height, tile_w, tile_h)
    y, x = np.mgrid[0:height, 0:width]
    tile = (y // tile_h) * tiles_x + (x // tile_w)
    return (tile % world_size).astype(np.int32)


def owned_mask(width, height, rank, world_size, tile_w=TILE_W, tile_h=TILE_H):
    return owner_map(width, height, world_size, tile_w, tile_h) == rank


def owned_slots(width, height, rank, world_size, tile_w=TILE_W, tile_h=TILE_H):
    """number of pixel slots a rank allocates (whole tiles, including out-of-image padding)"""
    tiles_x, tiles_y = tile_grid(width, height, tile_w, tile_h)
    n_tiles = tiles_x * tiles_y
    owned = (n_tiles - rank + world_size - 1) // world_size if n_tiles > rank else 0
    return owned * tile_w * tile_h


def shard_layout(width, height, rank, world_size, tile_w=TILE_W, tile_h=TILE_H):
    """(slots, slot offset) of a rank as the library computes them (lpt_shard_layout): the staging layout of
    lpt_renderer_exchange's owned-tile gather on rank 0"""
    import ctypes as C
    from . import _abi as A
    n, off = C.c_uint32(), C.c_uint32()
    if A.lib().lpt_shard_layout(width, height, tile_w, tile_h, world_size, rank, C.byref(n), C.byref(off)) != 0:
        raise ValueError(A.lib().lpt_last_error().decode())
    return n.value, off.value


def reduce_radiance(buf, dst=0):
    """sum the (rgb-sum, sample-count) accumulation buffers of all ranks onto `dst`.
    Ownership is disjoint, so the sum is a gather; a reduce is the contract (north star)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(buf, dst=dst, op=dist.ReduceOp.SUM)
    return buf


class OwnedTileGather:
    """The compact form of the frame exchange: every rank sends only the pixels it owns
    (W*H/N * 16 B instead of the whole W*H*16 B buffer) and `dst` scatters them into the frame.
    Ownership is disjoint and each rank's buffer is zero outside its tiles, so the result equals
    `reduce_radiance` bit for bit.  Index tensors are built once per (size, world); the collective is one
    `torch.distributed.gather` (RCCL send/recv pairs over xGMI on GPUs, gloo in the CPU tests)."""

    def __init__(self, width, height, rank, world_size, device="cpu", dst=0, tile_w=TILE_W, tile_h=TILE_H):
        import torch
        self.rank, self.world, self.dst = rank, world_size, dst
        om = owner_map(width, height, world_size, tile_w, tile_h).reshape(-1)
        self.counts = [int((om == r).sum()) for r in range(world_size)]
        self.pad = max(self.counts) if self.counts else 0
        self.mine = torch.from_numpy(np.flatnonzero(om == rank).astype(np.int64)).to(device)
        self.all = [torch.from_numpy(np.flatnonzero(om == r).astype(np.int64)).to(device) for r in range(world_size)] if rank == dst else None
        self.send = torch.zeros((self.pad, 4), dtype=torch.float32, device=device)
        self.recv = [torch.empty_like(self.send) for _ in range(world_size)] if rank == dst else None

    def __call__(self, buf):
        """buf: [H, W, 4] float32 accumulation buffer of this rank (modified in place on `dst`)"""
        import torch
        import torch.distributed as dist
        flat = buf.view(-1, 4)
        if not (dist.is_initialized() and self.world > 1):
            return buf
        n = self.counts[self.rank]
        torch.index_select(flat, 0, self.mine, out=self.send[:n])
        dist.gather(self.send, self.recv if self.rank == self.dst else None, dst=self.dst)
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.rank:
                    flat.index_copy_(0, self.all[r], self.recv[r][: self.counts[r]])
        return buf


class DevView:
    """a device pointer as a torch-importable buffer (__cuda_array_interface__): plumbing for the exchanges"""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def exchange_denoiser_inputs(renderer, device, dst=0, stream=None):
    """Sharded frame in a denoising BlitMode: sum every rank's filter inputs (noisy radiance, G-buffer, motion —
    zero outside the rank's tiles, so the sum is a gather) onto `dst`, then run the filter passes there.
    One RCCL reduce per buffer (float32 / int32 / float32), issued on `stream` (the renderer's) when given."""
    import torch
    import torch.distributed as dist
    noisy, gbuf, motion, n = renderer.denoiser_inputs()
    bufs = [torch.as_tensor(DevView(noisy, 4 * n, "<f4"), device=device), torch.as_tensor(DevView(gbuf, 4 * n, "<i4"), device=device),
            torch.as_tensor(DevView(motion, 2 * n, "<f4"), device=device)]
    ctx = torch.cuda.stream(stream) if stream is not None else _null()
    with ctx:
        if dist.is_initialized():
            for b in bufs:
                dist.reduce(b, dst=dst, op=dist.ReduceOp.SUM)
    if not dist.is_initialized() or dist.get_rank() == dst:
        renderer.denoise_filter()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
