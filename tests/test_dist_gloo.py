"""CPU, world_size 2 over gloo: the N>1 path of bench.py — tile ownership (loupiote_amd.dist), one
radiance reduce per frame to rank 0, MAX-over-ranks timing reduction — with the oracle standing in
for the GPU renderer on each rank (there is no GPU here; the device-side ownership rule is checked
against the same owner_map in the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from loupiote_amd import dist as D, testing as T
    glb = open(os.path.join(ROOT, "tests", "golden", "cornell-box.glb"), "rb").read()
    W, H, bounces, frames = 96, 40, 3, 2
    # each rank renders only its tiles (the oracle applies the same tile_id mod N rule)
    from oracle import gltf_oracle as G, orc
    s = G.Scene()
    G.load_gltf(glb, s)
    s.lights[0] = T.cornell_light()[0]
    sc = orc.OracleScene.from_scene(s, probe=T.CORNELL_PROBE)
    acc, cnt = sc.render(W, H, T.look(T.CORNELL_EYE, T.CORNELL_DIR), T.VFOV, bounces, frames=frames, rank=rank, world_size=world,
                         tile_w=D.TILE_W, tile_h=D.TILE_H, threads=2, want_counters=True)
    mask = D.owned_mask(W, H, rank, world)
    assert np.array_equal(acc[..., 3] > 0, mask)
    buf = torch.from_numpy(acc.copy())
    buf2 = torch.from_numpy(acc.copy())
    D.reduce_radiance(buf, dst=0)                       # the one data-path collective
    D.OwnedTileGather(W, H, rank, world)(buf2)          # its compact form: owned pixels only
    if rank == 0:
        assert buf2.numpy().tobytes() == buf.numpy().tobytes()
    rays = torch.tensor([cnt.closest + cnt.shadow], dtype=torch.float64)
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)         # whole-job ray count
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # MAX over ranks, as bench.py does with elapsed time
    assert float(t.item()) == 0.5 + (world - 1)
    if rank == 0:
        full, fc = sc.render(W, H, T.look(T.CORNELL_EYE, T.CORNELL_DIR), T.VFOV, bounces, frames=frames, threads=2, want_counters=True)
        ok = buf.numpy().tobytes() == full.tobytes() and int(rays.item()) == fc.closest + fc.shadow
        open(out_path, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharded_reduce_equals_single_rank(tmp_path):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def _worker_exchange(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from loupiote_amd import dist as D
    ok = True
    for (W, H) in [(100, 37), (64, 16), (33, 9)]:       # ragged tiles, uneven tile counts per rank
        rng = np.random.default_rng(100 * W + rank)
        a = rng.random((H, W, 4), dtype=np.float32)
        a[~D.owned_mask(W, H, rank, world)] = 0.0
        r1, r2 = torch.from_numpy(a.copy()), torch.from_numpy(a.copy())
        D.reduce_radiance(r1, dst=0)
        D.OwnedTileGather(W, H, rank, world)(r2)
        if rank == 0:
            ok = ok and r1.numpy().tobytes() == r2.numpy().tobytes() and bool((r1[..., 3] > 0).all())
    if rank == 0:
        open(out_path, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_owned_tile_gather_equals_reduce_three_ranks(tmp_path):
    out = tmp_path / "result.txt"
    mp.spawn(_worker_exchange, args=(3, _free_port(), str(out)), nprocs=3, join=True)
    assert out.read_text() == "ok"


def test_owner_map_properties():
    sys.path.insert(0, ROOT)
    from loupiote_amd import dist as D
    for (w, h, n) in [(1920, 1080, 8), (200, 120, 3), (33, 9, 2), (31, 7, 4)]:
        om = D.owner_map(w, h, n)
        assert om.min() >= 0 and om.max() < n
        total = 0
        for r in range(n):
            m = D.owned_mask(w, h, r, n)
            total += int(m.sum())
            assert D.owned_slots(w, h, r, n) >= int(m.sum())
        assert total == w * h
        # interleaved: neighbouring tiles go to different ranks when n > 1
        if w > D.TILE_W and n > 1:
            assert om[0, 0] != om[0, D.TILE_W]
    # 1080p splits evenly over 8 GPUs (8100 tiles)
    om = D.owner_map(1920, 1080, 8)
    counts = np.bincount(om.reshape(-1), minlength=8)
    assert counts.max() - counts.min() <= D.TILE_W * D.TILE_H


# ---------------------------------------------------------------------------------------------------------------
# The native exchange (lpt_renderer_exchange, LPT_EXCHANGE_GATHER_TILES) on the CPU: its staging layout comes from the
# library (lpt_shard_layout: pure host arithmetic, callable without a GPU); the slot <-> pixel rule is restated here
# from kernels.h (slot_to_pixel / k_pack_owned / k_unpack_frame), the send / recv pairs become one gloo gather.
def _slot_pixels(w, h, rank, world, tw, th):
    """pixel index of every slot of `rank`, -1 outside the image: tile after owned tile; inside a tile in 8x8 blocks (block after
    block along the tile's rows, row-major inside a block) when both tile sides are multiples of 8, else row-major (kernels.h
    within_to_xy)"""
    tiles_x, tiles_y = (w + tw - 1) // tw, (h + th - 1) // th
    out = []
    for tile in range(rank, tiles_x * tiles_y, world):
        ty, tx = divmod(tile, tiles_x)
        ys, xs = np.mgrid[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
        px = np.where((xs < w) & (ys < h), ys * w + xs, -1)
        if tw % 8 == 0 and th % 8 == 0:
            px = px.reshape(th // 8, 8, tw // 8, 8).transpose(0, 2, 1, 3)   # (block row, block column, y in block, x in block)
        out.append(px.reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def test_library_shard_layout_matches_the_ownership_rule():
    sys.path.insert(0, ROOT)
    from loupiote_amd import dist as D
    for (w, h, n, tw, th) in [(1920, 1080, 8, 32, 8), (200, 120, 3, 32, 8), (33, 9, 2, 32, 8), (31, 7, 4, 8, 8), (97, 61, 5, 16, 8), (64, 16, 7, 32, 8)]:
        off_expected = 0
        for r in range(n):
            slots, off = D.shard_layout(w, h, r, n, tw, th)
            px = _slot_pixels(w, h, r, n, tw, th)
            assert slots == px.size == D.owned_slots(w, h, r, n, tw, th) and off == off_expected
            mask = np.zeros(w * h, bool)
            mask[px[px >= 0]] = True
            assert np.array_equal(mask.reshape(h, w), D.owned_mask(w, h, r, n, tw, th))
            off_expected += slots
    with pytest.raises(ValueError):
        D.shard_layout(64, 64, 3, 3)


def _worker_native_layout(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from loupiote_amd import dist as D
    ok = True
    for (W, H, tw, th) in [(100, 37, 32, 8), (203, 117, 8, 8), (33, 9, 16, 8)]:
        rng = np.random.default_rng(7 * W + rank)
        accum = rng.random((H * W, 4), dtype=np.float32)
        accum[~D.owned_mask(W, H, rank, world, tw, th).reshape(-1)] = 0.0          # owned-pixels-only accumulation buffer
        px = _slot_pixels(W, H, rank, world, tw, th)
        packed = np.where((px >= 0)[:, None], accum[np.maximum(px, 0)], 0.0).astype(np.float32)      # k_pack_owned
        layout = [D.shard_layout(W, H, q, world, tw, th) for q in range(world)]
        pad = max(n for n, _ in layout)
        send = torch.zeros((pad, 4))
        send[: packed.shape[0]] = torch.from_numpy(packed)
        recv = [torch.zeros((pad, 4)) for _ in range(world)] if rank == 0 else None
        dist.gather(send, recv, dst=0)                                                # ncclSend / grouped ncclRecv
        ref = torch.from_numpy(accum.copy())
        dist.reduce(ref, dst=0, op=dist.ReduceOp.SUM)                                 # LPT_EXCHANGE_REDUCE
        if rank == 0:
            staged = np.zeros((sum(n for n, _ in layout), 4), np.float32)
            for q, (n, off) in enumerate(layout):
                staged[off:off + n] = recv[q][:n].numpy()
            frame = np.zeros((H * W, 4), np.float32)
            for q in range(world):                                                   # k_unpack_frame
                pq = _slot_pixels(W, H, q, world, tw, th)
                sel = pq >= 0
                frame[pq[sel]] = staged[layout[q][1] + np.flatnonzero(sel)]
            ok = ok and frame.tobytes() == ref.numpy().tobytes() and bool((frame[:, 3] > 0).all())
    if rank == 0:
        open(out_path, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_native_gather_layout_equals_reduce(tmp_path, world):
    out = tmp_path / "result.txt"
    mp.spawn(_worker_native_layout, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"
