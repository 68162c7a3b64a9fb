"""CPU: the tile-ownership rule of the multi-GPU path as the library computes it on the host — lpt_shard_layout(_weighted)
and lpt_shard_owner are pure arithmetic, no GPU needed.  Tile t belongs to virtual rank t % V (V = the sum of the weights); the
virtual ranks are dealt to the ranks by weight; a rank's slots are its tiles in ascending order; rank q's slots start behind those
of ranks 0..q-1 in rank 0's staging area."""
import ctypes as C

import numpy as np
import pytest

from loupiote_amd import _abi as A


def _layout(w, h, tw, th, world, weights):
    wp = None if weights is None else np.ascontiguousarray(weights, np.uint32)
    out = []
    for q in range(world):
        n, off = C.c_uint32(), C.c_uint32()
        assert A.lib().lpt_shard_layout_weighted(w, h, tw, th, world, q, A.ptr(wp), C.byref(n), C.byref(off)) == 0
        out.append((n.value, off.value))
    return out


def _owner(world, weights, tile):
    wp = None if weights is None else np.ascontiguousarray(weights, np.uint32)
    r = C.c_uint32()
    assert A.lib().lpt_shard_owner(world, A.ptr(wp), tile, C.byref(r)) == 0
    return r.value


@pytest.mark.parametrize("size,tile,weights", [
    ((1920, 1080), (32, 8), (5, 8, 8, 8, 8, 8, 8, 8)),
    ((3840, 2160), (32, 8), (3, 8, 8, 8)),
    ((203, 117), (8, 8), (1, 3, 2)),
    ((97, 61), (16, 8), (0, 2, 2, 1)),
    ((640, 360), (32, 8), (1, 1, 1, 1, 1)),
    ((64, 64), (32, 8), (8,) * 8),
])
def test_weighted_layout_follows_the_ownership_rule(size, tile, weights):
    (w, h), (tw, th) = size, tile
    world = len(weights)
    n_tiles = -(-w // tw) * -(-h // th)
    owners = [_owner(world, weights, t) for t in range(n_tiles)]
    lay = _layout(w, h, tw, th, world, weights)
    off = 0
    for q in range(world):
        assert lay[q] == (owners.count(q) * tw * th, off)          # slots = owned tiles x tile area, offsets cumulative
        off += lay[q][0]
    V = sum(weights)
    for start in range(0, max(n_tiles - V, 1), max(V // 2, 1)):    # every window of V consecutive tiles holds weight[q] tiles of rank q
        win = owners[start:start + V]
        if len(win) == V:
            assert [win.count(q) for q in range(world)] == list(weights)
    # interleaving: a rank with weight k of V never waits longer than twice its mean spacing V / k for its next tile
    for q in range(world):
        if weights[q]:
            pos = [t for t, o in enumerate(owners[:4 * V]) if o == q]
            assert all(b - a <= 2 * -(-V // weights[q]) for a, b in zip(pos, pos[1:]))


def test_unit_weights_are_tile_id_mod_world():
    for world in (1, 2, 3, 8):
        assert [_owner(world, None, t) for t in range(40)] == [t % world for t in range(40)]
        assert [_owner(world, (1,) * world, t) for t in range(40)] == [t % world for t in range(40)]
        assert _layout(1920, 1080, 32, 8, world, None) == _layout(1920, 1080, 32, 8, world, (1,) * world)
        plain = []
        for q in range(world):
            n, off = C.c_uint32(), C.c_uint32()
            assert A.lib().lpt_shard_layout(1920, 1080, 32, 8, world, q, C.byref(n), C.byref(off)) == 0
            plain.append((n.value, off.value))
        assert plain == _layout(1920, 1080, 32, 8, world, None)


def test_bad_weights_are_rejected():
    n, off = C.c_uint32(), C.c_uint32()
    for weights in ((0, 0), (9, 1), (8,) * 9):
        wp = np.ascontiguousarray(weights, np.uint32)
        assert A.lib().lpt_shard_layout_weighted(64, 64, 32, 8, len(weights), 0, A.ptr(wp), C.byref(n), C.byref(off)) == A.LPT_ERR_INVALID_ARG
