"""Host logic: the 8-wide BVH builder (loupiote_amd/csrc/bvh.cpp) checked on the CPU by tests/tools/bvh_check.cpp,
a test-only walker of the Node8 format: random rays must find exactly the brute-force closest hit (t and prim),
every triangle must be referenced once, for both collapse modes and for degenerate soups."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "loupiote_amd", "csrc")


@pytest.fixture(scope="module")
def bvh_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bvh") / "bvh_check")
    src = [os.path.join(ROOT, "tests", "tools", "bvh_check.cpp")] + [os.path.join(CSRC, f) for f in
                                                                      ("scene.cpp", "bvh.cpp", "png.cpp", "jpeg.cpp", "gltf.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-DLPT_EXPERIMENTS", "-o", exe] + src + ["-lpthread"], check=True)   # LPT_EXPERIMENTS: the builder A/B knobs (LPT_BVH_*) exist in this test build only
    return exe


def run(exe, tmp_path, tris, rays, mode=None, eye=None, reinsert=None, split=None):
    path = str(tmp_path / "soup.bin")
    tris = np.ascontiguousarray(tris, "<f4").reshape(-1, 9)
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(tris)))
        f.write(tris.tobytes())
    env = dict(os.environ)
    if mode:
        env["LPT_BVH_COLLAPSE"] = mode
    if reinsert is not None:
        env["LPT_BVH_REINSERT"] = reinsert
    if split is not None:
        env["LPT_BVH_SPLIT"] = split
    cmd = [exe, path, str(rays), "1"] + ([str(v) for v in eye] if eye else [])
    p = subprocess.run(cmd, env=env, capture_output=True, text=True)
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0, (out, p.stderr)
    return out


def random_soup(rng, n, extent=10.0, size=0.6):
    c = rng.uniform(-extent, extent, (n, 1, 3))
    return (c + rng.normal(0, size, (n, 3, 3))).astype(np.float32)


@pytest.mark.parametrize("mode", ["dp", "greedy"])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 9, 64, 3000])
def test_random_soups_match_brute_force(bvh_check, tmp_path, mode, n):
    rng = np.random.default_rng(n)
    out = run(bvh_check, tmp_path, random_soup(rng, n), 4000, mode)
    assert out["triangles"] == n and out["mismatches"] == 0 and out["bad_refs"] == 0
    assert out["hits"] > 0 or n < 9


def test_degenerate_soups(bvh_check, tmp_path):
    rng = np.random.default_rng(7)
    one = random_soup(rng, 1)
    same = np.repeat(one, 200, axis=0)                                   # 200 coincident triangles: tie rule -> lowest prim
    out = run(bvh_check, tmp_path, same, 2000)
    assert out["mismatches"] == 0 and out["bad_refs"] == 0 and out["depth"] <= 30
    flat = random_soup(rng, 500)
    flat[..., 1] = 0.25                                                   # coplanar: zero-extent axis, exponent clamp
    out = run(bvh_check, tmp_path, flat, 4000)
    assert out["mismatches"] == 0 and out["bad_refs"] == 0
    sliver = random_soup(rng, 300, extent=1000.0, size=1e-3)             # tiny triangles far apart
    sliver[::2] *= 1e-3
    out = run(bvh_check, tmp_path, sliver, 4000)
    assert out["mismatches"] == 0 and out["bad_refs"] == 0
    out = run(bvh_check, tmp_path, np.zeros((0, 9), np.float32), 100)     # empty scene: one empty node, nothing hit
    assert out["triangles"] == 0 and out["hits"] == 0 and out["nodes"] == 1


def test_grid_mesh_quality(bvh_check, tmp_path):
    """a tessellated wall: the SAH-optimal collapse fills nodes (fewer, fuller nodes than the greedy one) and
    does not visit more nodes per ray"""
    n = 96
    u, v = np.meshgrid(np.linspace(-5, 5, n + 1), np.linspace(-5, 5, n + 1), indexing="xy")
    p = np.stack([u, v, 0.3 * np.sin(u) * np.cos(v)], -1).astype(np.float32)
    a, b, c, d = p[:-1, :-1], p[:-1, 1:], p[1:, :-1], p[1:, 1:]
    tris = np.concatenate([np.stack([a, b, d], -2), np.stack([a, d, c], -2)]).reshape(-1, 9)
    dp = run(bvh_check, tmp_path, tris, 20000, "dp", eye=(0, 0, 8))
    gr = run(bvh_check, tmp_path, tris, 20000, "greedy", eye=(0, 0, 8))
    assert dp["mismatches"] == gr["mismatches"] == 0 and dp["hits"] == gr["hits"]
    assert dp["nodes"] < 0.8 * gr["nodes"] and dp["nodes_per_ray"] <= 1.02 * gr["nodes_per_ray"]


@pytest.mark.parametrize("reinsert", ["0", "1,0.05", "12,1.0"])
def test_reinsertion_optimiser_keeps_every_hit(bvh_check, tmp_path, reinsert):
    """the insertion-based optimisation of the binary tree (bvh.cpp Reinserter) only moves subtrees: whatever its settings
    (off, light, every node twelve times), random rays find exactly the brute-force hits and every triangle sits in one leaf"""
    rng = np.random.default_rng(11)
    soup = np.concatenate([random_soup(rng, 2500), random_soup(rng, 1500, extent=2.0, size=0.05), random_soup(rng, 40, extent=8.0, size=6.0)])
    out = run(bvh_check, tmp_path, soup, 6000, reinsert=reinsert)
    assert out["triangles"] == len(soup) and out["mismatches"] == 0 and out["bad_refs"] == 0 and out["hits"] > 0
    assert out["depth"] <= 30


def test_reinsertion_optimiser_does_not_cost_visits(bvh_check, tmp_path):
    """clustered geometry of mixed scale (long triangles across clusters of small ones): the optimised tree is not worse"""
    rng = np.random.default_rng(5)
    soup = np.concatenate([random_soup(rng, 6000, extent=10.0, size=0.15), random_soup(rng, 200, extent=6.0, size=5.0)])
    off = run(bvh_check, tmp_path, soup, 20000, reinsert="0")
    on = run(bvh_check, tmp_path, soup, 20000)
    assert off["mismatches"] == on["mismatches"] == 0 and off["hits"] == on["hits"]
    assert on["nodes_per_ray"] <= 1.01 * off["nodes_per_ray"]


def test_mixed_scale_hall_matches_brute_force(bvh_check, tmp_path):
    """scenes.synthetic_hall: walls of two triangles each around 12 000 centimetre-sized ones, slivers through the whole volume, 120 coincident triangles and a
    telescope of nested ones — the closest hit (t and lowest primitive id among equals) is the brute-force one from inside the hall, for both collapse modes"""
    import sys
    sys.path.insert(0, ROOT)
    from loupiote_amd import scenes
    d = scenes.synthetic_hall()
    tris = np.concatenate([m["positions"][m["indices"]].reshape(-1, 9) for m in d["meshes"]])
    assert len(tris) == d["triangles"] == 12382
    for mode in ("dp", "greedy"):
        out = run(bvh_check, tmp_path, tris, 20000, mode, eye=d["camera"]["origin"])
        assert out["triangles"] == 12382 and out["mismatches"] == 0 and out["bad_refs"] == 0 and out["hits"] > 15000, out


def test_grazing_rays_at_a_scale_ratio_of_a_million_lose_no_hit(bvh_check, tmp_path):
    """the margin of SPEC §7 in numbers (scenes.origin_dust): 3 000 millimetre-sized triangles around the world origin in a scene 2 000 units across (two far triangles
    set the extent) — a quarter of the walker's rays is aimed at vertices and edge points of random triangles from up to a thousand units away, where the Woop test's rounding
    (which grows with the ray's coordinates) is largest against the triangle padding (which grows with the triangle's): every closest hit is the brute-force one.
    (At a ratio of 10^12 — a chain of triangles at x = 2^(0.08 k) — the same walker loses 5 of 5 000 such hits: SPEC §7 states the domain.)"""
    import sys
    sys.path.insert(0, ROOT)
    from loupiote_amd import scenes
    d = scenes.origin_dust()
    tris = np.concatenate([m["positions"][m["indices"]].reshape(-1, 9) for m in d["meshes"]])
    out = run(bvh_check, tmp_path, tris, 150000)
    assert out["triangles"] == 3002 and out["mismatches"] == 0 and out["bad_refs"] == 0 and out["hits"] > 10000, out


def test_presplit_slivers_are_split_walls_are_not_and_no_hit_is_lost(bvh_check, tmp_path):
    """round 6 (bvh.cpp presplit): triangles whose boxes are mostly EMPTY — long slivers across a cloud of small triangles — get several references (clipped boxes),
    the rays test fewer triangles, and every random ray still finds exactly the brute-force closest hit (t and prim: the clipped boxes are conservative; a split
    triangle is simply tested in more than one leaf); two-triangle axis-aligned walls, which fill their flat boxes, are left alone; so is a soup of evenly sized triangles"""
    rng = np.random.default_rng(11)
    small = random_soup(rng, 4000, extent=8.0, size=0.05)
    a0 = rng.uniform(-8, 8, (60, 3)).astype(np.float32)
    dirn = rng.normal(size=(60, 3)).astype(np.float32)
    dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    side = np.cross(dirn, rng.normal(size=(60, 3))).astype(np.float32)
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    slivers = np.stack([a0, a0 + dirn * 12.0, a0 + dirn * 6.0 + side * 0.01], axis=1).astype(np.float32)
    soup = np.concatenate([small, slivers])
    off = run(bvh_check, tmp_path, soup, 6000, split="1e30,0")
    on = run(bvh_check, tmp_path, soup, 6000)
    assert off["references"] == off["triangles"] == len(soup) and off["mismatches"] == 0
    assert on["references"] > on["triangles"] == len(soup) and on["references"] <= 1.3 * len(soup) + 1
    assert on["mismatches"] == 0 and on["bad_refs"] == 0
    assert on["tris_per_ray"] < 0.8 * off["tris_per_ray"], (on, off)
    # walls: two axis-aligned triangles per face around the same cloud — they fill their boxes, nothing to split
    def quad(o, eu, ev):
        o, eu, ev = (np.asarray(x, np.float32) for x in (o, eu, ev))
        return [[o, o + eu, o + eu + ev], [o, o + eu + ev, o + ev]]
    walls = np.asarray(quad((-9, -9, -9), (18, 0, 0), (0, 0, 18)) + quad((-9, -9, -9), (18, 0, 0), (0, 18, 0)) + quad((-9, -9, -9), (0, 18, 0), (0, 0, 18)), np.float32)
    out = run(bvh_check, tmp_path, np.concatenate([small, walls]), 4000)
    assert out["references"] == out["triangles"] and out["mismatches"] == 0
    out = run(bvh_check, tmp_path, random_soup(rng, 3000), 3000)
    assert out["references"] == out["triangles"] and out["mismatches"] == 0
