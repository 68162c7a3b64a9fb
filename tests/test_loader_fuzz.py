"""CPU: the host-side loaders (glTF / GLB, PNG, baseline JPEG, Radiance HDR) under AddressSanitizer + UBSan on mutated
files.  They parse untrusted bytes (gltf.cpp:119), so every input must be accepted or rejected without an
out-of-bounds access, an overflow the sanitizers flag, or an allocation sized by an unchecked header field.
The sanitizer build is CPU-only (tests/tools/fuzz_loaders.cpp + the host sources, g++); a longer campaign
(3 x 88 000 inputs) was run during development, this test keeps a short one in the suite."""
import os
import shutil
import struct
import subprocess
import zlib

import numpy as np
import pytest

import loupiote_amd as lp
from test_loader import QUAD, _hdr_bytes, _jpeg, _smooth_rgb, make_gltf, png_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    d = tmp_path_factory.mktemp("fuzz")
    exe = str(d / "fuzz_loaders")
    src = [os.path.join(ROOT, "tests", "tools", "fuzz_loaders.cpp")] + [os.path.join(ROOT, "loupiote_amd", "csrc", f + ".cpp") for f in ("scene", "gltf", "png", "jpeg", "hdr")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include")] + src + ["-o", exe])
    return exe, d


def test_mutated_files_never_trip_the_sanitizers(fuzzer, cornell_glb):
    exe, d = fuzzer
    img = _smooth_rgb()
    rng = np.random.default_rng(3)
    px = rng.integers(0, 255, (24, 40, 4), dtype=np.uint8)
    px[:, 10:30] = px[:, 10:11]
    seeds = {
        "c.glb": cornell_glb,
        "t.glb": make_gltf([[dict(pos=QUAD, idx=[0, 1, 2, 0, 2, 3], material=0)]], [{"mesh": 0, "translation": [1, 2, 3]}],
                           [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}], images=[_jpeg(img, quality=85, subsampling=2)], textures=[0], glb=True),
        "u.gltf": make_gltf([[dict(pos=QUAD, mode=5)]], [{"mesh": 0}], images=[png_bytes(img)], textures=[0]),
        "s.png": png_bytes(img), "g.png": png_bytes(img[..., 0]), "a.png": png_bytes(np.dstack([img, img[..., :1]])),
        "s.jpg": _jpeg(img, quality=90, subsampling=2), "t.jpg": _jpeg(img, quality=70, subsampling=0, restart_marker_blocks=5), "u.jpg": _jpeg(img[..., 0], quality=80),
        "p.jpg": _jpeg(img, quality=85, subsampling=2, progressive=True), "q.jpg": _jpeg(img[..., 0], quality=60, progressive=True, restart_marker_rows=1),
        "r.hdr": _hdr_bytes(px, rle=True), "f.hdr": _hdr_bytes(px, rle=False),
    }
    paths = []
    for name, data in seeds.items():
        p = d / name
        p.write_bytes(data)
        paths.append(str(p))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:allocator_may_return_null=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, "500:7"] + paths, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "FUZZ_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _png_with_header(w, h):
    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")


def test_regressions_found_by_the_fuzzer():
    """(1) a PNG / JPEG / HDR header may claim any size: the decoders bound it by what the file could encode before they
    allocate (a 100-byte PNG asking for 433 GB); (2) a JPEG scan that names a Huffman table no DHT defined read the table
    uninitialised."""
    for bad in (_png_with_header(0x4000_0000, 0x4000_0000), _png_with_header(65535, 65535)):
        with pytest.raises(lp.Error) as e:
            lp.loaders.load_gltf(make_gltf([[{"pos": QUAD}]], [{"mesh": 0}], images=[bad], textures=[0]), lp.Scene())
        assert e.value.kind == "FileNotFound"
    raw = bytearray(_jpeg(_smooth_rgb(), quality=90, subsampling=0))
    i = raw.index(b"\xff\xda")                       # SOS: component table selectors follow
    ncomp = raw[i + 4]
    for k in range(ncomp):
        raw[i + 6 + 2 * k] = 0x33                    # DC / AC table 3: never defined
    with pytest.raises(lp.Error) as e:
        lp.loaders.load_gltf(make_gltf([[{"pos": QUAD}]], [{"mesh": 0}], images=[bytes(raw)], textures=[0]), lp.Scene())
    assert e.value.kind == "FileNotFound"
    huge_hdr = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 16000 +X 16000\n" + b"\0" * 64
    with pytest.raises(lp.Error):
        lp.load_env(huge_hdr)
