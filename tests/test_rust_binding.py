"""CPU: the Rust side of the C ABI (bindings/rust/loupiote_hip; north star: "Rust host code drives a thin C-ABI layer").  The image has
no cargo / rustc, so the crate cannot be compiled here; what is checked instead: src/ffi.rs is exactly what tools/gen_rust_ffi.py
generates from the current include/lpt.h, it declares every exported entry point, every plain-data struct and constant, and the
hand-written safe layer (src/lib.rs: Device / Scene / SceneGPU / ProbeGPU / Renderer / BlitMode / Error / loaders, the names of the
reference's crates/lib) only calls functions and constants that the generated module declares."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "bindings", "rust", "loupiote_hip")


def _header_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "lpt.h")).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(lpt_[a-z0-9_]+)\s*\(", text))), text


def test_generated_ffi_is_current_and_complete():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    ffi = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    syms, header = _header_symbols()
    declared = set(re.findall(r"pub fn (lpt_[a-z0-9_]+)\(", ffi))
    assert declared == set(syms) and len(declared) >= 100
    for struct in re.findall(r"typedef struct (\w+) \{", header):
        assert "pub struct %s {" % struct in ffi, struct
    for const in re.findall(r"\b(LPT_[A-Z0-9_]+)\s*=\s*\d+", header) + ["LPT_ABI_VERSION", "LPT_INVALID_INDEX", "LPT_UPLOAD_NO_TEXTURE_PAIRS"]:
        assert re.search(r"pub const %s: " % const, ffi), const
    # a few translations by eye: const-ness of pointers, arrays that decay, pointer to const pointer
    assert "pub fn lpt_renderer_raytrace(r: *mut lpt_renderer, view_transform: *const f32) -> c_int;" in ffi
    assert "peers: *const *mut lpt_renderer" in ffi and "pub fn lpt_last_error() -> *const c_char;" in ffi
    assert "pub label: [c_char; 32]," in ffi and "pub color: [f32; 4]," in ffi


def test_safe_layer_only_uses_what_the_ffi_declares():
    ffi = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    lib = open(os.path.join(CRATE, "src", "lib.rs")).read()
    declared = set(re.findall(r"pub fn (lpt_[a-z0-9_]+)\(", ffi)) | set(re.findall(r"pub const (LPT_[A-Z0-9_]+):", ffi)) | set(re.findall(r"pub struct (lpt_\w+)", ffi))
    used = set(re.findall(r"\bffi::(lpt_\w+|LPT_[A-Z0-9_]*[A-Z0-9])\b", lib))
    assert used and used <= declared, sorted(used - declared)
    for name in ("pub struct Device", "pub struct Scene", "pub struct SceneGPU", "pub struct ProbeGPU", "pub struct Renderer", "pub enum BlitMode", "pub enum Error", "pub fn load_gltf"):
        assert name in lib, name      # crates/lib/src/lib.rs:1-11
    # balanced delimiters: the cheapest syntax check available without a compiler
    for path in ("src/lib.rs", "src/ffi.rs", "build.rs"):
        src = re.sub(r"//[^\n]*", "", open(os.path.join(CRATE, path)).read())
        src = re.sub(r'"(?:[^"\\]|\\.)*"', '""', src)
        for a, b in ("()", "[]", "{}"):
            assert src.count(a) == src.count(b), (path, a)
    assert os.path.exists(os.path.join(CRATE, "Cargo.toml"))
