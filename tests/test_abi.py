"""CPU: the C-ABI library loads and exports every symbol include/lpt.h declares; plain-data
layouts; status strings; the product fails loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lpt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lpt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    syms = _declared_symbols()
    assert len(syms) >= 60
    L = C.CDLL(A.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), "libloupiote_hip.so does not export %s" % s
        assert s in A.SIGNATURES, "%s has no ctypes signature" % s
    assert sorted(A.SIGNATURES) == syms


def test_every_entry_point_cites_the_reference():
    """each declaration in lpt.h carries a 'replaces:' citation or is a documented build-only extension"""
    text = open(os.path.join(ROOT, "include", "lpt.h")).read()
    assert text.count("replaces:") >= 30
    assert "renderer.rs:392-549" in text  # Renderer::raytrace, the hot path


def test_struct_layouts():
    assert A.MATERIAL_DT.itemsize == 32      # Material: 5 fields, binary.rs:63-69
    assert A.VERTEX_DT.itemsize == 32        # Vertex {position[4], normal[4]}, binary.rs:20-28
    assert A.LIGHT_DT.itemsize == 64
    assert A.INSTANCE_DT.itemsize == 80
    assert A.HIT_DT.itemsize == 16
    assert A.lib().lpt_abi_version() == 6
    assert A.lib().lpt_max_per_pixel_bytes() == 48


def test_status_strings_match_reference_error_text():
    L = A.lib()
    # errors.rs:8-20
    assert L.lpt_status_string(A.LPT_ERR_FILE_NOT_FOUND).decode().startswith("file not found")
    assert L.lpt_status_string(A.LPT_ERR_READBACK).decode() == "failed to read pixels from GPU to CPU"
    assert L.lpt_status_string(A.LPT_ERR_ACCEL_BUILD).decode().startswith("failed to build acceleration structure")
    assert L.lpt_status_string(0).decode() == "ok"


def test_scene_default_has_one_dummy_per_array():
    """Scene::default() (scene.rs:37-54)"""
    s = lp.Scene()
    c = s.counts()
    assert (c.materials, c.entries, c.vertices, c.instances, c.lights, c.images, c.indices) == (1, 1, 1, 1, 1, 0, 0)
    l = s.lights[0]
    assert tuple(l["normal"]) == (0, 0, 1, 0) and l["tangent"][3] == 0.5 and l["origin"][3] == 1.0
    assert np.array_equal(s.instances[0]["model_to_world"], np.eye(4, dtype=np.float32).reshape(16))


def test_add_mesh_rejects_bad_indices_with_accel_build():
    s = lp.Scene()
    pos = np.zeros((3, 3), np.float32)
    with pytest.raises(lp.Error) as e:
        s.add_mesh(pos, indices=[0, 1, 5])
    assert e.value.kind == "AccelBuild"
    with pytest.raises(lp.Error) as e:
        s.add_mesh(pos, indices=[0, 1])
    assert e.value.kind == "AccelBuild"
    assert s.counts().entries == 1  # nothing appended


def test_invalid_arguments_are_reported_not_crashed():
    s = lp.Scene()
    with pytest.raises(lp.Error) as e:
        s.set_light(7, lp.default_light())
    assert e.value.kind == "InvalidArg"
    with pytest.raises(lp.Error):
        s.set_instance_transform(99, np.eye(4, dtype=np.float32))


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_device_create_fails_loudly_without_gpu():
    """no CPU fallback: the product path refuses to run without a HIP device"""
    with pytest.raises(lp.Error) as e:
        lp.Device(0)
    assert e.value.kind == "Hip"
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under loupiote_amd/ imports, links or executes it — not even lazily
    (the oracle-side helpers that tests, smoke() and bench's cpu_baseline leg use live in oracle/harness.py)."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "loupiote_amd")):
        for f in files:
            if not f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                continue
            text = open(os.path.join(dirpath, f), errors="replace").read()
            if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M):  # any import, module level or inside a function
                bad.append(os.path.join(dirpath, f))
            if "lpt_oracle" in text or "liblpt_oracle" in text or "oracle/_ref" in text:
                bad.append(os.path.join(dirpath, f))
    assert bad == [], bad
