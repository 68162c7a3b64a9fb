"""ctypes binding of the CPU ORACLE (oracle/lpt_oracle.c).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import gltf_oracle as G

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("prim", C.c_uint32)]


HIT_DT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4")])


class Image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba8", C.c_void_p)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("view", C.c_float * 16), ("vfov", C.c_float),
                ("max_bounces", C.c_uint32), ("user_seed", C.c_uint32), ("seed_counter", C.c_uint32),
                ("frames", C.c_uint32), ("rank", C.c_uint32), ("world_size", C.c_uint32),
                ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("threads", C.c_uint32),
                ("brute_force", C.c_uint32), ("use_noise", C.c_uint32),
                ("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("closest", C.c_uint64), ("shadow", C.c_uint64), ("shaded", C.c_uint64),
                ("nodes", C.c_uint64), ("tris", C.c_uint64)]


def build(force=False):
    so = os.path.join(_HERE, "liblpt_oracle.so")
    src = os.path.join(_HERE, "lpt_oracle.c")
    hdr = os.path.join(_HERE, "lpt_oracle.h")
    stale = (not os.path.exists(so)) or (os.path.exists(src) and max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


NATIVE_FLAGS = "-O3 -march=native -ffp-contract=off"   # oracle/Makefile NATIVE_CFLAGS
CHECKER_FLAGS = "-O2 -march=x86-64-v3 -ffp-contract=off"


def use_native_build():
    """bench.py's cpu_baseline leg only: from here on this process runs the oracle built with -O3 -march=native ON THIS HOST, made now.
    Returns the flags that are in effect (the checker's if the native build fails: no compiler, an unknown CPU).  Same source, same arithmetic
    (-ffp-contract=off either way)."""
    global _LIB
    so = os.path.join(_HERE, "liblpt_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liblpt_oracle_native.so"])
    except Exception:
        return CHECKER_FLAGS
    _LIB = None
    lib(so)
    return NATIVE_FLAGS


def lib(path=None):
    global _LIB
    if _LIB is None:
        L = C.CDLL(path or build())
        vp, u32, f32p = C.c_void_p, C.c_uint32, C.POINTER(C.c_float)
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [u32, vp, vp, u32, vp, u32, vp, u32, vp, u32, u32, vp]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_set_noise.argtypes = [vp, vp, u32, u32, u32]
        L.orc_trace_closest.argtypes = [vp, vp, vp, u32, vp, C.c_int, vp]
        L.orc_trace_occluded.argtypes = [vp, vp, vp, vp, u32, vp, C.c_int]
        L.orc_render.restype = u32
        L.orc_render.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
        L.orc_raygen.argtypes = [C.POINTER(RenderParams), u32, u32, f32p, f32p]
        L.orc_denoiser_create.restype = vp
        L.orc_denoiser_create.argtypes = [u32, u32]
        L.orc_denoiser_destroy.argtypes = [vp]
        L.orc_denoise_frame.argtypes = [vp, vp, C.POINTER(RenderParams), C.c_int, vp]
        L.orc_denoiser_read.argtypes = [vp, vp, vp, vp, vp]
        L.orc_resolve.argtypes = [vp, u32, vp]
        L.orc_tonemap.argtypes = [vp, u32, vp]
        L.orc_pcg_hash.restype = u32
        L.orc_pcg_hash.argtypes = [u32]
        L.orc_rng_stream.argtypes = [u32, u32, u32, u32, u32, vp]
        L.orc_sincos2pi.argtypes = [C.c_float, f32p, f32p]
        L.orc_atan2.restype = C.c_float
        L.orc_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_acos.restype = C.c_float
        L.orc_acos.argtypes = [C.c_float]
        L.orc_woop.argtypes = [vp, vp, vp, vp]
        L.orc_ray_triangle.restype = C.c_int
        L.orc_ray_triangle.argtypes = [vp, vp, vp, C.c_float, C.c_float, f32p, f32p, f32p]
        L.orc_onb.argtypes = [vp, vp, vp]
        L.orc_bsdf_eval.argtypes = [vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, f32p]
        L.orc_bsdf_sample.restype = C.c_int
        L.orc_bsdf_sample.argtypes = [vp, C.c_float, C.c_float, vp, vp, vp, C.c_float, C.c_float, C.c_float, vp, vp, f32p]
        L.orc_env_lookup.argtypes = [vp, vp, vp]
        L.orc_texture_lookup.argtypes = [vp, u32, C.c_float, C.c_float, C.c_int, vp]
        L.orc_srgb_lut.restype = C.c_float
        L.orc_srgb_lut.argtypes = [u32]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f3(v):
    return np.ascontiguousarray(v, np.float32)


class OracleScene:
    """World-space triangle soup + materials/lights/images/probe handed to the C oracle."""

    def __init__(self, tri_verts, tri_material, materials, lights, images=(), probe=None, noise=None):
        L = lib()
        self.tri_verts = np.ascontiguousarray(tri_verts, G.VERTEX_DT)
        self.tri_material = np.ascontiguousarray(tri_material, np.uint32)
        self.materials = np.ascontiguousarray(materials, G.MATERIAL_DT)
        self.lights = np.ascontiguousarray(lights, G.LIGHT_DT)
        self.images = [np.ascontiguousarray(i, np.uint8) for i in images]
        n_tris = self.tri_material.shape[0]
        imgs = (Image * max(1, len(self.images)))()
        for k, im in enumerate(self.images):
            imgs[k].width = im.shape[1]
            imgs[k].height = im.shape[0]
            imgs[k].rgba8 = im.ctypes.data
        pw = ph = 0
        pp = None
        if probe is not None:
            self.probe = np.ascontiguousarray(probe, np.uint8)
            ph, pw = self.probe.shape[0], self.probe.shape[1]
            pp = self.probe.ctypes.data
        self.h = L.orc_scene_create(n_tris, _p(self.tri_verts), _p(self.tri_material),
                                    self.materials.shape[0], _p(self.materials),
                                    self.lights.shape[0], _p(self.lights),
                                    len(self.images), C.cast(imgs, C.c_void_p), pw, ph, pp)
        if noise is not None:
            nz = np.ascontiguousarray(noise, np.uint8)
            L.orc_scene_set_noise(self.h, _p(nz), nz.shape[1], nz.shape[0], nz.shape[1] * 4)

    @classmethod
    def from_scene(cls, scene, probe=None, noise=None):
        tv, tm = G.bake(scene)
        return cls(tv, tm, scene.materials, scene.lights, scene.images, probe, noise)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def trace_closest(self, origins, dirs, brute_force=False, counters=None):
        o, d = f3(origins), f3(dirs)
        n = o.shape[0]
        out = np.zeros(n, HIT_DT)
        lib().orc_trace_closest(self.h, _p(o), _p(d), n, _p(out), int(brute_force),
                                C.addressof(counters) if counters is not None else None)
        return out

    def trace_occluded(self, origins, dirs, tmax, brute_force=False):
        o, d, t = f3(origins), f3(dirs), f3(tmax)
        out = np.zeros(o.shape[0], np.uint8)
        lib().orc_trace_occluded(self.h, _p(o), _p(d), _p(t), o.shape[0], _p(out), int(brute_force))
        return out

    def render(self, width, height, view, vfov, max_bounces, frames=1, user_seed=0, seed_counter=0,
               rank=0, world_size=1, tile_w=32, tile_h=8, threads=None, brute_force=False, use_noise=False,
               crop=None, want_counters=False):
        p = RenderParams()
        p.width, p.height = width, height
        p.view = (C.c_float * 16)(*[float(x) for x in np.asarray(view, np.float32).reshape(16)])
        p.vfov = vfov
        p.max_bounces, p.user_seed, p.seed_counter, p.frames = max_bounces, user_seed, seed_counter, frames
        p.rank, p.world_size, p.tile_w, p.tile_h = rank, world_size, tile_w, tile_h
        p.threads = threads if threads else (os.cpu_count() or 1)
        p.brute_force = int(brute_force)
        p.use_noise = int(use_noise)
        if crop is not None:
            p.x0, p.y0, p.x1, p.y1 = crop
        accum = np.zeros((height, width, 4), np.float32)
        cnt = Counters()
        seed_end = lib().orc_render(self.h, C.byref(p), _p(accum), C.addressof(cnt) if want_counters else None)
        self.last_seed = seed_end
        if want_counters:
            return accum, cnt
        return accum


class Denoiser:
    """BlitMode::DenoisedPathrace (mode 1) / Temporal (mode 2) frame sequence on the oracle (SPEC §15)."""

    def __init__(self, scene, width, height, vfov, max_bounces, user_seed=0):
        self.scene, self.w, self.h = scene, width, height
        self.vfov, self.bounces, self.user_seed = vfov, max_bounces, user_seed
        self.seed_counter = 0
        self.h_ = lib().orc_denoiser_create(width, height)

    def __del__(self):
        try:
            if getattr(self, "h_", None):
                lib().orc_denoiser_destroy(self.h_)
                self.h_ = None
        except Exception:
            pass

    def frame(self, view, mode=1):
        p = RenderParams()
        p.width, p.height = self.w, self.h
        p.view = (C.c_float * 16)(*[float(x) for x in np.asarray(view, np.float32).reshape(16)])
        p.vfov, p.max_bounces, p.user_seed, p.seed_counter, p.frames = self.vfov, self.bounces, self.user_seed, self.seed_counter, 1
        p.world_size = 1
        out = np.zeros((self.h, self.w, 4), np.float32)
        lib().orc_denoise_frame(self.h_, self.scene.h, C.byref(p), int(mode), _p(out))
        self.seed_counter += self.bounces
        return out

    def read(self):
        g, m = np.zeros((self.h, self.w, 4), np.uint32), np.zeros((self.h, self.w, 2), np.float32)
        rad, hist = np.zeros((self.h, self.w, 4), np.float32), np.zeros((self.h, self.w), np.uint32)
        lib().orc_denoiser_read(self.h_, _p(g), _p(m), _p(rad), _p(hist))
        return g, m, rad, hist


def resolve(accum):
    a = np.ascontiguousarray(accum, np.float32)
    out = np.zeros_like(a)
    lib().orc_resolve(_p(a), a.size // 4, _p(out))
    return out


def tonemap(accum):
    a = np.ascontiguousarray(accum, np.float32)
    out = np.zeros(a.shape[:-1] + (4,), np.uint8)
    lib().orc_tonemap(_p(a), a.size // 4, _p(out))
    return out


def raygen(width, height, view, vfov, x, y, user_seed=0, seed_counter=0):
    p = RenderParams()
    p.width, p.height = width, height
    p.view = (C.c_float * 16)(*[float(v) for v in np.asarray(view, np.float32).reshape(16)])
    p.vfov = vfov
    p.user_seed, p.seed_counter = user_seed, seed_counter
    o = (C.c_float * 3)()
    d = (C.c_float * 3)()
    lib().orc_raygen(C.byref(p), x, y, o, d)
    return np.array(o[:], np.float32), np.array(d[:], np.float32)
