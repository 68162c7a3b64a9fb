"""ctypes view of the C ABI declared in include/lpt.h (libloupiote_hip.so).

The library is built in-tree by ``loupiote_amd.build.build()`` (hipcc, gfx950).  Loading
fails loudly when it is missing — there is no Python or CPU fallback for any entry point.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LPT_LIB_PATH") or os.path.join(_HERE, "libloupiote_hip.so")   # LPT_LIB_PATH: A/B builds of experiments (csrc/Makefile `variant`)

LPT_OK = 0
LPT_ERR_FILE_NOT_FOUND = 1
LPT_ERR_READBACK = 2
LPT_ERR_ACCEL_BUILD = 3
LPT_ERR_HIP = 4
LPT_ERR_RCCL = 5
LPT_ERR_INVALID_ARG = 6
COMM_ID_BYTES = 128
# lpt_option / lpt_experiment (include/lpt.h): launch tuning behind lpt_renderer_set_option; every value gives the same frame.  The five options a host may
# sanely set, then the experiment knobs (LPT_OPT_EXPERIMENT(name) = 256 + name: A/B tools and variant tests, not a stable surface)
OPT_EXPERIMENT_BASE = 256
OPTIONS = {"packet_primary": 1, "wavefront_rays": 2, "path_rays": 3, "coop_rays": 4, "tail_lanes": 5}
EXPERIMENTS = {"pipe_rays": 0, "refill": 1, "trace_waves_per_cu": 2, "shade_blocks_per_cu": 3, "path_waves_per_cu": 4, "path_refill": 5, "occ_cell_milli": 6,
               "step_budget": 7, "budget_rays": 8, "packet_quads": 9, "split_rays": 10, "budget_split": 11}
OPTIONS.update({k: OPT_EXPERIMENT_BASE + v for k, v in EXPERIMENTS.items()})
EXCHANGE_GATHER_TILES = 0
HOST_FRAME_HOST_ONLY = 1
EXCHANGE_REDUCE = 1
INVALID_INDEX = 0xFFFFFFFF
LIGHT_BIT = 0x80000000

MATERIAL_DT = np.dtype([("color", "<f4", 4), ("roughness", "<f4"), ("reflectivity", "<f4"),
                        ("albedo_texture", "<u4"), ("mra_texture", "<u4")])
VERTEX_DT = np.dtype([("position", "<f4", 4), ("normal", "<f4", 4)])
LIGHT_DT = np.dtype([("normal", "<f4", 4), ("tangent", "<f4", 4), ("bitangent", "<f4", 4), ("origin", "<f4", 4)])
INSTANCE_DT = np.dtype([("model_to_world", "<f4", 16), ("blas_index", "<u4"), ("material_index", "<u4"),
                        ("pad", "<u4", 2)])
ENTRY_DT = np.dtype([("vertex_offset", "<u4"), ("vertex_count", "<u4"), ("index_offset", "<u4"),
                     ("index_count", "<u4")])
HIT_DT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4")])


class SceneCounts(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("materials", "entries", "vertices", "indices", "instances", "lights", "images")]


class AccelStats(C.Structure):
    _fields_ = [("triangles", C.c_uint32), ("nodes", C.c_uint32), ("node_bytes", C.c_uint32),
                ("tri_bytes", C.c_uint32), ("max_depth", C.c_uint32), ("build_ms", C.c_float),
                ("host_baked_triangles", C.c_uint32), ("upload_ms", C.c_float),
                ("texture_pairs", C.c_uint32), ("texture_bytes_resident", C.c_uint64)]


class RayCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("closest", "shadow", "shaded", "nodes", "tris", "shadow_nodes", "shadow_tris",
                                             "wave_steps", "live_lanes", "node_lanes", "tri_lanes", "primary", "packet_nodes", "packet_tris",
                                             "shadow_occluded", "occluder_cache_found", "occluder_cache_hits", "wave_rays")]


class Timing(C.Structure):
    _fields_ = [("label", C.c_char * 32), ("ms", C.c_float), ("launches", C.c_uint32)]


_vp, _u32, _i, _f, _sz = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_size_t
_pvp = C.POINTER(C.c_void_p)
_pu32 = C.POINTER(C.c_uint32)

# name -> (restype, argtypes); every symbol include/lpt.h declares
SIGNATURES = {
    "lpt_last_error": (C.c_char_p, []),
    "lpt_status_string": (C.c_char_p, [_i]),
    "lpt_abi_version": (_u32, []),
    "lpt_device_create": (_i, [_i, _pvp]),
    "lpt_device_destroy": (_i, [_vp]),
    "lpt_device_synchronize": (_i, [_vp]),
    "lpt_device_info": (_i, [_vp, C.c_char_p, _sz, C.POINTER(_i)]),
    "lpt_device_stream": (_i, [_vp, _pvp]),
    "lpt_scene_create": (_i, [_pvp]),
    "lpt_scene_destroy": (_i, [_vp]),
    "lpt_scene_counts_get": (_i, [_vp, C.POINTER(SceneCounts)]),
    "lpt_scene_add_mesh": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _u32, _vp, _u32, _pu32]),
    "lpt_scene_add_instance": (_i, [_vp, _u32, _vp, _u32, _pu32]),
    "lpt_scene_set_instance_transform": (_i, [_vp, _u32, _vp]),
    "lpt_scene_add_material": (_i, [_vp, _vp, _pu32]),
    "lpt_scene_add_image": (_i, [_vp, _vp, _u32, _u32, _pu32]),
    "lpt_scene_add_light": (_i, [_vp, _vp, _pu32]),
    "lpt_scene_set_light": (_i, [_vp, _u32, _vp]),
    "lpt_light_default": (_i, [_vp]),
    "lpt_scene_get_materials": (_i, [_vp, _u32, _u32, _vp]),
    "lpt_scene_get_entries": (_i, [_vp, _u32, _u32, _vp]),
    "lpt_scene_get_vertices": (_i, [_vp, _u32, _u32, _vp]),
    "lpt_scene_get_indices": (_i, [_vp, _u32, _u32, _vp]),
    "lpt_scene_get_instances": (_i, [_vp, _u32, _u32, _vp]),
    "lpt_scene_get_lights": (_i, [_vp, _u32, _u32, _vp]),
    "lpt_scene_get_image": (_i, [_vp, _u32, _pu32, _pu32, _vp]),
    "lpt_load_gltf": (_i, [_vp, _vp, _sz]),
    "lpt_load_gltf_path": (_i, [_vp, C.c_char_p]),
    "lpt_decode_hdr": (_i, [_vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "lpt_decode_image": (_i, [_vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "lpt_write_hdr": (_i, [C.c_char_p, _vp, _u32, _u32, _sz]),
    "lpt_write_png": (_i, [C.c_char_p, _vp, _u32, _u32, _sz]),
    "lpt_scene_upload": (_i, [_vp, _vp, _pvp]),
    "lpt_scene_upload_ex": (_i, [_vp, _vp, _u32, _pvp]),
    "lpt_scene_gpu_destroy": (_i, [_vp]),
    "lpt_scene_gpu_stats": (_i, [_vp, C.POINTER(AccelStats)]),
    "lpt_scene_gpu_rebuild": (_i, [_vp, _vp]),
    "lpt_scene_gpu_update_instances": (_i, [_vp, _vp, C.POINTER(C.c_uint32)]),
    "lpt_probe_upload": (_i, [_vp, _vp, _u32, _u32, _pvp]),
    "lpt_probe_destroy": (_i, [_vp]),
    "lpt_trace_closest": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "lpt_trace_occluded": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "lpt_renderer_create": (_i, [_vp, _u32, _u32, _pvp]),
    "lpt_renderer_destroy": (_i, [_vp]),
    "lpt_renderer_set_downsample": (_i, [_vp, _f]),
    "lpt_renderer_resize": (_i, [_vp, _vp, _vp, _u32, _u32]),
    "lpt_renderer_get_size": (_i, [_vp, _pu32, _pu32]),
    "lpt_max_per_pixel_bytes": (_u32, []),
    "lpt_renderer_set_resources": (_i, [_vp, _vp, _vp]),
    "lpt_renderer_raytrace": (_i, [_vp, _vp]),
    "lpt_renderer_raytrace_n": (_i, [_vp, _vp, _u32]),
    "lpt_renderer_submit": (_i, [_vp]),
    "lpt_renderer_set_max_fused": (_i, [_vp, _u32]),
    "lpt_renderer_get_submission_stats": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _pu32]),
    "lpt_host_alloc": (_i, [_sz, _pvp]),
    "lpt_host_free": (_i, [_vp]),
    "lpt_host_register": (_i, [_vp, _sz]),
    "lpt_host_unregister": (_i, [_vp]),
    "lpt_renderer_read_radiance_owned": (_i, [_vp, _vp]),
    "lpt_host_frame_create": (_i, [C.c_char_p, _u32, _u32, _u32, _u32, _pvp]),
    "lpt_host_frame_attach": (_i, [C.c_char_p, _u32, _u32, _u32, _u32, _pvp]),
    "lpt_host_frame_ptr": (_i, [_vp, _pvp]),
    "lpt_host_frame_barrier": (_i, [_vp, _u32, _u32, _u32]),
    "lpt_host_frame_destroy": (_i, [_vp]),
    "lpt_renderer_reset_accumulation": (_i, [_vp]),
    "lpt_renderer_set_accumulate": (_i, [_vp, _i]),
    "lpt_renderer_get_accumulate": (_i, [_vp, C.POINTER(_i)]),
    "lpt_renderer_get_frame_state": (_i, [_vp, _pu32, _pu32]),
    "lpt_renderer_upload_noise": (_i, [_vp, _vp, _u32, _u32, _u32]),
    "lpt_renderer_use_noise": (_i, [_vp, _i]),
    "lpt_renderer_set_blit_mode": (_i, [_vp, _i]),
    "lpt_renderer_blit_rgba8": (_i, [_vp, _vp, _sz]),
    "lpt_renderer_read_pixels": (_i, [_vp, _vp]),
    "lpt_renderer_read_radiance": (_i, [_vp, _vp]),
    "lpt_renderer_read_denoiser": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "lpt_renderer_get_timings": (_i, [_vp, _vp, C.POINTER(_i)]),
    "lpt_renderer_enable_timings": (_i, [_vp, _i]),
    "lpt_renderer_set_max_bounces": (_i, [_vp, _u32]),
    "lpt_renderer_set_seed": (_i, [_vp, _u32]),
    "lpt_renderer_set_vfov": (_i, [_vp, _f]),
    "lpt_renderer_set_shard": (_i, [_vp, _u32, _u32, _u32, _u32]),
    "lpt_renderer_radiance_device_ptr": (_i, [_vp, _pvp, C.POINTER(_sz)]),
    "lpt_renderer_denoiser_inputs": (_i, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "lpt_renderer_denoise_filter": (_i, [_vp]),
    "lpt_comm_unique_id": (_i, [_vp]),
    "lpt_comm_create": (_i, [_vp, _vp, _i, _i, _pvp]),
    "lpt_comm_destroy": (_i, [_vp]),
    "lpt_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "lpt_comm_group_begin": (_i, []),
    "lpt_comm_group_end": (_i, []),
    "lpt_renderer_set_comm": (_i, [_vp, _vp]),
    "lpt_renderer_exchange": (_i, [_vp, _i]),
    "lpt_renderer_exchange_local": (_i, [_vp, _pvp, _i]),
    "lpt_shard_layout": (_i, [_u32, _u32, _u32, _u32, _u32, _u32, _pu32, _pu32]),
    "lpt_shard_layout_weighted": (_i, [_u32, _u32, _u32, _u32, _u32, _u32, _vp, _pu32, _pu32]),
    "lpt_shard_owner": (_i, [_u32, _vp, _u32, _pu32]),
    "lpt_renderer_set_shard_weighted": (_i, [_vp, _u32, _u32, _u32, _u32, _vp]),
    "lpt_renderer_set_comm_weighted": (_i, [_vp, _vp, _vp]),
    "lpt_renderer_set_lanes": (_i, [_vp, _i]),
    "lpt_renderer_set_sort_queues": (_i, [_vp, _i]),
    "lpt_renderer_set_option": (_i, [_vp, _i, C.c_uint64]),
    "lpt_renderer_get_option": (_i, [_vp, _i, C.POINTER(C.c_uint64)]),
    "lpt_renderer_get_queue_counts": (_i, [_vp, _vp, _vp, _u32]),
    "lpt_renderer_get_ray_counts": (_i, [_vp, C.POINTER(RayCounts)]),
    "lpt_renderer_reset_ray_counts": (_i, [_vp]),
    "lpt_renderer_get_step_histogram": (_i, [_vp, _pu32, _vp]),
    "lpt_renderer_enable_stats": (_i, [_vp, _i]),
    "lpt_renderer_synchronize": (_i, [_vp]),
    "lpt_renderer_stream": (_i, [_vp, _pvp]),
}

_LIB = None


def lib():
    """Load libloupiote_hip.so; raises if the HIP extension has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "loupiote_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no fallback path." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("LPT_ABI_LENIENT") and not hasattr(L, name):
                continue  # A/B runs against an older build of the library (experiments only)
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
