"""Host-side mirror of the reference's `loupiote-core` API over the C ABI.

Same names, argument meaning and error behaviour as reference crates/lib/src:
``Device`` (device.rs:72-141), ``Scene`` / ``SceneGPU`` / ``ProbeGPU`` (scene.rs:30-188),
``Renderer`` / ``BlitMode`` (renderer.rs:160-811), ``Error`` (errors.rs:1-20) and
``loaders.load_gltf`` (loaders/gltf.rs:46-161).  wgpu handles (device, queue, encoder,
texture view) have no meaning here and are dropped from the signatures; everything else is
forwarded 1:1 to ``lpt_*``.  No arithmetic happens in this file.
"""
import ctypes as C
import enum

import numpy as np

from . import _abi as A


class Error(Exception):
    """errors.rs:2-6 — FileNotFound / TextureToBufferReadFail / AccelBuild (+ the ABI's own codes)."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status
        self.kind = {A.LPT_ERR_FILE_NOT_FOUND: "FileNotFound", A.LPT_ERR_READBACK: "TextureToBufferReadFail",
                     A.LPT_ERR_ACCEL_BUILD: "AccelBuild", A.LPT_ERR_HIP: "Hip", A.LPT_ERR_RCCL: "Rccl",
                     A.LPT_ERR_INVALID_ARG: "InvalidArg"}.get(status, "Unknown")


def _check(status):
    if status != A.LPT_OK:
        raise Error(status, A.lib().lpt_last_error().decode("utf-8", "replace"))


class BlitMode(enum.IntEnum):
    """renderer.rs:160-167 (spelling `Pahtrace` is the reference's)."""
    Pahtrace = 0
    DenoisedPathrace = 1
    Temporal = 2
    GBuffer = 3
    MotionVector = 4


class Device:
    """device.rs:80 `Device::new(wgpu::Device)` -> one HIP device + the stream all work is enqueued on."""

    def __init__(self, hip_ordinal=0):
        h = C.c_void_p()
        _check(A.lib().lpt_device_create(int(hip_ordinal), C.byref(h)))
        self._h = h

    def inner(self):
        return self._h

    def info(self):
        name = C.create_string_buffer(128)
        cus = C.c_int()
        _check(A.lib().lpt_device_info(self._h, name, 128, C.byref(cus)))
        return name.value.decode(), cus.value

    def stream(self):
        s = C.c_void_p()
        _check(A.lib().lpt_device_stream(self._h, C.byref(s)))
        return s.value

    def synchronize(self):
        _check(A.lib().lpt_device_synchronize(self._h))

    def close(self):
        if self._h:
            A.lib().lpt_device_destroy(self._h)
            self._h = None


class Comm:
    """One rank of the frame exchange (new functionality, include/lpt.h "multi-GPU frame exchange"): a plain RCCL
    communicator owned by the library.  `Comm.unique_id()` on one rank, ship the 128 bytes, `Comm(device, id, rank, world)`
    on every rank (collective)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(A.COMM_ID_BYTES)
        _check(A.lib().lpt_comm_unique_id(buf))
        return buf.raw

    def __init__(self, device, unique_id, rank, world_size):
        if len(unique_id) != A.COMM_ID_BYTES:
            raise Error(A.LPT_ERR_INVALID_ARG, "unique id must be %d bytes" % A.COMM_ID_BYTES)
        h = C.c_void_p()
        _check(A.lib().lpt_comm_create(device.inner(), C.c_char_p(bytes(unique_id)), int(rank), int(world_size), C.byref(h)))
        self._h = h

    def info(self):
        r, w = C.c_int(), C.c_int()
        _check(A.lib().lpt_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    @staticmethod
    def group_begin():
        """ncclGroupStart for a thread that drives several communicators: the exchanges inside the bracket are issued together"""
        _check(A.lib().lpt_comm_group_begin())

    @staticmethod
    def group_end():
        """the outermost end issues the RCCL operations and then enqueues what consumes the received data"""
        _check(A.lib().lpt_comm_group_end())

    def close(self):
        if self._h:
            A.lib().lpt_comm_destroy(self._h)
            self._h = None


class Scene:
    """scene.rs:30-54 — `Scene::default()` holds one dummy element in every array."""

    def __init__(self):
        h = C.c_void_p()
        _check(A.lib().lpt_scene_create(C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                A.lib().lpt_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def counts(self):
        c = A.SceneCounts()
        _check(A.lib().lpt_scene_counts_get(self._h, C.byref(c)))
        return c

    # BLASArray::add_bvh / add_bvh_indexed (gltf.rs:97-105)
    def add_mesh(self, positions, normals=None, uvs=None, indices=None):
        pos = np.ascontiguousarray(positions, np.float32)
        n = pos.shape[0]
        nrm = None if normals is None else np.ascontiguousarray(normals, np.float32)
        uv = None if uvs is None else np.ascontiguousarray(uvs, np.float32)
        idx = None if indices is None else np.ascontiguousarray(indices, np.uint32).reshape(-1)
        out = C.c_uint32()
        _check(A.lib().lpt_scene_add_mesh(self._h, A.ptr(pos), pos.strides[0], A.ptr(nrm), 0 if nrm is None else nrm.strides[0],
                                          A.ptr(uv), 0 if uv is None else uv.strides[0], n, A.ptr(idx),
                                          0 if idx is None else idx.size, C.byref(out)))
        return out.value

    # BLASArray::add_instance (gltf.rs:141-145)
    def add_instance(self, blas_index, model_to_world, material_index):
        m = np.ascontiguousarray(model_to_world, np.float32).reshape(16)
        out = C.c_uint32()
        _check(A.lib().lpt_scene_add_instance(self._h, int(blas_index), A.ptr(m), int(material_index), C.byref(out)))
        return out.value

    def set_instance_transform(self, index, model_to_world):
        m = np.ascontiguousarray(model_to_world, np.float32).reshape(16)
        _check(A.lib().lpt_scene_set_instance_transform(self._h, int(index), A.ptr(m)))

    def add_material(self, color, roughness, reflectivity, albedo_texture=A.INVALID_INDEX, mra_texture=A.INVALID_INDEX):
        m = np.zeros(1, A.MATERIAL_DT)
        m["color"] = color
        m["roughness"] = roughness
        m["reflectivity"] = reflectivity
        m["albedo_texture"] = albedo_texture
        m["mra_texture"] = mra_texture
        out = C.c_uint32()
        _check(A.lib().lpt_scene_add_material(self._h, A.ptr(m), C.byref(out)))
        return out.value

    def add_image(self, rgba8):
        im = np.ascontiguousarray(rgba8, np.uint8)
        out = C.c_uint32()
        _check(A.lib().lpt_scene_add_image(self._h, A.ptr(im), im.shape[1], im.shape[0], C.byref(out)))
        return out.value

    def add_light(self, light):
        l = np.ascontiguousarray(light, A.LIGHT_DT).reshape(1)
        out = C.c_uint32()
        _check(A.lib().lpt_scene_add_light(self._h, A.ptr(l), C.byref(out)))
        return out.value

    def set_light(self, index, light):
        l = np.ascontiguousarray(light, A.LIGHT_DT).reshape(1)
        _check(A.lib().lpt_scene_set_light(self._h, int(index), A.ptr(l)))

    def _get(self, fn, dt, count):
        out = np.zeros(count, dt)
        _check(fn(self._h, 0, count, A.ptr(out)))
        return out

    @property
    def materials(self):
        return self._get(A.lib().lpt_scene_get_materials, A.MATERIAL_DT, self.counts().materials)

    @property
    def entries(self):
        return self._get(A.lib().lpt_scene_get_entries, A.ENTRY_DT, self.counts().entries)

    @property
    def vertices(self):
        return self._get(A.lib().lpt_scene_get_vertices, A.VERTEX_DT, self.counts().vertices)

    @property
    def indices(self):
        return self._get(A.lib().lpt_scene_get_indices, np.uint32, self.counts().indices)

    @property
    def instances(self):
        return self._get(A.lib().lpt_scene_get_instances, A.INSTANCE_DT, self.counts().instances)

    @property
    def lights(self):
        return self._get(A.lib().lpt_scene_get_lights, A.LIGHT_DT, self.counts().lights)

    def image(self, index):
        w, h = C.c_uint32(), C.c_uint32()
        _check(A.lib().lpt_scene_get_image(self._h, index, C.byref(w), C.byref(h), None))
        out = np.zeros((h.value, w.value, 4), np.uint8)
        _check(A.lib().lpt_scene_get_image(self._h, index, None, None, A.ptr(out)))
        return out


def default_light():
    """`Light::new()` (scene.rs:50)."""
    l = np.zeros(1, A.LIGHT_DT)
    _check(A.lib().lpt_light_default(A.ptr(l)))
    return l


class SceneGPU:
    """scene.rs:151 `SceneGPU::new_from_scene(&Scene, &Device, &Queue)`."""

    def __init__(self, handle, device):
        self._h = handle
        self._dev = device

    @classmethod
    def new_from_scene(cls, scene, device, gpu_build=False, pair_textures=True):
        """scene.rs:151 `SceneGPU::new_from_scene`; gpu_build=True builds the BVH on the GPU (Morton radix tree, a few
        ms, lower quality) instead of the host SAH builder — for scenes that are rebuilt every frame; pair_textures=False
        keeps every image on its own in the atlas (LPT_UPLOAD_NO_TEXTURE_PAIRS; experiments: the frame is the same)"""
        h = C.c_void_p()
        flags = (1 if gpu_build else 0) | (0 if pair_textures else 0x100)
        _check(A.lib().lpt_scene_upload_ex(device.inner(), scene._h, flags, C.byref(h)))
        return cls(h, device)

    def stats(self):
        s = A.AccelStats()
        _check(A.lib().lpt_scene_gpu_stats(self._h, C.byref(s)))
        return s

    def update_instances(self, scene):
        """after `scene.set_instance_transform(...)` (standalone/src/lib.rs:118-121): re-bake the changed instances
        and refit the wide BVH on the GPU; returns how many instances were re-baked"""
        n = C.c_uint32(0)
        _check(A.lib().lpt_scene_gpu_update_instances(self._h, scene._h, C.byref(n)))
        return int(n.value)

    def rebuild(self, scene):
        """re-bake every instance and rebuild the BVH on the GPU (for edits too large for a refit)"""
        _check(A.lib().lpt_scene_gpu_rebuild(self._h, scene._h))

    def trace_closest(self, origins, dirs):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        out = np.zeros(o.shape[0], A.HIT_DT)
        _check(A.lib().lpt_trace_closest(self._dev.inner(), self._h, A.ptr(o), A.ptr(d), o.shape[0], A.ptr(out)))
        return out

    def trace_occluded(self, origins, dirs, tmax):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        t = np.ascontiguousarray(tmax, np.float32)
        out = np.zeros(o.shape[0], np.uint8)
        _check(A.lib().lpt_trace_occluded(self._dev.inner(), self._h, A.ptr(o), A.ptr(d), A.ptr(t), o.shape[0], A.ptr(out)))
        return out

    def close(self):
        if self._h:
            A.lib().lpt_scene_gpu_destroy(self._h)
            self._h = None


class ProbeGPU:
    """scene.rs:72 `ProbeGPU::new(device, queue, data, width, height)` — RGBE8, 4 bytes per pixel."""

    def __init__(self, device, data, width, height):
        buf = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data, np.uint8)
        if buf.size != width * height * 4:
            raise Error(A.LPT_ERR_INVALID_ARG, "probe data must be width*height*4 bytes")
        h = C.c_void_p()
        _check(A.lib().lpt_probe_upload(device.inner(), A.ptr(buf), width, height, C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            A.lib().lpt_probe_destroy(self._h)
            self._h = None


# launch options (`Renderer.set_option`) every new Renderer of this process starts with: how the test suite runs one test body over
# both forms of the frame pipeline (tests/conftest.py `pipeline`) and bench.py applies `--opt name=value`.  Host-side only: the
# library itself reads no environment variable and has no process-wide state for this.
DEFAULT_OPTIONS = {}


class Renderer:
    """renderer.rs:169-811."""

    def __init__(self, device, original_size, swapchain_format=None):
        h = C.c_void_p()
        _check(A.lib().lpt_renderer_create(device.inner(), int(original_size[0]), int(original_size[1]), C.byref(h)))
        self._h = h
        self._dev = device
        self._downsample = 0.5
        for k, v in DEFAULT_OPTIONS.items():
            self.set_option(k, v)

    @staticmethod
    def max_ssbo_element_in_bytes():
        return A.lib().lpt_max_per_pixel_bytes()

    # pub downsample_factor (renderer.rs:203)
    @property
    def downsample_factor(self):
        return self._downsample

    @downsample_factor.setter
    def downsample_factor(self, f):
        _check(A.lib().lpt_renderer_set_downsample(self._h, float(f)))
        self._downsample = float(f)

    # pub accumulate (renderer.rs:204)
    @property
    def accumulate(self):
        a = C.c_int()
        _check(A.lib().lpt_renderer_get_accumulate(self._h, C.byref(a)))
        return bool(a.value)

    @accumulate.setter
    def accumulate(self, flag):
        _check(A.lib().lpt_renderer_set_accumulate(self._h, int(bool(flag))))

    def resize(self, device, scene_resources, probe, size):
        _check(A.lib().lpt_renderer_resize(self._h, scene_resources._h, probe._h if probe is not None else None,
                                           int(size[0]), int(size[1])))

    def get_size(self):
        w, h = C.c_uint32(), C.c_uint32()
        _check(A.lib().lpt_renderer_get_size(self._h, C.byref(w), C.byref(h)))
        return (w.value, h.value)

    def set_resources(self, device, scene_resources, probe=None):
        _check(A.lib().lpt_renderer_set_resources(self._h, scene_resources._h, probe._h if probe is not None else None))

    def raytrace(self, view_transform):
        m = np.ascontiguousarray(view_transform, np.float32).reshape(16)
        _check(A.lib().lpt_renderer_raytrace(self._h, A.ptr(m)))

    def raytrace_n(self, view_transform, n_samples):
        """build-only: n x { raytrace(view); accumulate = true } as one wavefront (bit-identical result)"""
        m = np.ascontiguousarray(view_transform, np.float32).reshape(16)
        _check(A.lib().lpt_renderer_raytrace_n(self._h, A.ptr(m), int(n_samples)))

    def submit(self):
        """queue.submit(encoder.finish()) (app.rs:335-337): launch what raytrace() has recorded; asynchronous"""
        _check(A.lib().lpt_renderer_submit(self._h))

    def set_max_fused(self, n):
        """0 = automatic (recorded calls wait for the next submission point and leave as wavefronts of about 4 M rays: runs of tile
        rows with all the samples); n >= 1: n calls are one whole-frame wavefront, launched when the n-th is recorded (1 = at once)"""
        _check(A.lib().lpt_renderer_set_max_fused(self._h, int(n)))

    def submission_stats(self):
        """(raytrace() calls recorded, wavefronts launched for them, calls recorded but not yet submitted) — host state only"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint32()
        _check(A.lib().lpt_renderer_get_submission_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def reset_accumulation(self):
        _check(A.lib().lpt_renderer_reset_accumulation(self._h))

    def upload_noise_texture(self, data, width, height, bytes_per_row):
        buf = np.ascontiguousarray(data, np.uint8)
        _check(A.lib().lpt_renderer_upload_noise(self._h, A.ptr(buf), width, height, bytes_per_row))

    def use_noise_texture(self, flag):
        _check(A.lib().lpt_renderer_use_noise(self._h, int(bool(flag))))

    def set_blit_mode(self, mode):
        _check(A.lib().lpt_renderer_set_blit_mode(self._h, int(mode)))

    def blit(self, row_bytes=None):
        """tonemapped sRGB RGBA8 of the current target; `row_bytes` >= w * 4: a destination with padded rows (the padding stays zero)"""
        w, h = self.get_size()
        pitch = w * 4 if row_bytes is None else int(row_bytes)
        buf = np.zeros((h, pitch), np.uint8)
        _check(A.lib().lpt_renderer_blit_rgba8(self._h, A.ptr(buf), pitch))
        return buf[:, :w * 4].reshape(h, w, 4) if row_bytes is None else buf

    def read_pixels(self):
        w, h = self.get_size()
        out = np.zeros((h, w, 4), np.uint8)
        _check(A.lib().lpt_renderer_read_pixels(self._h, A.ptr(out)))
        return out

    # ---- build-only extensions
    def read_radiance(self, out=None):
        """mean radiance (h, w, 4) float32.  `out`: a destination to fill instead of a fresh array — e.g. a `pinned_array`
        (page-locked: one DMA at link speed instead of the runtime's staged copy into pageable memory)"""
        w, h = self.get_size()
        if out is None:
            out = np.empty((h, w, 4), np.float32)
        elif out.shape != (h, w, 4) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("read_radiance: out must be a C-contiguous float32 array of shape (%d, %d, 4)" % (h, w))
        _check(A.lib().lpt_renderer_read_radiance(self._h, A.ptr(out)))
        return out

    def read_radiance_owned(self, out):
        """host-side gather of a tile-sharded frame: this rank's OWNED pixels of the mean radiance straight into `out`, a whole-frame (h, w, 4)
        float32 array in page-locked memory (`pinned_array`, or memory passed to `host_register` — a shared-memory frame every rank maps);
        the other ranks' pixels are left alone"""
        w, h = self.get_size()
        if out.shape != (h, w, 4) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("read_radiance_owned: out must be a C-contiguous float32 array of shape (%d, %d, 4)" % (h, w))
        _check(A.lib().lpt_renderer_read_radiance_owned(self._h, A.ptr(out)))
        return out

    def read_denoiser(self):
        """current G-buffer / motion / accumulated radiance+variance / history of the ASVGF path"""
        w, h = self.get_size()
        g, m = np.zeros((h, w, 4), np.uint32), np.zeros((h, w, 2), np.float32)
        rad, hist = np.zeros((h, w, 4), np.float32), np.zeros((h, w), np.uint32)
        _check(A.lib().lpt_renderer_read_denoiser(self._h, A.ptr(g), A.ptr(m), A.ptr(rad), A.ptr(hist)))
        return g, m, rad, hist

    def denoiser_inputs(self):
        """device pointers of this frame's filter inputs for the multi-GPU exchange:
        (noisy float4*, gbuffer uint4*, motion float2*, n_pixels)"""
        n, g, m, c = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t()
        _check(A.lib().lpt_renderer_denoiser_inputs(self._h, C.byref(n), C.byref(g), C.byref(m), C.byref(c)))
        return n.value, g.value, m.value, c.value

    def denoise_filter(self):
        """rank 0 of a sharded frame, after the inputs have been summed over the ranks: temporal + a-trous + composite"""
        _check(A.lib().lpt_renderer_denoise_filter(self._h))

    def frame_state(self):
        fc, seed = C.c_uint32(), C.c_uint32()
        _check(A.lib().lpt_renderer_get_frame_state(self._h, C.byref(fc), C.byref(seed)))
        return fc.value, seed.value

    def set_max_bounces(self, n):
        _check(A.lib().lpt_renderer_set_max_bounces(self._h, int(n)))

    def set_seed(self, s):
        _check(A.lib().lpt_renderer_set_seed(self._h, int(s)))

    def set_vfov(self, radians):
        _check(A.lib().lpt_renderer_set_vfov(self._h, float(radians)))

    def set_shard(self, rank, world_size, tile_w=32, tile_h=8, weights=None):
        """this process traces the tiles of `rank`; `weights` (one small integer per rank, the same everywhere) gives ranks unequal
        shares — e.g. fewer tiles for the rank that also assembles and reads back the frame"""
        w = None if weights is None else np.ascontiguousarray(weights, np.uint32)
        if w is not None and w.size != world_size:
            raise ValueError("weights: one entry per rank")
        _check(A.lib().lpt_renderer_set_shard_weighted(self._h, rank, world_size, tile_w, tile_h, A.ptr(w)))

    def set_comm(self, comm, weights=None):
        """bind a `Comm` (None unbinds); implies set_shard(rank, world, 32, 8, weights)"""
        w = None if weights is None else np.ascontiguousarray(weights, np.uint32)
        if w is not None and (comm is None or w.size != comm.info()[1]):
            raise Error(A.LPT_ERR_INVALID_ARG, "set_comm: one weight per rank of the communicator")   # the C side reads weights[0 .. world)
        _check(A.lib().lpt_renderer_set_comm_weighted(self._h, comm._h if comm is not None else None, A.ptr(w)))

    def exchange(self, mode=A.EXCHANGE_GATHER_TILES):
        """combine the ranks' frames into rank 0's presented frame (RCCL, asynchronous on the renderer's stream)"""
        _check(A.lib().lpt_renderer_exchange(self._h, int(mode)))

    def exchange_local(self, peers):
        """the same exchange among sharded renderers of this process (self = rank 0) without RCCL"""
        arr = (C.c_void_p * max(len(peers), 1))(*[p._h for p in peers])
        _check(A.lib().lpt_renderer_exchange_local(self._h, arr, len(peers)))

    def set_lanes(self, lanes):
        """wavefront lanes (1..4, default 2): consecutive raytrace() calls overlap on their own streams; bit-identical results"""
        _check(A.lib().lpt_renderer_set_lanes(self._h, int(lanes)))

    def set_sort_queues(self, flag):
        """the shading pass emits both ray queues ordered by direction octant inside each block (bit-identical results)"""
        _check(A.lib().lpt_renderer_set_sort_queues(self._h, int(flag)))

    def set_option(self, name, value):
        """launch tuning (`_abi.OPTIONS`: the five `lpt_option` values — packet_primary, wavefront_rays, path_rays, coop_rays, tail_lanes — and, by
        name, the `LPT_OPT_EXPERIMENT` knobs of the A/B tools and variant tests); every value gives the same frame bit for bit"""
        _check(A.lib().lpt_renderer_set_option(self._h, A.OPTIONS[name] if isinstance(name, str) else int(name), int(value)))

    def get_option(self, name):
        v = C.c_uint64()
        _check(A.lib().lpt_renderer_get_option(self._h, A.OPTIONS[name] if isinstance(name, str) else int(name), C.byref(v)))
        return v.value

    def step_histogram(self):
        """stats kernels only: (longest ray in traversal steps, histogram by power of two) of the last wavefront's per-bounce traversal launches"""
        mx = C.c_uint32()
        h = np.zeros(12, np.uint32)
        _check(A.lib().lpt_renderer_get_step_histogram(self._h, C.byref(mx), A.ptr(h)))
        return mx.value, h

    def queue_counts(self, n=64):
        """per-bounce (closest-hit, shadow) queue sizes of the last traced frame"""
        c, s = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        _check(A.lib().lpt_renderer_get_queue_counts(self._h, A.ptr(c), A.ptr(s), n))
        return c, s

    def radiance_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        _check(A.lib().lpt_renderer_radiance_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def ray_counts(self):
        c = A.RayCounts()
        _check(A.lib().lpt_renderer_get_ray_counts(self._h, C.byref(c)))
        return c

    def reset_ray_counts(self):
        _check(A.lib().lpt_renderer_reset_ray_counts(self._h))

    def enable_stats(self, flag):
        _check(A.lib().lpt_renderer_enable_stats(self._h, int(bool(flag))))

    def enable_timings(self, flag):
        _check(A.lib().lpt_renderer_enable_timings(self._h, int(bool(flag))))

    def timings(self):
        n = C.c_int(16)
        arr = (A.Timing * 16)()
        _check(A.lib().lpt_renderer_get_timings(self._h, arr, C.byref(n)))
        return {arr[i].label.decode(): (arr[i].ms, arr[i].launches) for i in range(min(n.value, 16))}

    def stream(self):
        s = C.c_void_p()
        _check(A.lib().lpt_renderer_stream(self._h, C.byref(s)))
        return s.value

    def synchronize(self):
        _check(A.lib().lpt_renderer_synchronize(self._h))

    def close(self):
        if self._h:
            A.lib().lpt_renderer_destroy(self._h)
            self._h = None


class _PinnedBlock:
    """owner of one lpt_host_alloc block (freed when the last array viewing it dies)"""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _check(A.lib().lpt_host_alloc(int(nbytes), C.byref(p)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                A.lib().lpt_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class HostFrame:
    """The host-side gather of a tile-sharded frame (lpt_host_frame_*): ONE whole frame in POSIX shared memory that every rank of a node maps and
    page-locks.  Rank 0: `HostFrame.create(name, w, h, world)`; the others, once rank 0 has: `HostFrame.attach(...)`.  Per frame every rank calls
    `renderer.read_radiance_owned(frame.array)` (its own pixels, over its own PCIe link) and then `frame.barrier(rank, frame_no)`; after the
    barrier `frame.array` (h, w, 4 float32) is the complete frame on every rank.  `host_only`: a participant without a GPU (it only reads)."""

    def __init__(self, handle, width, height, world):
        self._h, self.width, self.height, self.world = handle, width, height, world
        p = C.c_void_p()
        _check(A.lib().lpt_host_frame_ptr(self._h, C.byref(p)))
        buf = (C.c_float * (width * height * 4)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=np.float32).reshape(height, width, 4)
        self.frame_no = 0

    @classmethod
    def create(cls, name, width, height, world, host_only=False):
        h = C.c_void_p()
        _check(A.lib().lpt_host_frame_create(name.encode(), width, height, world, A.HOST_FRAME_HOST_ONLY if host_only else 0, C.byref(h)))
        return cls(h, width, height, world)

    @classmethod
    def attach(cls, name, width, height, world, host_only=False):
        h = C.c_void_p()
        _check(A.lib().lpt_host_frame_attach(name.encode(), width, height, world, A.HOST_FRAME_HOST_ONLY if host_only else 0, C.byref(h)))
        return cls(h, width, height, world)

    def barrier(self, rank, frame_no=None, timeout_ms=60000):
        """frame_no None: this object's own counter (every rank calls barrier once per frame)"""
        if frame_no is None:
            self.frame_no += 1
            frame_no = self.frame_no
        _check(A.lib().lpt_host_frame_barrier(self._h, rank, frame_no, timeout_ms))

    def close(self):
        if self._h:
            self.array = None
            A.lib().lpt_host_frame_destroy(self._h)
            self._h = None


def host_register(array):
    """page-lock and map an existing C-contiguous numpy array (e.g. over a shared-memory segment) for the device: lpt_host_register"""
    _check(A.lib().lpt_host_register(A.ptr(array), array.nbytes))


def host_unregister(array):
    _check(A.lib().lpt_host_unregister(A.ptr(array)))


def pinned_array(shape, dtype=np.float32):
    """numpy array in page-locked host memory (lpt_host_alloc): a read-back destination the GPU writes with one DMA.
    A Device must exist."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    block = _PinnedBlock(max(n, 1))
    buf = (C.c_char * max(n, 1)).from_address(block.ptr)
    buf._lpt_owner = block   # keeps the block alive as long as the ctypes buffer (and any array over it) lives
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class loaders:
    """crates/lib/src/loaders/mod.rs"""

    @staticmethod
    def load_gltf(data, scene):
        buf = np.frombuffer(bytes(data), np.uint8)
        _check(A.lib().lpt_load_gltf(scene._h, A.ptr(buf), buf.size))

    @staticmethod
    def load_gltf_path(path, scene):
        _check(A.lib().lpt_load_gltf_path(scene._h, str(path).encode()))


def load_env(data):
    """the decoding half of ApplicationContext::load_env (crates/standalone/src/app.rs:138-155): Radiance .hdr bytes ->
    (h, w, 4) uint8 RGBE pixels, ready for `ProbeGPU(device, pixels, w, h)`"""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_uint32(), C.c_uint32()
    _check(A.lib().lpt_decode_hdr(A.ptr(buf), buf.size, None, 0, C.byref(w), C.byref(h)))
    out = np.zeros((h.value, w.value, 4), np.uint8)
    _check(A.lib().lpt_decode_hdr(A.ptr(buf), buf.size, A.ptr(out), out.size, C.byref(w), C.byref(h)))
    return out


def decode_image(data):
    """PNG / JPEG bytes -> (h, w, 4) uint8 RGBA (lpt_decode_image): the decoding half of load_blue_noise (app.rs:116-132)"""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_uint32(), C.c_uint32()
    _check(A.lib().lpt_decode_image(A.ptr(buf), buf.size, None, 0, C.byref(w), C.byref(h)))
    out = np.zeros((h.value, w.value, 4), np.uint8)
    _check(A.lib().lpt_decode_image(A.ptr(buf), buf.size, A.ptr(out), out.size, C.byref(w), C.byref(h)))
    return out


def load_blue_noise(renderer, path):
    """ApplicationContext::load_blue_noise (crates/standalone/src/app.rs:116-132): image file -> Renderer::upload_noise_texture"""
    with open(path, "rb") as f:
        px = decode_image(f.read())
    renderer.upload_noise_texture(px, px.shape[1], px.shape[0], px.shape[1] * 4)
    return px


def load_env_path(path):
    with open(path, "rb") as f:
        return load_env(f.read())


def save_screenshot(renderer, path):
    """ApplicationContext::save_screenshot (crates/standalone/src/app.rs:172-187): read_pixels -> PNG file"""
    px = renderer.read_pixels()
    _check(A.lib().lpt_write_png(str(path).encode(), A.ptr(px), px.shape[1], px.shape[0], px.shape[1] * 4))


def save_radiance(renderer_or_array, path):
    """linear radiance (Renderer.read_radiance() or an (h, w, 4) float32 array) -> Radiance RGBE .hdr file"""
    a = renderer_or_array.read_radiance() if hasattr(renderer_or_array, "read_radiance") else renderer_or_array
    a = np.ascontiguousarray(a, np.float32)
    _check(A.lib().lpt_write_hdr(str(path).encode(), A.ptr(a), a.shape[1], a.shape[0], a.shape[1] * 4))


class CameraController:
    """Convention-only mirror of crates/standalone/src/camera.rs:46-116: `update()` returns the
    camera-to-world Mat4 = T(origin) * [right up direction W] (column-major, 16 floats)."""

    def __init__(self, origin=(0.0, 0.0, 0.0), direction=(0.0, 0.0, -1.0)):
        self.origin = np.asarray(origin, np.float32)
        self.direction = np.asarray(direction, np.float32)

    @classmethod
    def from_origin_dir(cls, origin, direction):
        return cls(origin, direction)

    def update(self, delta=0.0):
        d = self.direction / np.float32(np.sqrt(np.dot(self.direction, self.direction)))
        right = np.cross(d, np.array([0, 1, 0], np.float32)).astype(np.float32)
        right /= np.float32(np.sqrt(np.dot(right, right)))
        up = np.cross(right, d).astype(np.float32)
        up /= np.float32(np.sqrt(np.dot(up, up)))
        m = np.zeros((4, 4), np.float32)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, d, self.origin
        m[3, 3] = 1.0
        return m.T.reshape(16).copy()

    def is_static(self):
        return True
