//! Safe layer over `libloupiote_hip.so` with the names and the call shapes of the reference's `crates/lib`
//! (`Device`, `Scene`, `SceneGPU`, `ProbeGPU`, `Renderer`, `BlitMode`, `Error`, `loaders::load_gltf`;
//! reference `crates/lib/src/lib.rs:1-11`).  The wgpu parameters of the reference's signatures (`&wgpu::Device`, `&wgpu::Queue`,
//! `&mut wgpu::CommandEncoder`) have no meaning on a compute-only part and are dropped here; INTEGRATION.md §3 shows the
//! adapters that keep them for `crates/standalone`.  NOT COMPILED in the development image (no Rust toolchain — run `cargo check` where one
//! exists before relying on it): the `ffi` module is generated from `include/lpt.h` (`tools/gen_rust_ffi.py`), this file is written against it by hand.
//!
//! Ownership.  The C library keeps the pointers it is handed (`lpt_renderer_set_resources`, `lpt_renderer_resize`, `lpt_renderer_set_comm`), so the
//! handles here are reference-counted (`Rc`) and whoever uses one holds a clone: a `Renderer` keeps its `Device`, its `SceneGPU`, its `ProbeGPU` and
//! its `Comm` alive, a `SceneGPU` / `ProbeGPU` / `Comm` keeps its `Device` alive — what the reference gets from wgpu's reference-counted bind groups
//! (`renderer.rs:687-725`).  Dropping a scene before its renderer is therefore safe in safe code.  (The C side is defensive as well: destroying a
//! scene, probe or communicator detaches the renderers still bound to it.)
//! Threads.  The handles are raw pointers behind `Rc`: neither `Send` nor `Sync`.  One thread drives a device and everything created from it — the
//! reference drives everything from the winit event-loop thread (`crates/standalone/src/app.rs:259-344`).
pub mod ffi;

use std::ffi::{CStr, CString};
use std::ptr;
use std::rc::Rc;

/// reference `crates/lib/src/errors.rs:2-6` (+ the ABI's own codes)
#[derive(Debug)]
pub enum Error {
    FileNotFound(String),
    TextureToBufferReadFail,
    AccelBuild(String),
    Hip(String),
    Rccl(String),
    InvalidArg(String),
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::lpt_last_error()).to_string_lossy().into_owned() }
}

fn check(status: i32) -> Result<(), Error> {
    match status {
        ffi::LPT_OK => Ok(()),
        ffi::LPT_ERR_FILE_NOT_FOUND => Err(Error::FileNotFound(last_error())),
        ffi::LPT_ERR_READBACK => Err(Error::TextureToBufferReadFail),
        ffi::LPT_ERR_ACCEL_BUILD => Err(Error::AccelBuild(last_error())),
        ffi::LPT_ERR_HIP => Err(Error::Hip(last_error())),
        ffi::LPT_ERR_RCCL => Err(Error::Rccl(last_error())),
        _ => Err(Error::InvalidArg(last_error())),
    }
}

/// reference `crates/lib/src/device.rs:80` `Device::new`: one GPU.  A cheap handle: clones share the device, the last one destroys it
#[derive(Clone)]
pub struct Device { inner: Rc<DeviceInner> }
struct DeviceInner { h: *mut ffi::lpt_device }
impl Drop for DeviceInner { fn drop(&mut self) { unsafe { ffi::lpt_device_destroy(self.h); } } }
impl Device {
    pub fn new(hip_ordinal: i32) -> Result<Self, Error> {
        let mut h = ptr::null_mut();
        check(unsafe { ffi::lpt_device_create(hip_ordinal, &mut h) })?;
        Ok(Self { inner: Rc::new(DeviceInner { h }) })
    }
    fn h(&self) -> *mut ffi::lpt_device { self.inner.h }
    pub fn synchronize(&self) -> Result<(), Error> { check(unsafe { ffi::lpt_device_synchronize(self.h()) }) }
}

/// reference `crates/lib/src/scene.rs:30-54` `Scene` (`Scene::default()` seeds one dummy element per array)
pub struct Scene { h: *mut ffi::lpt_scene }
impl Default for Scene {
    fn default() -> Self {
        let mut h = ptr::null_mut();
        check(unsafe { ffi::lpt_scene_create(&mut h) }).expect("lpt_scene_create");
        Self { h }
    }
}
impl Scene {
    /// `BLASArray::add_bvh_indexed` (reference `loaders/gltf.rs:97-105`): positions are `[f32; 4]`, normals `[f32; 3]`, uvs `[f32; 2]`
    pub fn add_mesh(&mut self, positions: &[[f32; 4]], normals: Option<&[[f32; 3]]>, uvs: Option<&[[f32; 2]]>, indices: Option<&[u32]>) -> Result<u32, Error> {
        let mut out = 0u32;
        let n = positions.len() as u32;
        check(unsafe {
            ffi::lpt_scene_add_mesh(self.h, positions.as_ptr() as *const _, 16,
                                    normals.map_or(ptr::null(), |x| x.as_ptr() as *const _), 12,
                                    uvs.map_or(ptr::null(), |x| x.as_ptr() as *const _), 8, n,
                                    indices.map_or(ptr::null(), |x| x.as_ptr()), indices.map_or(0, |x| x.len() as u32), &mut out)
        })?;
        Ok(out)
    }
    /// `add_instance(blas, transform, material)` (reference `loaders/gltf.rs:141-145`); column-major `glam::Mat4::to_cols_array()`
    pub fn add_instance(&mut self, blas_index: u32, model_to_world: &[f32; 16], material_index: u32) -> Result<u32, Error> {
        let mut out = 0u32;
        check(unsafe { ffi::lpt_scene_add_instance(self.h, blas_index, model_to_world.as_ptr(), material_index, &mut out) })?;
        Ok(out)
    }
    /// `Instance::set_transform` (reference `crates/standalone/src/lib.rs:118-121`)
    pub fn set_instance_transform(&mut self, index: u32, model_to_world: &[f32; 16]) -> Result<(), Error> {
        check(unsafe { ffi::lpt_scene_set_instance_transform(self.h, index, model_to_world.as_ptr()) })
    }
    pub fn add_material(&mut self, material: &ffi::lpt_material) -> Result<u32, Error> {
        let mut out = 0u32;
        check(unsafe { ffi::lpt_scene_add_material(self.h, material, &mut out) })?;
        Ok(out)
    }
    pub fn add_image(&mut self, rgba8: &[u8], width: u32, height: u32) -> Result<u32, Error> {
        assert!(rgba8.len() >= (width as usize) * (height as usize) * 4);
        let mut out = 0u32;
        check(unsafe { ffi::lpt_scene_add_image(self.h, rgba8.as_ptr(), width, height, &mut out) })?;
        Ok(out)
    }
}
impl Drop for Scene { fn drop(&mut self) { unsafe { ffi::lpt_scene_destroy(self.h); } } }

pub mod loaders {
    use super::{check, ffi, Error, Scene};
    /// reference `crates/lib/src/loaders/gltf.rs:46`: appends to `scene`
    pub fn load_gltf(data: &[u8], scene: &mut Scene) -> Result<(), Error> {
        check(unsafe { ffi::lpt_load_gltf(scene.h, data.as_ptr(), data.len()) })
    }
}

/// reference `crates/lib/src/scene.rs:151` `SceneGPU::new_from_scene` (the CPU scene stays with the caller).  Clones share the device-side scene;
/// a `Renderer` holds one for as long as it is bound to it
#[derive(Clone)]
pub struct SceneGPU { inner: Rc<SceneGpuInner> }
struct SceneGpuInner { h: *mut ffi::lpt_scene_gpu, _device: Device }
impl Drop for SceneGpuInner { fn drop(&mut self) { unsafe { ffi::lpt_scene_gpu_destroy(self.h); } } }
impl SceneGPU {
    pub fn new_from_scene(scene: &Scene, device: &Device) -> Result<Self, Error> {
        let mut h = ptr::null_mut();
        check(unsafe { ffi::lpt_scene_upload(device.h(), scene.h, &mut h) })?;
        Ok(Self { inner: Rc::new(SceneGpuInner { h, _device: device.clone() }) })
    }
    fn h(&self) -> *mut ffi::lpt_scene_gpu { self.inner.h }
    /// after `Scene::set_instance_transform`: re-bake the moved instances and refit the tree on the GPU (the library submits the frames
    /// recorded against the old poses first)
    pub fn update_instances(&self, scene: &Scene) -> Result<u32, Error> {
        let mut n = 0u32;
        check(unsafe { ffi::lpt_scene_gpu_update_instances(self.h(), scene.h, &mut n) })?;
        Ok(n)
    }
}

/// reference `crates/lib/src/scene.rs:72` `ProbeGPU::new`: RGBE8, equirectangular
#[derive(Clone)]
pub struct ProbeGPU { inner: Rc<ProbeInner> }
struct ProbeInner { h: *mut ffi::lpt_probe, _device: Device }
impl Drop for ProbeInner { fn drop(&mut self) { unsafe { ffi::lpt_probe_destroy(self.h); } } }
impl ProbeGPU {
    pub fn new(device: &Device, rgbe8: &[u8], width: u32, height: u32) -> Result<Self, Error> {
        assert!(rgbe8.len() >= (width as usize) * (height as usize) * 4);
        let mut h = ptr::null_mut();
        check(unsafe { ffi::lpt_probe_upload(device.h(), rgbe8.as_ptr(), width, height, &mut h) })?;
        Ok(Self { inner: Rc::new(ProbeInner { h, _device: device.clone() }) })
    }
    fn h(&self) -> *mut ffi::lpt_probe { self.inner.h }
}

/// reference `crates/lib/src/renderer.rs:160-167` (the spelling `Pahtrace` is the reference's)
#[derive(Copy, Clone, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum BlitMode { Pahtrace = 0, DenoisedPathrace = 1, Temporal = 2, GBuffer = 3, MotionVector = 4 }

/// reference `crates/lib/src/renderer.rs:169-811`
pub struct Renderer {
    h: *mut ffi::lpt_renderer,
    /// pub field of the reference (`renderer.rs:203`); applied by `resize`
    pub downsample_factor: f32,
    /// pub field of the reference (`renderer.rs:204`); handed to the library by `raytrace`
    pub accumulate: bool,
    size: (u32, u32),
    // what the C renderer points at: kept alive here (declared after `h`'s users; `Drop for Renderer` destroys the C renderer first)
    _device: Device,
    scene: Option<SceneGPU>,
    probe: Option<ProbeGPU>,
    comm: Option<Comm>,
}
impl Renderer {
    /// `renderer.rs:209`
    pub fn max_ssbo_element_in_bytes() -> u32 { unsafe { ffi::lpt_max_per_pixel_bytes() } }
    /// `renderer.rs:220` (the swapchain format has no meaning here)
    pub fn new(device: &Device, original_size: (u32, u32)) -> Result<Self, Error> {
        let mut h = ptr::null_mut();
        check(unsafe { ffi::lpt_renderer_create(device.h(), original_size.0, original_size.1, &mut h) })?;
        let mut r = Self { h, downsample_factor: 0.5, accumulate: false, size: (0, 0), _device: device.clone(), scene: None, probe: None, comm: None };
        unsafe { ffi::lpt_renderer_get_size(r.h, &mut r.size.0, &mut r.size.1); }
        Ok(r)
    }
    /// `renderer.rs:326`
    pub fn resize(&mut self, scene: &SceneGPU, probe: Option<&ProbeGPU>, size: (u32, u32)) -> Result<(), Error> {
        check(unsafe { ffi::lpt_renderer_set_downsample(self.h, self.downsample_factor) })?;
        check(unsafe { ffi::lpt_renderer_resize(self.h, scene.h(), probe.map_or(ptr::null(), |p| p.h() as *const _), size.0, size.1) })?;
        self.scene = Some(scene.clone());
        self.probe = probe.cloned();
        check(unsafe { ffi::lpt_renderer_get_size(self.h, &mut self.size.0, &mut self.size.1) })
    }
    /// `renderer.rs:687`
    pub fn set_resources(&mut self, scene: &SceneGPU, probe: Option<&ProbeGPU>) -> Result<(), Error> {
        check(unsafe { ffi::lpt_renderer_set_resources(self.h, scene.h(), probe.map_or(ptr::null(), |p| p.h() as *const _)) })?;
        self.scene = Some(scene.clone());   // the C renderer keeps these pointers: so does this one
        self.probe = probe.cloned();
        Ok(())
    }
    /// `renderer.rs:392`: RECORDS one sample per pixel from `view_transform` (camera-to-world, column-major); launched by the next
    /// submission point (`submit`, any read) — INTEGRATION.md §3a
    pub fn raytrace(&mut self, view_transform: &[f32; 16]) -> Result<(), Error> {
        check(unsafe { ffi::lpt_renderer_set_accumulate(self.h, self.accumulate as i32) })?;
        check(unsafe { ffi::lpt_renderer_raytrace(self.h, view_transform.as_ptr()) })
    }
    /// the app's `queue.submit(Some(encoder.finish()))` (`crates/standalone/src/app.rs:335-337`)
    pub fn submit(&mut self) -> Result<(), Error> { check(unsafe { ffi::lpt_renderer_submit(self.h) }) }
    /// `renderer.rs:609`
    pub fn reset_accumulation(&mut self) -> Result<(), Error> {
        self.accumulate = false;
        check(unsafe { ffi::lpt_renderer_reset_accumulation(self.h) })
    }
    /// `renderer.rs:620`
    pub fn upload_noise_texture(&mut self, data: &[u8], width: u32, height: u32, bytes_per_row: u32) -> Result<(), Error> {
        assert!(data.len() >= (bytes_per_row as usize) * (height as usize));
        check(unsafe { ffi::lpt_renderer_upload_noise(self.h, data.as_ptr(), width, height, bytes_per_row) })
    }
    /// `renderer.rs:666`
    pub fn use_noise_texture(&mut self, flag: bool) -> Result<(), Error> { check(unsafe { ffi::lpt_renderer_use_noise(self.h, flag as i32) }) }
    /// `renderer.rs:675`
    pub fn set_blit_mode(&mut self, mode: BlitMode) -> Result<(), Error> { check(unsafe { ffi::lpt_renderer_set_blit_mode(self.h, mode as i32) }) }
    /// `renderer.rs:683`
    pub fn get_size(&self) -> &(u32, u32) { &self.size }
    /// `renderer.rs:727`: tight rows of sRGB RGBA8; blocking (the reference awaits `device.poll(Wait)`, `:791`)
    pub fn read_pixels(&mut self) -> Result<Vec<u8>, Error> {
        let mut out = vec![0u8; (self.size.0 as usize) * (self.size.1 as usize) * 4];
        check(unsafe { ffi::lpt_renderer_read_pixels(self.h, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// new: the mean radiance as float RGBA (the parity surface)
    pub fn read_radiance(&mut self) -> Result<Vec<f32>, Error> {
        let mut out = vec![0f32; (self.size.0 as usize) * (self.size.1 as usize) * 4];
        check(unsafe { ffi::lpt_renderer_read_radiance(self.h, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// new (multi-GPU, host-side gather): this rank's OWNED pixels of the mean radiance straight into `frame`, a whole-frame buffer in
    /// page-locked host memory (`HostFrame`, or memory passed to `ffi::lpt_host_register` — a shared-memory segment every rank maps);
    /// the other ranks' pixels are left alone.  INTEGRATION.md §9.
    pub fn read_radiance_owned(&mut self, frame: &mut HostFrame) -> Result<(), Error> {
        assert!(frame.len >= (self.size.0 as usize) * (self.size.1 as usize) * 4);
        check(unsafe { ffi::lpt_renderer_read_radiance_owned(self.h, frame.ptr) })
    }
    /// the same into the node's shared frame (`SharedFrame`): every rank writes its own pixels, `SharedFrame::barrier` completes the frame
    pub fn read_radiance_owned_shared(&mut self, frame: &SharedFrame) -> Result<(), Error> {
        assert!(frame.len >= (self.size.0 as usize) * (self.size.1 as usize) * 4);
        check(unsafe { ffi::lpt_renderer_read_radiance_owned(self.h, frame.ptr) })
    }
    /// `renderer.rs:551` without a swapchain: sRGB RGBA8 into the caller's rows
    pub fn blit_rgba8(&mut self, dst: &mut [u8], row_bytes: usize) -> Result<(), Error> {
        assert!(row_bytes >= (self.size.0 as usize) * 4 && dst.len() >= row_bytes * (self.size.1 as usize));
        check(unsafe { ffi::lpt_renderer_blit_rgba8(self.h, dst.as_mut_ptr(), row_bytes) })
    }
    /// build-only knobs (the reference's constants: 3 bounces, `renderer.rs:398-399`)
    pub fn set_max_bounces(&mut self, bounces: u32) -> Result<(), Error> { check(unsafe { ffi::lpt_renderer_set_max_bounces(self.h, bounces) }) }
    pub fn set_seed(&mut self, seed: u32) -> Result<(), Error> { check(unsafe { ffi::lpt_renderer_set_seed(self.h, seed) }) }
    /// launch tuning (the `LPT_OPT_` constants of `ffi`): never changes a frame
    pub fn set_option(&mut self, option: i32, value: u64) -> Result<(), Error> { check(unsafe { ffi::lpt_renderer_set_option(self.h, option, value) }) }
    /// tile-sharded frames: bind a communicator (implies `set_shard(rank, world, 32, 8)`), then `exchange` after the frame's `raytrace` calls
    pub fn set_comm(&mut self, comm: &Comm) -> Result<(), Error> {
        check(unsafe { ffi::lpt_renderer_set_comm(self.h, comm.h()) })?;
        self.comm = Some(comm.clone());
        Ok(())
    }
    pub fn exchange(&mut self, mode: i32) -> Result<(), Error> { check(unsafe { ffi::lpt_renderer_exchange(self.h, mode) }) }
}
impl Drop for Renderer { fn drop(&mut self) { unsafe { ffi::lpt_renderer_destroy(self.h); } } }

/// page-locked host memory (`lpt_host_alloc`): a read-back destination the GPU writes by DMA or by zero-copy stores
pub struct HostFrame { ptr: *mut f32, len: usize }
impl HostFrame {
    pub fn new(floats: usize) -> Result<Self, Error> {
        let mut p: *mut std::os::raw::c_void = ptr::null_mut();
        check(unsafe { ffi::lpt_host_alloc(floats * 4, &mut p) })?;
        Ok(Self { ptr: p as *mut f32, len: floats })
    }
    pub fn as_slice(&self) -> &[f32] { unsafe { std::slice::from_raw_parts(self.ptr, self.len) } }
}
impl Drop for HostFrame { fn drop(&mut self) { unsafe { ffi::lpt_host_free(self.ptr as *mut _); } } }

/// the host-side gather of a tile-sharded frame (`lpt_host_frame_*`): ONE whole frame in POSIX shared memory that every rank
/// of a node maps and page-locks.  Per frame: `renderer.read_radiance_owned_shared(&frame)` on every rank, then `frame.barrier(rank, frame_no)`.
pub struct SharedFrame { h: *mut ffi::lpt_host_frame, ptr: *mut f32, len: usize }
impl SharedFrame {
    fn open(name: &str, size: (u32, u32), world: u32, host_only: bool, create: bool) -> Result<Self, Error> {
        let c = CString::new(name).map_err(|_| Error::InvalidArg("name contains a NUL".into()))?;
        let flags = if host_only { ffi::LPT_HOST_FRAME_HOST_ONLY as u32 } else { 0 };
        let mut h = ptr::null_mut();
        check(unsafe {
            if create { ffi::lpt_host_frame_create(c.as_ptr(), size.0, size.1, world, flags, &mut h) } else { ffi::lpt_host_frame_attach(c.as_ptr(), size.0, size.1, world, flags, &mut h) }
        })?;
        let mut p: *mut f32 = ptr::null_mut();
        check(unsafe { ffi::lpt_host_frame_ptr(h, &mut p) })?;
        Ok(Self { h, ptr: p, len: (size.0 as usize) * (size.1 as usize) * 4 })
    }
    /// rank 0; `name` as for `shm_open` ("/something")
    pub fn create(name: &str, size: (u32, u32), world: u32) -> Result<Self, Error> { Self::open(name, size, world, false, true) }
    /// the other ranks, once rank 0 has created it; `host_only`: a participant without a GPU that only reads the finished frame
    pub fn attach(name: &str, size: (u32, u32), world: u32, host_only: bool) -> Result<Self, Error> { Self::open(name, size, world, host_only, false) }
    /// returns when every participant has called it with this frame number (1, 2, 3, ...)
    pub fn barrier(&self, rank: u32, frame_no: u32, timeout_ms: u32) -> Result<(), Error> { check(unsafe { ffi::lpt_host_frame_barrier(self.h, rank, frame_no, timeout_ms) }) }
    /// the frame (complete after `barrier`)
    pub fn as_slice(&self) -> &[f32] { unsafe { std::slice::from_raw_parts(self.ptr, self.len) } }
}
impl Drop for SharedFrame { fn drop(&mut self) { unsafe { ffi::lpt_host_frame_destroy(self.h); } } }

/// one rank of a node-wide frame (new functionality: the reference is single-GPU); RCCL lives inside the library
#[derive(Clone)]
pub struct Comm { inner: Rc<CommInner> }
struct CommInner { h: *mut ffi::lpt_comm, _device: Device }
impl Drop for CommInner { fn drop(&mut self) { unsafe { ffi::lpt_comm_destroy(self.h); } } }
impl Comm {
    fn h(&self) -> *mut ffi::lpt_comm { self.inner.h }
    /// rank 0 creates the id and ships its 128 bytes to the other ranks out of band
    pub fn unique_id() -> Result<[u8; 128], Error> {
        let mut id = [0u8; 128];
        check(unsafe { ffi::lpt_comm_unique_id(id.as_mut_ptr() as *mut _) })?;
        Ok(id)
    }
    pub fn new(device: &Device, unique_id: &[u8; 128], rank: i32, world_size: i32) -> Result<Self, Error> {
        let mut h = ptr::null_mut();
        check(unsafe { ffi::lpt_comm_create(device.h(), unique_id.as_ptr() as *const _, rank, world_size, &mut h) })?;
        Ok(Self { inner: Rc::new(CommInner { h, _device: device.clone() }) })
    }
}
