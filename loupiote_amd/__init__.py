"""loupiote_amd — MI355X-native path-tracing core behind Loupiote's Renderer/Scene/Camera API.

The product is ``libloupiote_hip.so`` (C ABI in include/lpt.h, HIP kernels for gfx950);
this package is the thin host-side mirror of the reference's `loupiote-core` crate."""
import os as _os

# Host-side decision (this package IS the host above the C ABI): one hardware queue per HIP stream for hosts that keep
# several frames in flight.  Read by the HIP runtime when it initialises; an explicit setting of the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._abi import EXCHANGE_GATHER_TILES, EXCHANGE_REDUCE, INVALID_INDEX, LIB_PATH, LIGHT_BIT  # noqa: F401
from .api import (BlitMode, CameraController, Comm, Device, Error, ProbeGPU, Renderer, Scene, SceneGPU,  # noqa: F401
                  decode_image, default_light, HostFrame, host_register, host_unregister, load_blue_noise, load_env, load_env_path, loaders, pinned_array, save_radiance,
                  save_screenshot)
