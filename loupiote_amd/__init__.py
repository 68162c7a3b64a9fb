"""loupiote_amd — MI355X-native path-tracing core behind Loupiote's Renderer/Scene/Camera API.

The product is ``libloupiote_hip.so`` (C ABI in include/lpt.h, HIP kernels for gfx950);
this package is the thin host-side mirror of the reference's `loupiote-core` crate."""
from ._abi import EXCHANGE_GATHER_TILES, EXCHANGE_REDUCE, INVALID_INDEX, LIB_PATH, LIGHT_BIT  # noqa: F401
from .api import (BlitMode, CameraController, Comm, Device, Error, ProbeGPU, Renderer, Scene, SceneGPU,  # noqa: F401
                  decode_image, default_light, load_blue_noise, load_env, load_env_path, loaders, save_radiance, save_screenshot)
