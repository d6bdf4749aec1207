"""unityraytracer_amd — MI355X-native drop-in for the GPU path of RemyMuj/UnityRayTracer.

The product is `libunityraytracer_amd.so` (hand-written HIP kernels for gfx950 behind the C ABI of
include/urt.h).  The Python modules are the host-side mirror of the reference's C# driver
(RayTraceMaster.cs) used by tests and bench.py, plus deterministic synthetic scenes.
Importing this package does not load the library; creating a `Context` does, and fails loudly when
the library has not been built or no GPU is usable (there is no CPU fallback).
"""
from . import host_io, host_scene, scenes, strips  # noqa: F401
from ._lib import ABI_SYMBOLS, LIB_PATH, UrtError  # noqa: F401
from .ray_trace_debug import RayTraceDebug  # noqa: F401
from .ray_trace_master import RayTraceMaster, RayTraceObject  # noqa: F401
from .unity_api import (ComputeBuffer, ComputeShader, Context, DeviceGroup, Graphics, Material, RenderTexture, Texture2D,  # noqa: F401
                        debug_build_blas)
