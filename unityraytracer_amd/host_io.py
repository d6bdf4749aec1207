"""Host-side image I/O through the C ABI (csrc/host_io.cpp): Radiance .hdr sky loader and frame writers
(.pfm / .png, f4).  No GPU needed."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import UrtError


def _check(lib, rc):
    if rc != 0:
        raise UrtError(rc, lib.urt_host_io_last_error().decode())


def load_hdr(path: str) -> np.ndarray:
    """Radiance RGBE file -> (H, W, 4) float32, row 0 = bottom (ready for RenderTexture.SetPixels as the sky)."""
    lib = _lib.load()
    w, h = C.c_int(), C.c_int()
    _check(lib, lib.urt_host_load_hdr(path.encode(), C.byref(w), C.byref(h), None, 0))
    out = np.zeros((h.value, w.value, 4), dtype=np.float32)
    _check(lib, lib.urt_host_load_hdr(path.encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p), out.size))
    return out


def write_pfm(path: str, rgba: np.ndarray):
    lib = _lib.load()
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    _check(lib, lib.urt_host_write_pfm(path.encode(), a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0]))


def write_png(path: str, rgba: np.ndarray):
    """Linear RGBA32F -> 8-bit sRGB PNG (the display transfer of the reference's linear-colour-space project)."""
    lib = _lib.load()
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    _check(lib, lib.urt_host_write_png(path.encode(), a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0]))


def encode_srgb8(rgba: np.ndarray) -> np.ndarray:
    """Linear RGBA32F -> RGBA8, colour through the sRGB transfer function, alpha UNORM8: write_png's pixel encoding on its own
    (include/urt.h urt_host_encode_srgb8) — the bytes RenderTexture.ReadBegin("RGBA8_SRGB") delivers."""
    lib = _lib.load()
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.uint8)
    _check(lib, lib.urt_host_encode_srgb8(a.ctypes.data_as(C.c_void_p), a.size // 4, out.ctypes.data_as(C.c_void_p)))
    return out


def srgb8_first_floats() -> np.ndarray:
    """[k] = the smallest float32 whose sRGB code is >= k ([0] = -inf): the step table the GPU encoder searches."""
    lib = _lib.load()
    out = np.empty(256, dtype=np.float32)
    _check(lib, lib.urt_host_srgb8_first_floats(out.ctypes.data_as(C.c_void_p)))
    return out


def resize(rgba: np.ndarray, new_width: int, new_height: int) -> np.ndarray:
    """Separable Mitchell-Netravali resize of an (H, W, 4) float32 image (include/urt.h urt_host_resize_rgba)."""
    lib = _lib.load()
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.zeros((new_height, new_width, 4), dtype=np.float32)
    _check(lib, lib.urt_host_resize_rgba(a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0], out.ctypes.data_as(C.c_void_p), new_width, new_height))
    return out


def load_sky(path: str, max_texture_size: int = 2048) -> np.ndarray:
    """A .hdr sky as the reference's importer settings would deliver it to `_SkyboxTexture` (RM:776): loaded, then downscaled so
    that neither side exceeds `maxTextureSize` (Assets/Skyboxes/*.hdr.meta:36), aspect kept.  (BC6H compression not reproduced.)"""
    img = load_hdr(path)
    h, w = img.shape[:2]
    big = max(w, h)
    if big <= max_texture_size:
        return img
    s = max_texture_size / big
    return resize(img, max(1, int(round(w * s))), max(1, int(round(h * s))))
