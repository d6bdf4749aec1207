"""Host-side mirror of the UnityEngine objects RayTraceMaster.cs drives, over the C ABI (include/urt.h).

Same names, argument meaning and failure behaviour as the calls at RayTraceMaster.cs:233-259,
772-845 so that code (and tests) written against the reference's host side read the same here:

    ComputeBuffer(count, stride).SetData(list) / .Release() / .count / .stride        RM:233-252
    ComputeShader.SetMatrix/SetVector/SetFloat/SetInt/SetTexture/SetBuffer/Dispatch   RM:772-810
    RenderTexture(w, h) / .Release() / .width / .height                               RM:824-845
    Material("Hidden/AdditionShader").SetFloat("_Sample", n); Graphics.Blit(...)      RM:813-819

Unity's methods return void and log errors; here a failed call raises UrtError (carrying the C
status and message) — a caller wanting Unity's behaviour catches, logs and continues.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Counters, UrtError


class Context:
    """One per process and GPU (one-process-per-GPU model).  Owns the HIP stream all work is issued on."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        self._h = C.c_void_p()
        rc = self.lib.urt_context_create(int(device), C.byref(self._h))
        if rc != 0:
            raise UrtError(rc, self.lib.urt_last_error(None).decode())
        self.device = device

    def check(self, rc: int):
        if rc != 0:
            raise UrtError(rc, self.lib.urt_last_error(self._h).decode())

    def close(self):
        if self._h:
            self.lib.urt_context_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        self.check(self.lib.urt_synchronize(self._h))

    def flush(self):
        """Submit deferred (batched) frames to the stream without waiting (include/urt.h urt_flush)."""
        self.check(self.lib.urt_flush(self._h))

    def set_stream(self, hip_stream: int | None):
        self.check(self.lib.urt_context_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def set_option(self, name: str, value: int):
        self.check(self.lib.urt_set_option(self._h, name.encode(), int(value)))

    def counters(self) -> dict:
        c = Counters()
        self.check(self.lib.urt_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def blas_cache_stats(self) -> tuple:
        """(MeshObject BVHs reused, built) by this context's scene preparations so far."""
        a, b = C.c_uint64(), C.c_uint64()
        self.check(self.lib.urt_debug_blas_cache_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reset_counters(self):
        self.check(self.lib.urt_reset_counters(self._h))

    def refit_stats(self) -> tuple:
        """(MeshObjects refitted on the GPU, scene preparations done in place) since the context was created."""
        a, b = C.c_uint64(), C.c_uint64()
        self.check(self.lib.urt_debug_refit_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def serve_stats(self) -> dict:
        """kernel_mode 5 with count_stats: the shared traversal service since the last reset_counters()."""
        a = (C.c_ulonglong * 6)()
        self.check(self.lib.urt_debug_serve_stats(self._h, a))
        k = ("visits", "trips", "lane_trips", "claim_rounds", "claimed", "suspended")
        return dict(zip(k, [int(x) for x in a]))

    def scene_info(self) -> dict:
        """Sizes of the current device scene's triangle BVH and the host time its preparation took (prepares it if stale)."""
        nn, nt, md, ms = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        self.check(self.lib.urt_debug_scene_info(self._h, C.byref(nn), C.byref(nt), C.byref(md), C.byref(ms)))
        return {"n_nodes": nn.value, "n_tris": nt.value, "max_depth": md.value, "prepare_ms": ms.value}

    def launch_info(self) -> dict:
        """The last trace launch (deferred frames are submitted first): kernel instantiation by rocprofv3's name, grid, LDS, batching."""
        li = _lib.LaunchInfo()
        self.check(self.lib.urt_debug_launch_info(self._h, C.byref(li)))
        return li.as_dict()

    def read_scene_blas(self, n_meshes: int):
        """(nodes[n,16], tri_index[n_tris], mesh_root[n_meshes]) of the current device scene, read back from the GPU."""
        info = self.scene_info()
        nodes = np.zeros((info["n_nodes"], 16), dtype=np.float32)
        tri = np.zeros(info["n_tris"], dtype=np.int32)
        root = np.zeros(n_meshes, dtype=np.int32)
        self.check(self.lib.urt_debug_read_scene_blas(self._h, nodes.ctypes.data_as(C.c_void_p), tri.ctypes.data_as(C.c_void_p), root.ctypes.data_as(C.c_void_p)))
        return nodes, tri, root, info


class _GroupLib:
    """Maps the per-context entry points onto their urt_group_* counterparts, so that ComputeBuffer / RenderTexture /
    ComputeShader / Graphics / RayTraceMaster drive a DeviceGroup exactly as they drive a Context."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        if not name.startswith("urt_"):
            raise AttributeError(name)
        return getattr(self._lib, "urt_group_" + name[4:])       # AttributeError for calls a group does not offer


class DeviceGroup:
    """urt_group: ONE host thread drives N GPUs (include/urt.h "device groups").  Scene, uniforms, textures and blits are
    replicated on every rank, Dispatch is partitioned into 8-row strips with global pixel ids, `gather` is the one exchange
    per frame (strips -> full image on rank 0).  `devices` may repeat an ordinal (several ranks on one card)."""

    def __init__(self, devices):
        self._raw = _lib.load()
        self.lib = _GroupLib(self._raw)
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._raw.urt_group_create(arr, len(devices), C.byref(self._h))
        if rc != 0:
            raise UrtError(rc, self._raw.urt_group_last_error(None).decode())
        self.devices = list(devices)

    @property
    def size(self) -> int:
        return self._raw.urt_group_size(self._h)

    def check(self, rc: int):
        if rc != 0:
            raise UrtError(rc, self._raw.urt_group_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._raw.urt_group_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        self.check(self._raw.urt_group_synchronize(self._h))

    def flush(self):
        self.check(self._raw.urt_group_flush(self._h))

    def set_option(self, name: str, value: int):
        self.check(self._raw.urt_group_set_option(self._h, name.encode(), int(value)))

    def gather(self, src: "RenderTexture", dst: "RenderTexture"):
        """The ONE exchange per frame: every rank's strips of `src` -> the full image `dst` on rank 0."""
        self.check(self._raw.urt_group_gather(self._h, src.handle, dst.handle))

    def counters(self) -> dict:
        c = Counters()
        self.check(self._raw.urt_group_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        self.check(self._raw.urt_group_reset_counters(self._h))


class ComputeBuffer:
    """UnityEngine.ComputeBuffer (RM:247-250)."""

    def __init__(self, ctx: Context, count: int, stride: int):
        self.ctx = ctx
        h = C.c_uint64()
        ctx.check(ctx.lib.urt_buffer_create(ctx._h, int(count), int(stride), C.byref(h)))
        self.handle = h.value
        self.count, self.stride = int(count), int(stride)

    def SetData(self, data):
        a = np.ascontiguousarray(data)
        if a.nbytes % self.stride:
            raise UrtError(1, f"SetData: {a.nbytes} bytes is not a multiple of stride {self.stride}")
        self.ctx.check(self.ctx.lib.urt_buffer_set_data(self.ctx._h, self.handle, a.ctypes.data_as(C.c_void_p), a.nbytes // self.stride))

    def Release(self):
        if self.handle:
            self.ctx.check(self.ctx.lib.urt_buffer_release(self.ctx._h, self.handle))
            self.handle = 0


class RenderTexture:
    """UnityEngine.RenderTexture(w, h, 0, ARGBFloat, Linear) with enableRandomWrite (RM:834-840).
    Row 0 is the bottom row.  `external_ptr` wraps caller-owned device memory (e.g. a torch tensor)."""

    def __init__(self, ctx: Context, width: int, height: int, external_ptr: int | None = None):
        self.ctx = ctx
        h = C.c_uint64()
        if external_ptr is None:
            ctx.check(ctx.lib.urt_texture_create(ctx._h, int(width), int(height), C.byref(h)))
        else:
            ctx.check(ctx.lib.urt_texture_create_external(ctx._h, int(width), int(height), C.c_void_p(external_ptr), C.byref(h)))
        self.handle = h.value
        self.width, self.height = int(width), int(height)

    def Release(self):
        if self.handle:
            self.ctx.check(self.ctx.lib.urt_texture_release(self.ctx._h, self.handle))
            self.handle = 0

    def SetPixels(self, rgba):
        a = np.ascontiguousarray(rgba, dtype=np.float32)
        if a.size != self.width * self.height * 4:
            raise UrtError(1, "SetPixels: expected height*width*4 floats")
        self.ctx.check(self.ctx.lib.urt_texture_set_pixels(self.ctx._h, self.handle, a.ctypes.data_as(C.c_void_p)))

    def GetPixels(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self.ctx.check(self.ctx.lib.urt_texture_get_pixels(self.ctx._h, self.handle, out.ctypes.data_as(C.c_void_p)))
        return out

    FORMATS = {"RGBA32F": (0, np.float32), "RGBA8_SRGB": (1, np.uint8), "RGBA16F": (2, np.float16)}   # include/urt.h URT_FORMAT_*

    def ReadBegin(self, format: str = "RGBA32F") -> int:
        """Start a pipelined readback of the image as it is now (include/urt.h urt_texture_read_begin_format), converted on the GPU to
        `format` ("RGBA32F", "RGBA8_SRGB": what an 8-bit back buffer holds after RM:819, "RGBA16F"); returns a ticket."""
        t = C.c_uint64()
        self.ctx.check(self.ctx.lib.urt_texture_read_begin_format(self.ctx._h, self.handle, self.FORMATS[format][0], C.byref(t)))
        self._read_formats = getattr(self, "_read_formats", {})
        self._read_formats[t.value] = format
        return t.value

    def ReadEnd(self, ticket: int, copy: bool = True) -> np.ndarray:
        """Wait for that readback; the image as (height, width, 4) of the ticket's format — a copy, or (copy=False) a view of the
        library's pinned buffer that stays valid until the third ReadBegin after the ticket's."""
        dtype = self.FORMATS[getattr(self, "_read_formats", {}).pop(ticket, "RGBA32F")][1]
        p, n = C.c_void_p(), C.c_size_t()
        self.ctx.check(self.ctx.lib.urt_texture_read_end_format(self.ctx._h, C.c_uint64(ticket), C.byref(p), C.byref(n)))
        assert n.value == self.height * self.width * 4 * np.dtype(dtype).itemsize
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).view(dtype).reshape(self.height, self.width, 4)
        return a.copy() if copy else a

    def device_ptr(self) -> int:
        p = C.c_void_p()
        self.ctx.check(self.ctx.lib.urt_texture_get_info(self.ctx._h, self.handle, None, None, C.byref(p)))
        return p.value

    def packed_bytes(self, first_group_row: int, row_stride: int) -> int:
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.urt_texture_pack_rows(self.ctx._h, self.handle, first_group_row, row_stride, None, C.byref(n)))
        return n.value

    def pack_rows(self, first_group_row: int, row_stride: int, device_dst: int, rgb: bool = False):
        """This rank's strips -> a dense device buffer; rgb: three channels per pixel (include/urt.h urt_texture_pack_rows_rgb)."""
        fn = self.ctx.lib.urt_texture_pack_rows_rgb if rgb else self.ctx.lib.urt_texture_pack_rows
        self.ctx.check(fn(self.ctx._h, self.handle, first_group_row, row_stride, C.c_void_p(device_dst), None))

    def unpack_rows_rgb(self, first_group_row: int, row_stride: int, device_src: int, alpha: float, stream: int | None = None):
        """De-interleave RGB strips into this image, writing `alpha` (running_mean_alpha) into the fourth channel."""
        self.ctx.check(self.ctx.lib.urt_texture_unpack_rows_rgb(self.ctx._h, self.handle, first_group_row, row_stride, C.c_void_p(device_src),
                                                                float(alpha), C.c_void_p(stream or 0)))

    def unpack_rows(self, first_group_row: int, row_stride: int, device_src: int, stream: int | None = None):
        """De-interleave packed strips into this image; `stream` = a caller-ordered hipStream_t (include/urt.h
        urt_texture_unpack_rows_on), default = the context's stream."""
        if stream:
            self.ctx.check(self.ctx.lib.urt_texture_unpack_rows_on(self.ctx._h, self.handle, first_group_row, row_stride, C.c_void_p(device_src), C.c_void_p(stream)))
        else:
            self.ctx.check(self.ctx.lib.urt_texture_unpack_rows(self.ctx._h, self.handle, first_group_row, row_stride, C.c_void_p(device_src)))


Texture2D = RenderTexture   # the sky is an ordinary RGBA32F image here


import struct as _struct
_PACK4 = _struct.Struct("4f").pack


class ComputeShader:
    """UnityEngine.ComputeShader for RayTraceShader.compute; kernel 0 is CSMain."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._bound = ctx.__dict__.setdefault("_bound", {})   # what this context last received, per name (shared by every wrapper of the context)

    def FindKernel(self, name: str) -> int:
        if name != "CSMain":
            raise UrtError(1, f"FindKernel: no kernel named {name}")
        return 0

    # The reference re-sets every uniform and binding every frame (RM:772-795).  Setting a name to the value it already has is
    # a no-op at the boundary, so the wrapper remembers what this context last received per name and skips the call: the host
    # loop of a frame drops from ~18 to ~9 us, which is GPU idle time before a batch of deferred frames is submitted.
    def SetMatrix(self, name: str, m16):
        a = np.ascontiguousarray(m16, dtype=np.float32).reshape(16)
        key, val = ("m", name), a.tobytes()
        if self._bound.get(key) == val:
            return
        self.ctx.check(self.ctx.lib.urt_shader_set_matrix(self.ctx._h, name.encode(), a.ctypes.data_as(C.c_void_p)))
        self._bound[key] = val                                 # only what the library accepted

    def SetVector(self, name: str, v):
        # (the per-frame _PixelOffset: packed with struct, not numpy — 3 us -> 0.6 us of the host's 9 us per frame)
        n = len(v)
        val = _PACK4(float(v[0]) if n > 0 else 0.0, float(v[1]) if n > 1 else 0.0, float(v[2]) if n > 2 else 0.0, float(v[3]) if n > 3 else 0.0)
        key = ("v", name)
        if self._bound.get(key) == val:
            return
        self.ctx.check(self.ctx.lib.urt_shader_set_vector(self.ctx._h, name.encode(), val))
        self._bound[key] = val

    def SetFloat(self, name: str, v: float):
        key, val = ("f", name), C.c_float(v).value             # the float32 the library will see
        if self._bound.get(key) == val and val == val:
            return
        self.ctx.check(self.ctx.lib.urt_shader_set_float(self.ctx._h, name.encode(), float(v)))
        self._bound[key] = val

    def SetInt(self, name: str, v: int):
        key, val = ("i", name), int(v)
        if self._bound.get(key) == val:
            return
        self.ctx.check(self.ctx.lib.urt_shader_set_int(self.ctx._h, name.encode(), val))
        self._bound[key] = val
        self._bound["owner"] = None                             # (a RayTraceMaster's "nothing changed" shortcut ends: ray_trace_master.SetShaderParameters)

    def SetTexture(self, kernel: int, name: str, tex: RenderTexture | None):
        key, h = ("t", kernel, name), tex.handle if tex else 0
        if self._bound.get(key) == h:
            return
        self.ctx.check(self.ctx.lib.urt_shader_set_texture(self.ctx._h, kernel, name.encode(), h))
        self._bound[key] = h

    def SetBuffer(self, kernel: int, name: str, buf: ComputeBuffer | None):
        key, h = ("b", kernel, name), buf.handle if buf else 0
        if self._bound.get(key) == h:
            return
        self.ctx.check(self.ctx.lib.urt_shader_set_buffer(self.ctx._h, kernel, name.encode(), h))
        self._bound[key] = h
        self._bound["owner"] = None

    def Dispatch(self, kernel: int, groups_x: int, groups_y: int, groups_z: int):
        self.ctx.check(self.ctx.lib.urt_shader_dispatch(self.ctx._h, kernel, groups_x, groups_y, groups_z))

    def DispatchRows(self, kernel: int, groups_x: int, groups_y: int, groups_z: int, first_group_row: int, row_stride: int):
        """Multi-GPU extension: only the 8-row strips first_group_row, +row_stride, ... (global pixel ids kept)."""
        self.ctx.check(self.ctx.lib.urt_shader_dispatch_rows(self.ctx._h, kernel, groups_x, groups_y, groups_z, first_group_row, row_stride))


class Material:
    """new Material(Shader.Find("Hidden/AdditionShader")) (RM:813-815)."""

    def __init__(self, shader_name: str = "Hidden/AdditionShader"):
        if shader_name != "Hidden/AdditionShader":
            raise UrtError(1, f"Shader.Find: {shader_name} not found")
        self.floats = {"_Sample": 0.0}

    def SetFloat(self, name: str, v: float):
        self.floats[name] = float(v)


class Graphics:
    @staticmethod
    def Blit(source: RenderTexture, dest: RenderTexture, mat: Material | None = None):
        """Graphics.Blit(src, dst[, additionMaterial]) — RM:818-819."""
        ctx = source.ctx
        if mat is None:
            ctx.check(ctx.lib.urt_blit(ctx._h, source.handle, dest.handle))
        else:
            ctx.check(ctx.lib.urt_blit_add(ctx._h, source.handle, dest.handle, mat.floats["_Sample"]))


def debug_build_blas(mesh_objects: np.ndarray, vertices: np.ndarray, indices: np.ndarray):
    """Run the library's triangle-BVH builder on host arrays (no GPU needed) and return
    (nodes[n,16] f32, tri_index[n_tris] i32, mesh_root[n_meshes] i32, mesh_first_tri, max_depth)."""
    lib = _lib.load()
    mo = np.ascontiguousarray(mesh_objects)
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    ix = np.ascontiguousarray(indices, dtype=np.int32)
    nn, nt, md = C.c_int(), C.c_int(), C.c_int()
    rc = lib.urt_debug_build_blas(mo.ctypes.data_as(C.c_void_p), len(mo), v.ctypes.data_as(C.c_void_p), len(v.reshape(-1, 3)),
                                  ix.ctypes.data_as(C.c_void_p), ix.size, C.byref(nn), C.byref(nt), C.byref(md))
    if rc != 0:
        raise UrtError(rc, lib.urt_last_error(None).decode())
    nodes = np.zeros((nn.value, 16), dtype=np.float32)
    tri = np.zeros(nt.value, dtype=np.int32)
    root = np.zeros(len(mo), dtype=np.int32)
    first = np.zeros(len(mo), dtype=np.int32)
    lib.urt_debug_get_blas(nodes.ctypes.data_as(C.c_void_p), tri.ctypes.data_as(C.c_void_p), root.ctypes.data_as(C.c_void_p),
                           first.ctypes.data_as(C.c_void_p))
    return nodes, tri, root, first, md.value
