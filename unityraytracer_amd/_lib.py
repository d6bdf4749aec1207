"""ctypes binding of libunityraytracer_amd.so (include/urt.h).

The shared library is the product: hand-written HIP kernels + the C ABI.  It is built in-tree by
`unityraytracer_amd.build.build_library()` (hipcc, gfx950).  There is no fallback of any kind: if the
library is missing this module raises, and if no GPU is usable `Context()` raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("URT_LIB_PATH") or os.path.join(_HERE, "libunityraytracer_amd.so")   # override: A/B builds in experiments

URT_OK = 0
ERROR_NAMES = {1: "INVALID_ARGUMENT", 2: "INVALID_HANDLE", 3: "NO_DEVICE", 4: "HIP", 5: "UNBOUND", 6: "LAYOUT", 7: "OUT_OF_MEMORY", 8: "SCENE", 9: "WATCHDOG"}

# every symbol include/urt.h declares (tests check that the .so exports each of them)
ABI_SYMBOLS = [
    "urt_abi_version", "urt_device_count", "urt_context_create", "urt_context_destroy", "urt_last_error", "urt_context_set_stream",
    "urt_synchronize", "urt_flush", "urt_buffer_create", "urt_buffer_set_data", "urt_buffer_get_info", "urt_buffer_release", "urt_texture_create",
    "urt_texture_create_external", "urt_texture_set_pixels", "urt_texture_get_pixels", "urt_texture_get_info", "urt_texture_read_begin", "urt_texture_read_end", "urt_texture_read_begin_format", "urt_texture_read_end_format", "urt_texture_release",
    "urt_shader_set_buffer", "urt_shader_set_texture", "urt_shader_set_matrix", "urt_shader_set_vector", "urt_shader_set_float",
    "urt_shader_set_int", "urt_shader_dispatch", "urt_shader_dispatch_rows", "urt_blit_add", "urt_blit", "urt_texture_pack_rows",
    "urt_texture_unpack_rows", "urt_texture_unpack_rows_on", "urt_texture_pack_rows_rgb", "urt_texture_unpack_rows_rgb", "urt_set_option", "urt_get_counters", "urt_reset_counters", "urt_debug_build_blas", "urt_debug_get_blas", "urt_debug_blas_cache_stats",
    "urt_debug_scene_info", "urt_debug_launch_info", "urt_debug_read_scene_blas", "urt_debug_serve_stats", "urt_debug_refit_stats", "urt_debug_build_walk_table", "urt_host_compute_normals", "urt_host_mesh_leaf_bounds", "urt_host_sphere_leaf_bounds", "urt_host_object_bvh_length",
    "urt_host_build_object_bvh", "urt_host_build_object_bvh_pairing", "urt_host_last_error", "urt_host_load_hdr", "urt_host_write_pfm", "urt_host_write_png", "urt_host_encode_srgb8", "urt_host_srgb8_first_floats",
    "urt_host_resize_rgba", "urt_host_io_last_error", "urt_host_log", "urt_host_log_scene_counts", "urt_host_log_tree_report", "urt_host_dump_bvh", "urt_host_dump_normals",
    "urt_host_debug_last_error",
    "urt_group_create", "urt_group_destroy", "urt_group_size", "urt_group_context", "urt_group_last_error", "urt_group_buffer_create",
    "urt_group_buffer_set_data", "urt_group_buffer_release", "urt_group_texture_create", "urt_group_texture_set_pixels",
    "urt_group_texture_get_pixels", "urt_group_texture_release", "urt_group_shader_set_buffer", "urt_group_shader_set_texture",
    "urt_group_shader_set_matrix", "urt_group_shader_set_vector", "urt_group_shader_set_float", "urt_group_shader_set_int",
    "urt_group_set_option", "urt_group_shader_dispatch", "urt_group_blit_add", "urt_group_blit", "urt_group_gather", "urt_group_flush",
    "urt_group_synchronize", "urt_group_get_counters", "urt_group_reset_counters",
]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "tlas_nodes", "blas_nodes", "tri_tests", "sphere_tests", "hit_tri", "hit_sphere",
                                          "hit_ground", "hit_sky", "pixels", "dispatches")] + [("trace_ms", C.c_float), ("watchdog_trips", C.c_uint32), ("launches", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class LaunchInfo(C.Structure):
    """urt_launch_info (include/urt.h): the last trace launch of a context."""
    _fields_ = [("kernel", C.c_char * 96)] + [(n, C.c_int) for n in (
        "kernel_mode", "front_mode", "count_stats", "n_blocks", "block_threads", "lds_bytes", "waves_per_cu", "n_frames", "frame_group",
        "xcd_run", "tile_order", "top_nodes", "tlas_stack", "blas_stack", "lds_tables", "slab_frames", "slab_frames_max",
        "slab_out_of_memory", "experiment", "blas_builder", "trace_stream", "slab_base", "overlapped", "overlapped_launches")]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["kernel"] = self.kernel.decode()
        return d


class UrtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"URT_ERR_{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


_lib = None


def load():
    """Load the shared library, declaring every prototype.  Raises ImportError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    lib.urt_abi_version.argtypes, lib.urt_abi_version.restype = [], C.c_int
    if lib.urt_abi_version() < 0 and os.environ.get("URT_ALLOW_EXPERIMENT") != "1":
        # an A/B, probe or diagnostic BUILD (csrc/experiments.h: -DURT_STAMPS, -DURT_PROBE_NOSTORE, ...): never the product.  Only the
        # measurement scripts under scripts/ opt in; tests, bench.py and the smoke must not run on one by accident (URT_LIB_PATH).
        raise ImportError(f"{LIB_PATH} is an experiment build (urt_abi_version() = {lib.urt_abi_version()}): refused. "
                          "Set URT_ALLOW_EXPERIMENT=1 to load it on purpose (scripts/ only).")
    vp, i, f, u64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64
    pi, pf = C.POINTER(C.c_int), C.POINTER(C.c_float)
    protos = {
        "urt_abi_version": ([], i),
        "urt_device_count": ([pi], i),
        "urt_context_create": ([i, C.POINTER(vp)], i),
        "urt_context_destroy": ([vp], i),
        "urt_last_error": ([vp], C.c_char_p),
        "urt_context_set_stream": ([vp, vp], i),
        "urt_synchronize": ([vp], i),
        "urt_flush": ([vp], i),
        "urt_buffer_create": ([vp, i, i, C.POINTER(u64)], i),
        "urt_buffer_set_data": ([vp, u64, vp, i], i),
        "urt_buffer_get_info": ([vp, u64, pi, pi], i),
        "urt_buffer_release": ([vp, u64], i),
        "urt_texture_create": ([vp, i, i, C.POINTER(u64)], i),
        "urt_texture_create_external": ([vp, i, i, vp, C.POINTER(u64)], i),
        "urt_texture_set_pixels": ([vp, u64, vp], i),
        "urt_texture_get_pixels": ([vp, u64, vp], i),
        "urt_texture_get_info": ([vp, u64, pi, pi, C.POINTER(vp)], i),
        "urt_texture_read_begin": ([vp, u64, C.POINTER(u64)], i),
        "urt_texture_read_end": ([vp, u64, C.POINTER(C.POINTER(C.c_float))], i),
        "urt_texture_read_begin_format": ([vp, u64, i, C.POINTER(u64)], i),
        "urt_texture_read_end_format": ([vp, u64, C.POINTER(vp), C.POINTER(C.c_size_t)], i),
        "urt_texture_release": ([vp, u64], i),
        "urt_shader_set_buffer": ([vp, i, C.c_char_p, u64], i),
        "urt_shader_set_texture": ([vp, i, C.c_char_p, u64], i),
        "urt_shader_set_matrix": ([vp, C.c_char_p, vp], i),
        "urt_shader_set_vector": ([vp, C.c_char_p, vp], i),
        "urt_shader_set_float": ([vp, C.c_char_p, f], i),
        "urt_shader_set_int": ([vp, C.c_char_p, i], i),
        "urt_shader_dispatch": ([vp, i, i, i, i], i),
        "urt_shader_dispatch_rows": ([vp, i, i, i, i, i, i], i),
        "urt_blit_add": ([vp, u64, u64, f], i),
        "urt_blit": ([vp, u64, u64], i),
        "urt_texture_pack_rows": ([vp, u64, i, i, vp, C.POINTER(u64)], i),
        "urt_texture_unpack_rows": ([vp, u64, i, i, vp], i),
        "urt_texture_unpack_rows_on": ([vp, u64, i, i, vp, vp], i),
        "urt_texture_pack_rows_rgb": ([vp, u64, i, i, vp, C.POINTER(u64)], i),
        "urt_texture_unpack_rows_rgb": ([vp, u64, i, i, vp, f, vp], i),
        "urt_set_option": ([vp, C.c_char_p, i], i),
        "urt_get_counters": ([vp, C.POINTER(Counters)], i),
        "urt_reset_counters": ([vp], i),
        "urt_debug_build_blas": ([vp, i, vp, i, vp, i, pi, pi, pi], i),
        "urt_debug_get_blas": ([vp, vp, vp, vp], i),
        "urt_debug_blas_cache_stats": ([vp, vp, vp], i),
        "urt_debug_scene_info": ([vp, pi, pi, pi, pf], i),
        "urt_debug_launch_info": ([vp, C.POINTER(LaunchInfo)], i),
        "urt_debug_serve_stats": ([vp, vp], i),
        "urt_debug_refit_stats": ([vp, vp, vp], i),
        "urt_debug_build_walk_table": ([vp, i, i, vp, vp, vp, i, pi], i),
        "urt_host_dump_normals": ([C.c_char_p, vp, i, vp, i, vp, i, vp, i, pi], i),
        "urt_debug_read_scene_blas": ([vp, vp, vp, vp], i),
        "urt_host_compute_normals": ([vp, i, vp, i, vp], i),
        "urt_host_mesh_leaf_bounds": ([vp, i, vp, i, vp, i, i, vp], i),
        "urt_host_sphere_leaf_bounds": ([vp, i, i, vp], i),
        "urt_host_object_bvh_length": ([i], i),
        "urt_host_build_object_bvh": ([vp, i, vp, i], i),
        "urt_host_build_object_bvh_pairing": ([vp, i, vp, i], i),
        "urt_host_last_error": ([], C.c_char_p),
        "urt_host_load_hdr": ([C.c_char_p, pi, pi, vp, C.c_size_t], i),
        "urt_host_write_pfm": ([C.c_char_p, vp, i, i], i),
        "urt_host_write_png": ([C.c_char_p, vp, i, i], i),
        "urt_host_encode_srgb8": ([vp, C.c_size_t, vp], i),
        "urt_host_srgb8_first_floats": ([vp], i),
        "urt_host_resize_rgba": ([vp, i, i, vp, i, i], i),
        "urt_host_io_last_error": ([], C.c_char_p),
        "urt_host_log": ([C.c_char_p, i, i, C.c_char_p], i),
        "urt_host_log_scene_counts": ([C.c_char_p, i, i, i, i, i, i], i),
        "urt_host_log_tree_report": ([C.c_char_p, i, i, i, i, i, i, i], i),
        "urt_host_dump_bvh": ([C.c_char_p, vp, i, i, vp, vp, pi], i),
        "urt_host_debug_last_error": ([], C.c_char_p),
        "urt_group_create": ([pi, i, C.POINTER(vp)], i),
        "urt_group_destroy": ([vp], i),
        "urt_group_size": ([vp], i),
        "urt_group_context": ([vp, i], vp),
        "urt_group_last_error": ([vp], C.c_char_p),
        "urt_group_buffer_create": ([vp, i, i, C.POINTER(u64)], i),
        "urt_group_buffer_set_data": ([vp, u64, vp, i], i),
        "urt_group_buffer_release": ([vp, u64], i),
        "urt_group_texture_create": ([vp, i, i, C.POINTER(u64)], i),
        "urt_group_texture_set_pixels": ([vp, u64, vp], i),
        "urt_group_texture_get_pixels": ([vp, u64, vp], i),
        "urt_group_texture_release": ([vp, u64], i),
        "urt_group_shader_set_buffer": ([vp, i, C.c_char_p, u64], i),
        "urt_group_shader_set_texture": ([vp, i, C.c_char_p, u64], i),
        "urt_group_shader_set_matrix": ([vp, C.c_char_p, vp], i),
        "urt_group_shader_set_vector": ([vp, C.c_char_p, vp], i),
        "urt_group_shader_set_float": ([vp, C.c_char_p, f], i),
        "urt_group_shader_set_int": ([vp, C.c_char_p, i], i),
        "urt_group_set_option": ([vp, C.c_char_p, i], i),
        "urt_group_shader_dispatch": ([vp, i, i, i, i], i),
        "urt_group_blit_add": ([vp, u64, u64, f], i),
        "urt_group_blit": ([vp, u64, u64], i),
        "urt_group_gather": ([vp, u64, u64], i),
        "urt_group_flush": ([vp], i),
        "urt_group_synchronize": ([vp], i),
        "urt_group_get_counters": ([vp, C.POINTER(Counters)], i),
        "urt_group_reset_counters": ([vp], i),
    }
    assert sorted(protos) == sorted(ABI_SYMBOLS)
    for name, (args, res) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if os.environ.get("URT_LIB_PATH"):     # an older build under A/B (scripts/sweep.py --libs): it simply lacks the newer entry points
                continue
            raise
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib
