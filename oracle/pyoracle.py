"""ctypes wrapper of oracle/liboracle.so — TEST INFRASTRUCTURE (see oracle.cpp header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


class OracleScene(C.Structure):
    _fields_ = [
        ("mesh_objects", C.c_void_p), ("n_mesh_objects", C.c_int32),
        ("vertices", C.c_void_p), ("n_vertices", C.c_int32),
        ("indices", C.c_void_p), ("n_indices", C.c_int32),
        ("normals", C.c_void_p), ("n_normals", C.c_int32),
        ("spheres", C.c_void_p), ("n_spheres", C.c_int32),
        ("mesh_bvh", C.c_void_p), ("n_mesh_bvh", C.c_int32),
        ("sphere_bvh", C.c_void_p), ("n_sphere_bvh", C.c_int32),
        ("sky", C.c_void_p), ("sky_w", C.c_int32), ("sky_h", C.c_int32),
        ("camera_to_world", C.c_float * 16),
        ("camera_inverse_projection", C.c_float * 16),
        ("pixel_offset", C.c_float * 2),
        ("seed", C.c_float),
        ("num_bounces", C.c_int32), ("num_rays", C.c_int32),
        ("width", C.c_int32), ("height", C.c_int32),
        ("blas_nodes", C.c_void_p), ("n_blas_nodes", C.c_int32),
        ("blas_tri_index", C.c_void_p), ("n_blas_tris", C.c_int32),
        ("blas_mesh_root", C.c_void_p),
        ("mesh_cull_ok", C.c_void_p),
    ]


class OracleCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "tlas_nodes", "blas_nodes", "tri_tests", "sphere_tests", "hit_tri", "hit_sphere",
                                          "hit_ground", "hit_sky", "pixels", "max_ray_steps", "rays_over_256_steps")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def load(build: bool = True):
    global _lib
    if _lib is not None:
        return _lib
    if build and (not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "oracle.cpp"))):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(LIB_PATH)
    vp, i = C.c_void_p, C.c_int
    lib.oracle_render.argtypes = [C.POINTER(OracleScene), i, i, i, i, i, i, vp, C.POINTER(OracleCounters)]
    lib.oracle_render.restype = i
    lib.oracle_compute_cull_ok.argtypes = [C.POINTER(OracleScene), vp]
    lib.oracle_compute_cull_ok.restype = None
    lib.oracle_accumulate.argtypes = [vp, vp, i, C.c_float]
    lib.oracle_accumulate.restype = None
    lib.oracle_math_probe.argtypes = [i, vp, vp, vp, vp, i]
    lib.oracle_math_probe.restype = None
    lib.oracle_check_div_const.argtypes = [C.c_float, i]
    lib.oracle_check_div_const.restype = C.c_ulonglong
    lib.oracle_probe_triangle.argtypes = [vp, vp, vp, vp, vp]
    lib.oracle_probe_triangle.restype = i
    lib.oracle_probe_cslab.argtypes = [vp, vp, vp, vp, vp, i]
    lib.oracle_probe_cslab.restype = None
    lib.oracle_probe_aabb.argtypes = [vp, vp]
    lib.oracle_probe_aabb.restype = i
    lib.oracle_probe_trace.argtypes = [C.POINTER(OracleScene), i, vp, vp]
    lib.oracle_probe_trace.restype = None
    lib.oracle_probe_sky.argtypes = [C.POINTER(OracleScene), vp, vp]
    lib.oracle_probe_sky.restype = None
    lib.oracle_build_blas.argtypes = [C.POINTER(OracleScene), vp, i, vp, vp]
    lib.oracle_build_blas.restype = i
    lib.oracle_compute_normals.argtypes = [vp, i, vp, i, vp]
    lib.oracle_compute_normals.restype = None
    lib.oracle_mesh_leaf_bounds.argtypes = [vp, i, vp, vp, vp]
    lib.oracle_mesh_leaf_bounds.restype = None
    lib.oracle_sphere_leaf_bounds.argtypes = [vp, i, vp]
    lib.oracle_sphere_leaf_bounds.restype = None
    lib.oracle_set_literal_division.argtypes = [i]
    lib.oracle_set_literal_division.restype = None
    lib.oracle_hardware_threads.argtypes = []
    lib.oracle_hardware_threads.restype = i
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Oracle:
    """Binds a scenes.Scene (reference-layout numpy arrays) to the C oracle.  Keeps the arrays alive."""

    def __init__(self, scene, blas=None):
        self.lib = load()
        self.scene = scene
        s = OracleScene()
        self._keep = k = {}
        k["mo"] = np.ascontiguousarray(scene.mesh_objects)
        k["v"] = np.ascontiguousarray(scene.vertices, dtype=np.float32)
        k["i"] = np.ascontiguousarray(scene.indices, dtype=np.int32)
        k["n"] = np.ascontiguousarray(scene.normals, dtype=np.float32)
        k["sp"] = np.ascontiguousarray(scene.spheres)
        k["mb"] = np.ascontiguousarray(scene.mesh_bvh)
        k["sb"] = np.ascontiguousarray(scene.sphere_bvh)
        k["sky"] = np.ascontiguousarray(scene.sky, dtype=np.float32)
        s.mesh_objects, s.n_mesh_objects = _ptr(k["mo"]), len(k["mo"])
        s.vertices, s.n_vertices = _ptr(k["v"]), k["v"].size // 3
        s.indices, s.n_indices = _ptr(k["i"]), k["i"].size
        s.normals, s.n_normals = _ptr(k["n"]), k["n"].size // 3
        s.spheres, s.n_spheres = _ptr(k["sp"]), len(k["sp"])
        s.mesh_bvh, s.n_mesh_bvh = _ptr(k["mb"]), len(k["mb"])
        s.sphere_bvh, s.n_sphere_bvh = _ptr(k["sb"]), len(k["sb"])
        s.sky, s.sky_w, s.sky_h = _ptr(k["sky"]), k["sky"].shape[1], k["sky"].shape[0]
        s.camera_to_world[:] = [float(x) for x in scene.camera_to_world]
        s.camera_inverse_projection[:] = [float(x) for x in scene.camera_inverse_projection]
        s.pixel_offset[:] = [float(scene.pixel_offset[0]), float(scene.pixel_offset[1])]
        s.seed = float(scene.seed)
        s.num_bounces, s.num_rays = int(scene.num_bounces), int(scene.num_rays)
        s.width, s.height = int(scene.width), int(scene.height)
        self.s = s
        self.set_cull(True)
        if blas is not None:
            self.set_blas(*blas)

    def set_cull(self, on: bool = True):
        """The product's object-level cull in the BVH-culled mode 1 (include/urt_math.h tlas_cull; on by default, like the library's
        option front_cull): the per-MeshObject flags come from the oracle's restatement of the product's eligibility + verification
        rule.  Mode 0 (literal brute force) never culls."""
        k = self._keep
        if on and len(k["mo"]) >= 1:
            k["cull"] = np.zeros(len(k["mo"]), dtype=np.int32)
            self.lib.oracle_compute_cull_ok(C.byref(self.s), _ptr(k["cull"]))
            self.s.mesh_cull_ok = _ptr(k["cull"])
        else:
            k.pop("cull", None)
            self.s.mesh_cull_ok = None

    def cull_flags(self):
        return None if "cull" not in self._keep else self._keep["cull"].copy()

    def set_frame(self, pixel_offset, seed):
        self.s.pixel_offset[:] = [float(pixel_offset[0]), float(pixel_offset[1])]
        self.s.seed = float(seed)

    def set_blas(self, nodes, tri_index, mesh_root):
        k = self._keep
        k["bn"] = np.ascontiguousarray(nodes, dtype=np.float32)
        k["bt"] = np.ascontiguousarray(tri_index, dtype=np.int32)
        k["br"] = np.ascontiguousarray(mesh_root, dtype=np.int32)
        self.s.blas_nodes, self.s.n_blas_nodes = _ptr(k["bn"]), k["bn"].size // 16
        self.s.blas_tri_index, self.s.n_blas_tris = _ptr(k["bt"]), k["bt"].size
        self.s.blas_mesh_root = _ptr(k["br"])

    def build_own_blas(self):
        """The oracle's independent median-split BVH (oracle_build_blas)."""
        nt = max(1, self.scene.n_triangles)
        nodes = np.zeros((nt, 16), dtype=np.float32)
        tri = np.zeros(nt, dtype=np.int32)
        root = np.zeros(max(1, len(self.scene.mesh_objects)), dtype=np.int32)
        n = self.lib.oracle_build_blas(C.byref(self.s), _ptr(nodes), nt, _ptr(tri), _ptr(root))
        assert n >= 0
        self.set_blas(nodes[:n], tri, root)
        return nodes[:n], tri, root

    def render(self, rect=None, mode: int = 0, threads: int = 1, counters: bool = False):
        """Returns (y1-y0, x1-x0, 4) float32 for rect = (x0, y0, x1, y1) (default: full frame; row 0 = bottom)."""
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, self.s.width, self.s.height)
        out = np.zeros((y1 - y0, x1 - x0, 4), dtype=np.float32)
        c = OracleCounters()
        rc = self.lib.oracle_render(C.byref(self.s), x0, y0, x1, y1, mode, threads, _ptr(out), C.byref(c))
        if rc != 0:
            raise ValueError("oracle_render: bad arguments")
        return (out, c.as_dict()) if counters else out

    def trace(self, origin, direction, mode: int = 0):
        ray = np.array(list(origin) + list(direction), dtype=np.float32)
        out = np.zeros(8, dtype=np.float32)
        self.lib.oracle_probe_trace(C.byref(self.s), mode, _ptr(ray), _ptr(out))
        return {"distance": float(out[0]), "position": out[1:4].copy(), "normal": out[4:7].copy(), "kind": int(out[7])}

    def sky(self, direction):
        d = np.array(direction, dtype=np.float32)
        out = np.zeros(3, dtype=np.float32)
        self.lib.oracle_probe_sky(C.byref(self.s), _ptr(d), _ptr(out))
        return out


def accumulate(target: np.ndarray, converged: np.ndarray, sample: float) -> np.ndarray:
    """AS:9,39-41 / RM:817-818 — returns the updated copy of `converged`."""
    lib = load()
    t = np.ascontiguousarray(target, dtype=np.float32)
    c = np.array(converged, dtype=np.float32, copy=True, order="C")
    lib.oracle_accumulate(_ptr(t), _ptr(c), t.size // 4, float(sample))
    return c


def math_probe(fn: str, a, b=None, c=None) -> np.ndarray:
    lib = load()
    ids = {"sin": 0, "cos": 1, "log2": 2, "exp2": 3, "pow": 4, "acos": 5, "atan2": 6, "frac": 7, "rand": 8}
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float32)
    c = np.ascontiguousarray(c if c is not None else np.zeros_like(a), dtype=np.float32)
    out = np.zeros_like(a)
    lib.oracle_math_probe(ids[fn], _ptr(a), _ptr(b), _ptr(c), _ptr(out), a.size)
    return out


def probe_cslab(boxes, rays, tbest):
    """(c, h)[n, 6] and (t_near, t_far)[n, 2] of the product's triangle-BVH slab test for n boxes (lo3 hi3) x n rays (origin3 direction3)."""
    lib = load()
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    tb = np.ascontiguousarray(tbest, dtype=np.float32).reshape(-1)
    n = len(boxes)
    assert len(rays) == n and len(tb) == n
    ch = np.zeros((n, 6), np.float32)
    tnf = np.zeros((n, 2), np.float32)
    lib.oracle_probe_cslab(_ptr(boxes), _ptr(rays), _ptr(tb), _ptr(ch), _ptr(tnf), n)
    return ch, tnf


def check_div_const(c: float, threads: int = 8) -> int:
    """Bit patterns x (all 2^32) for which urt::f_div_const(x, c, 1/c) differs from the IEEE quotient x / c."""
    return int(load().oracle_check_div_const(float(c), int(threads)))


def probe_triangle(origin, direction, v0, v1, v2):
    lib = load()
    ray = np.array(list(origin) + list(direction), dtype=np.float32)
    vs = [np.array(v, dtype=np.float32) for v in (v0, v1, v2)]
    tuv = np.zeros(3, dtype=np.float32)
    hit = lib.oracle_probe_triangle(_ptr(ray), _ptr(vs[0]), _ptr(vs[1]), _ptr(vs[2]), _ptr(tuv))
    return bool(hit), tuv


def probe_aabb(origin, direction, vmin, vmax, index=-1):
    from unityraytracer_amd.scenes import BVHNODE_DT
    lib = load()
    ray = np.array(list(origin) + list(direction), dtype=np.float32)
    node = np.zeros(1, dtype=BVHNODE_DT)
    node["vmin"], node["vmax"], node["index"] = vmin, vmax, index
    return bool(lib.oracle_probe_aabb(_ptr(ray), _ptr(node)))


def set_literal_division(on: bool):
    """Test-only variant: the object-level slab test (RS:282-283) with its two literal divisions per axis instead of the
    normative one-reciprocal form.  Process-wide; set it back to False afterwards."""
    load().oracle_set_literal_division(1 if on else 0)


def hardware_threads() -> int:
    return load().oracle_hardware_threads()


def compute_normals_literal(vertices, indices) -> np.ndarray:
    """RM:340-368 exactly as written (O(V * I)) — small meshes only."""
    lib = load()
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    ix = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
    out = np.zeros_like(v)
    lib.oracle_compute_normals(_ptr(v), len(v), _ptr(ix), ix.size, _ptr(out))
    return out


def mesh_leaf_bounds_literal(mesh_objects, vertices, indices) -> np.ndarray:
    from unityraytracer_amd.scenes import BVHNODE_DT
    lib = load()
    mo = np.ascontiguousarray(mesh_objects)
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    ix = np.ascontiguousarray(indices, dtype=np.int32)
    out = np.zeros(len(mo), dtype=BVHNODE_DT)
    lib.oracle_mesh_leaf_bounds(_ptr(mo), len(mo), _ptr(v), _ptr(ix), _ptr(out))
    return out


def sphere_leaf_bounds_literal(spheres) -> np.ndarray:
    from unityraytracer_amd.scenes import BVHNODE_DT
    lib = load()
    sp = np.ascontiguousarray(spheres)
    out = np.zeros(len(sp), dtype=BVHNODE_DT)
    lib.oracle_sphere_leaf_bounds(_ptr(sp), len(sp), _ptr(out))
    return out
