"""Host-side scene preparation through the C ABI (csrc/host_scene.cpp): the C++ counterparts of what
RayTraceMaster.cs does before it uploads buffers.  No GPU needed."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import UrtError
from .scenes import BVHNODE_DT


def _check(lib, rc):
    if rc != 0:
        raise UrtError(rc, lib.urt_host_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a.size else None


def compute_normals(vertices, indices) -> np.ndarray:
    """RayTraceMaster.ComputeNormals (RM:340-368) in O(V + I)."""
    lib = _lib.load()
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    ix = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
    out = np.zeros_like(v)
    _check(lib, lib.urt_host_compute_normals(_p(v), len(v), _p(ix), ix.size, _p(out)))
    return out


def mesh_leaf_bounds(mesh_objects, vertices, indices, literal: bool = False) -> np.ndarray:
    """SetupBVHLeaves(List<MeshObject>) (RM:405-433); literal=True keeps the reference's quirks."""
    lib = _lib.load()
    mo = np.ascontiguousarray(mesh_objects)
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    ix = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
    out = np.zeros(len(mo), dtype=BVHNODE_DT)
    _check(lib, lib.urt_host_mesh_leaf_bounds(_p(mo), len(mo), _p(v), len(v), _p(ix), ix.size, 1 if literal else 0, _p(out)))
    return out


def sphere_leaf_bounds(spheres, literal: bool = False) -> np.ndarray:
    """SetupBVHLeaves(List<Sphere>) (RM:436-455); literal=True keeps the inverted boxes."""
    lib = _lib.load()
    sp = np.ascontiguousarray(spheres)
    out = np.zeros(len(sp), dtype=BVHNODE_DT)
    _check(lib, lib.urt_host_sphere_leaf_bounds(_p(sp), len(sp), 1 if literal else 0, _p(out)))
    return out


def build_object_bvh(leaves: np.ndarray, pairing: bool = False) -> np.ndarray:
    """Implicit-heap object BVH in the format CreateBVH emits (RM:681-722) from per-object leaf boxes.  pairing=True runs the
    reference's own builder (SetupBVHRankList / PairBVHBounds / JoinBVH, RM:459-678, restated literally) instead of the median split."""
    lib = _lib.load()
    lv = np.ascontiguousarray(leaves, dtype=BVHNODE_DT)
    n = lib.urt_host_object_bvh_length(len(lv))
    out = np.zeros(n, dtype=BVHNODE_DT)
    fn = lib.urt_host_build_object_bvh_pairing if pairing else lib.urt_host_build_object_bvh
    _check(lib, fn(_p(lv), len(lv), _p(out), n))
    return out
