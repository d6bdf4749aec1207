"""Host-side mirror of Assets/Scripts/RayTraceDebug.cs ("RD") outside the Unity editor: the append-only log with its level
filter (RD:8,25-36) and a text stand-in for the BVH gizmos (RD:92-162), both implemented in C++ behind the C ABI
(csrc/host_debug.cpp).  RayTraceMaster calls it at the reference's call sites (RM:331-335, 731-735, 869-878)."""
from __future__ import annotations

import ctypes as C
import datetime
import os

import numpy as np

from . import _lib
from ._lib import UrtError
from .scenes import BVHNODE_DT


class RayTraceDebug:
    def __init__(self, directory: str = "Debug", logName: str = "log", debugLevel: int = 2):
        self.lib = _lib.load()
        self.directory, self.logName, self.debugLevel = directory, logName, debugLevel     # RD:7-8 ("Debug/" + logName + ".txt", RD:30)
        self.drawSphereTree = self.drawMeshTree = True                                      # RD:9-10
        self.drawRayTrace = False                                                           # RD:11
        self.drawNormals = True                                                             # RD:14
        self.startRay, self.testRay = (0.0, 1.0, -10.0), (0.0, 0.0, 1.0)                    # RD:12-13 (testRay is an offset from startRay, RD:130)
        os.makedirs(directory, exist_ok=True)
        self.Awake()

    @property
    def path(self) -> str:
        return os.path.join(self.directory, self.logName + ".txt")

    def _check(self, rc: int) -> int:
        if rc not in (0, 1):
            raise UrtError(rc, self.lib.urt_host_debug_last_error().decode())
        return rc

    # RD:19-22
    def Awake(self):
        self.Log("================================\nRun: " + str(datetime.datetime.now()) + "\n================================", 0)

    # RD:25-36: 0 = written, 1 = filtered by debugLevel
    def Log(self, text: str, level: int = 2) -> int:
        return self._check(self.lib.urt_host_log(self.path.encode(), self.debugLevel, level, text.encode()))

    # RM:331-335
    def LogSceneCounts(self, n_spheres, n_mesh_objects, n_vertices, n_indices, n_normals) -> int:
        return self._check(self.lib.urt_host_log_scene_counts(self.path.encode(), self.debugLevel, n_spheres, n_mesh_objects, n_vertices, n_indices, n_normals))

    # RM:731-735
    def LogTreeReport(self, n_mesh_objects, mesh_depth, mesh_real_length, n_spheres, sphere_depth, sphere_real_length) -> int:
        return self._check(self.lib.urt_host_log_tree_report(self.path.encode(), self.debugLevel, n_mesh_objects, mesh_depth, mesh_real_length,
                                                             n_spheres, sphere_depth, sphere_real_length))

    # RD:136-162: type 0 = mesh tree, 1 = sphere tree; returns 1 when that tree's toggle is off (as the reference does)
    def DrawBVHTree(self, nodes: np.ndarray, depth: int, type: int) -> int:
        if type == 0:
            if not self.drawMeshTree:
                return 1
            name = "mesh"
        elif type == 1:
            if not self.drawSphereTree:
                return 1
            name = "sphere"
        else:
            return 1
        nd = np.ascontiguousarray(nodes, dtype=BVHNODE_DT)
        out = os.path.join(self.directory, f"{self.logName}_{name}_bvh.txt")
        s = e = None
        if self.drawRayTrace:                                                               # RD:120-133: segment startRay -> startRay + testRay
            s = np.asarray(self.startRay, np.float32)
            e = (s + np.asarray(self.testRay, np.float32)).astype(np.float32)
        n = C.c_int()
        self._check(self.lib.urt_host_dump_bvh(out.encode(), nd.ctypes.data_as(C.c_void_p) if len(nd) else None, len(nd), int(depth),
                                               s.ctypes.data_as(C.c_void_p) if s is not None else None,
                                               e.ctypes.data_as(C.c_void_p) if e is not None else None, C.byref(n)))
        self.last_dump, self.last_dump_lines = out, n.value
        return 0

    # RD:165-183: returns 1 when drawNormals is off (as the reference does); else writes one line per index slot
    def DrawNormals(self, mesh_objects: np.ndarray, vertices: np.ndarray, indices: np.ndarray, normals: np.ndarray) -> int:
        if not self.drawNormals:
            return 1
        mo = np.ascontiguousarray(mesh_objects)
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        ix = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
        nn = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        out = os.path.join(self.directory, f"{self.logName}_normals.txt")
        n = C.c_int()
        self._check(self.lib.urt_host_dump_normals(out.encode(), mo.ctypes.data_as(C.c_void_p) if len(mo) else None, len(mo),
                                                   v.ctypes.data_as(C.c_void_p), len(v), ix.ctypes.data_as(C.c_void_p), len(ix),
                                                   nn.ctypes.data_as(C.c_void_p), len(nn), C.byref(n)))
        self.last_normals_dump, self.last_normals_lines = out, n.value
        return 0
