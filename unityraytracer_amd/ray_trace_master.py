"""Host-side mirror of Assets/Scripts/RayTraceMaster.cs — the frame driver around the hot path.

Method names and order of GPU calls follow the reference (RM = RayTraceMaster.cs):

    OnRenderImage (RM:848-866)  ->  [RebuildTrees (RM:725-746)]  ->  SetShaderParameters (RM:772-795)
                                ->  Render (RM:798-821)  ->  InitRenderTexture (RM:824-845)

The scene arrives already flattened (a `scenes.Scene`: what RebuildObjectLists RM:262-336 and the BVH
builder RM:405-722 produce); `RebuildTrees` here is the upload half of the reference's: the seven
CreateComputeBuffer calls (RM:738-745).  `UnityEngine.Random.value` (RM:777-778) is replaced by the
documented splitmix64 frame sequence of scenes.frame_uniforms so that frames are reproducible.

Multi-GPU (one process per GPU): construct with rank/world_size; each rank renders the 8-row strips
rank, rank+world, ... with GLOBAL pixel ids and accumulates locally; `gather_converged` moves the
strips to rank 0 with one collective at frame end.
"""
from __future__ import annotations

import math

import numpy as np

from dataclasses import dataclass, field

from . import host_scene, scenes
from .unity_api import ComputeBuffer, ComputeShader, Context, Graphics, Material, RenderTexture


@dataclass
class RayTraceObject:
    """Assets/Scripts/RayTraceObject.cs: what a scene object contributes (RO:9-19).  type 1 = analytic sphere
    (position, radius), anything else = mesh (vertices, triangles of submesh 0, localToWorldMatrix)."""
    type: int = 0
    albedoColor: tuple = (0.0, 0.4, 1.0)          # RO:12
    specularColor: tuple = (0.7, 0.0, 1.0)        # RO:13
    emissionColor: tuple = (0.0, 0.0, 0.0)        # RO:14
    smoothness: float = 0.69                      # RO:15
    position: tuple = (0.0, 0.0, 0.0)             # RO:34 (spheres)
    radius: float = 0.5                           # RO:33 (spheres)
    vertices: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))      # mesh.vertices
    triangles: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.int32))       # mesh.GetIndices(0)
    localToWorldMatrix: np.ndarray = field(default_factory=lambda: scenes.trs())            # transform.localToWorldMatrix


class RayTraceMaster:
    MeshObjectStructSize = 112   # RM:43
    SphereStructSize = 56        # RM:44
    BVHNodeSize = 28             # RM:45

    def __init__(self, ctx: Context, scene: scenes.Scene, rank: int = 0, world_size: int = 1, frame_seed: int = 0x5EED):
        self.ctx = ctx
        self.RayTraceShader = ComputeShader(ctx)
        self.scene = scene
        self.numBounces = scene.num_bounces          # RM:17
        self.numRays = scene.num_rays                # RM:18
        self.rank, self.world_size = rank, world_size
        self.frame_seed = frame_seed
        self._currentSample = 0                      # RM:19
        self._bindings_key = None                    # what SetShaderParameters last bound (its every-frame re-set is skipped while nothing changed)
        self._frame = 0
        self._target = None                          # RM:11
        self._converged = None                       # RM:12
        self._additionMaterial = None                # RM:20
        self._treesNeedRebuilding = True             # RM:24
        self._rayTraceObjects = []                   # RM:22 (empty: the scene arrives pre-flattened in `scene`)
        self.SkyboxTexture = None                    # RM:10
        self.rayDebug = None                         # RM:9: a RayTraceDebug (ray_trace_debug.py) or None
        self._meshObjectBuffer = self._vertexBuffer = self._indexBuffer = self._normalBuffer = None
        self._sphereBuffer = self._meshObjectBVHBuffer = self._sphereBVHBuffer = None
        self.screen_width, self.screen_height = scene.width, scene.height

    # RM:215-230
    def RegisterObject(self, obj: RayTraceObject):
        self._rayTraceObjects.append(obj)
        self._treesNeedRebuilding = True

    def UnregisterObject(self, obj: RayTraceObject):
        self._rayTraceObjects.remove(obj)
        self._treesNeedRebuilding = True

    # RM:262-336 — flatten the registered objects into the lists the buffers are made from; normals (RM:340-368) and the
    # object-level BVHs (RM:405-722 output contract) come from the C++ host library (csrc/host_scene.cpp)
    def RebuildObjectLists(self, literal_leaf_bounds: bool = False, pairing_heap: bool = False):
        s = self.scene
        spheres, mesh_objects, verts, idx = [], [], [], []
        nv = ni = 0
        for obj in self._rayTraceObjects:
            lighting = scenes._params(obj.albedoColor, obj.specularColor, obj.emissionColor, obj.smoothness)
            if obj.type == 1:                                                     # RM:277-293
                sp = np.zeros((), scenes.SPHERE_DT)
                sp["position"], sp["radius"], sp["lighting"] = obj.position, obj.radius, lighting
                spheres.append(sp)
            else:                                                                 # RM:295-321
                v = np.asarray(obj.vertices, np.float32).reshape(-1, 3)
                t = np.asarray(obj.triangles, np.int32).reshape(-1)
                mo = np.zeros((), scenes.MESHOBJECT_DT)
                mo["localToWorldMatrix"], mo["indices_offset"], mo["indices_count"], mo["lighting"] = obj.localToWorldMatrix, ni, len(t), lighting
                verts.append(v)
                idx.append(t + nv)                                                # RM:305: offset by the first vertex
                mesh_objects.append(mo)
                nv += len(v)
                ni += len(t)
        s.spheres = np.array(spheres, scenes.SPHERE_DT) if spheres else np.zeros(0, scenes.SPHERE_DT)
        s.mesh_objects = np.array(mesh_objects, scenes.MESHOBJECT_DT) if mesh_objects else np.zeros(0, scenes.MESHOBJECT_DT)
        s.vertices = np.concatenate(verts) if verts else np.zeros((0, 3), np.float32)
        s.indices = np.concatenate(idx).astype(np.int32) if idx else np.zeros(0, np.int32)
        s.normals = host_scene.compute_normals(s.vertices, s.indices)            # RM:328
        # CreateBVH(_meshObjects) / CreateBVH(_spheres), RM:727-728 (the reference throws on an empty list, A.7; here empty = no buffer)
        s.mesh_bvh = host_scene.build_object_bvh(host_scene.mesh_leaf_bounds(s.mesh_objects, s.vertices, s.indices, literal_leaf_bounds), pairing_heap) \
            if len(s.mesh_objects) else np.zeros(0, scenes.BVHNODE_DT)
        s.sphere_bvh = host_scene.build_object_bvh(host_scene.sphere_leaf_bounds(s.spheres, literal_leaf_bounds), pairing_heap) \
            if len(s.spheres) else np.zeros(0, scenes.BVHNODE_DT)
        if getattr(self, "rayDebug", None) is not None:                             # RM:331-335
            self.rayDebug.LogSceneCounts(len(s.spheres), len(s.mesh_objects), len(s.vertices), len(s.indices), len(s.normals))

    # RM:233-252
    def CreateComputeBuffer(self, buffer, data: np.ndarray, stride: int):
        count = data.nbytes // stride
        if buffer is not None and (count == 0 or buffer.count != count or buffer.stride != stride):
            buffer.Release()
            buffer = None
        if count != 0:
            if buffer is None:
                buffer = ComputeBuffer(self.ctx, count, stride)
            buffer.SetData(data)
        return buffer

    # RM:255-259
    def SetComputeBuffer(self, name: str, buffer):
        if buffer is not None:
            self.RayTraceShader.SetBuffer(0, name, buffer)

    # RM:725-746 (upload half)
    @staticmethod
    def tree_depth(n_objects: int) -> int:
        """MeshDepth / SphereDepth of CreateBVH (RM:683,705): ceil(log2 n) + 1 levels; 0 for an empty list."""
        return 0 if n_objects <= 0 else int(math.ceil(math.log2(n_objects))) + 1 if n_objects > 1 else 1

    def RebuildTrees(self):
        s = self.scene
        if self.rayDebug is not None:                                             # RM:731-735
            self.rayDebug.LogTreeReport(len(s.mesh_objects), self.tree_depth(len(s.mesh_objects)), len(s.mesh_bvh),
                                        len(s.spheres), self.tree_depth(len(s.spheres)), len(s.sphere_bvh))
        self._meshObjectBuffer = self.CreateComputeBuffer(self._meshObjectBuffer, s.mesh_objects, self.MeshObjectStructSize)
        self._vertexBuffer = self.CreateComputeBuffer(self._vertexBuffer, np.ascontiguousarray(s.vertices, np.float32), 12)
        self._indexBuffer = self.CreateComputeBuffer(self._indexBuffer, np.ascontiguousarray(s.indices, np.int32), 4)
        self._normalBuffer = self.CreateComputeBuffer(self._normalBuffer, np.ascontiguousarray(s.normals, np.float32), 12)
        self._sphereBuffer = self.CreateComputeBuffer(self._sphereBuffer, s.spheres, self.SphereStructSize)
        self._meshObjectBVHBuffer = self.CreateComputeBuffer(self._meshObjectBVHBuffer, s.mesh_bvh, self.BVHNodeSize)
        self._sphereBVHBuffer = self.CreateComputeBuffer(self._sphereBVHBuffer, s.sphere_bvh, self.BVHNodeSize)
        if self.SkyboxTexture is None and s.sky is not None:
            h, w = s.sky.shape[:2]
            self.SkyboxTexture = RenderTexture(self.ctx, w, h)
            self.SkyboxTexture.SetPixels(s.sky)

    # RM:772-795
    def SetShaderParameters(self):
        sh, s = self.RayTraceShader, self.scene
        sh.SetMatrix("_CameraToWorld", s.camera_to_world)
        sh.SetMatrix("_CameraInverseProjection", s.camera_inverse_projection)
        sh.SetTexture(0, "_SkyboxTexture", self.SkyboxTexture)
        ox, oy, seed = scenes.frame_uniforms(self._frame, self.frame_seed) if self._frame else (s.pixel_offset[0], s.pixel_offset[1], s.seed)
        sh.SetVector("_PixelOffset", (ox, oy))
        sh.SetFloat("_Seed", seed)
        # RM re-sets the four ints and the seven bindings every frame; setting a name to the value it has is a no-op at the boundary, and
        # when none of them changed since the last frame of this master the eleven calls are skipped as one (host time per frame is GPU
        # idle time before a batch of deferred frames is submitted: 9.4 -> ~5 us)
        key = (self.numBounces, self.numRays, len(s.mesh_bvh), len(s.sphere_bvh), self._meshObjectBuffer, self._vertexBuffer, self._indexBuffer,
               self._normalBuffer, self._sphereBuffer, self._meshObjectBVHBuffer, self._sphereBVHBuffer, id(sh._bound))
        if key == self._bindings_key and sh._bound.get("owner") is self:
            return
        sh.SetInt("_numBounces", self.numBounces)
        sh.SetInt("_numRays", self.numRays)
        sh.SetInt("_MeshBVH_len", len(s.mesh_bvh))
        sh.SetInt("_SphereBVH_len", len(s.sphere_bvh))
        self.SetComputeBuffer("_MeshObjects", self._meshObjectBuffer)
        self.SetComputeBuffer("_Vertices", self._vertexBuffer)
        self.SetComputeBuffer("_Indices", self._indexBuffer)
        self.SetComputeBuffer("_Normals", self._normalBuffer)
        self.SetComputeBuffer("_Spheres", self._sphereBuffer)
        self.SetComputeBuffer("_MeshBVH", self._meshObjectBVHBuffer)
        self.SetComputeBuffer("_SphereBVH", self._sphereBVHBuffer)
        self._bindings_key = key
        sh._bound["owner"] = self                               # another master (or direct Set* calls through another wrapper) on this context ends the shortcut

    # RM:824-845
    def InitRenderTexture(self):
        if self._target is None or self._target.width != self.screen_width or self._target.height != self.screen_height:
            if self._target is not None:
                self._target.Release()
                self._converged.Release()
            self._target = RenderTexture(self.ctx, self.screen_width, self.screen_height)
            self._converged = RenderTexture(self.ctx, self.screen_width, self.screen_height)
            self._currentSample = 0

    # RM:798-821
    def Render(self, destination: RenderTexture | None = None):
        self.InitRenderTexture()
        self.RayTraceShader.SetTexture(0, "Result", self._target)
        threadGroupsX = math.ceil(self.screen_width / 8.0)
        threadGroupsY = math.ceil(self.screen_height / 8.0)
        if self.world_size == 1:
            self.RayTraceShader.Dispatch(0, threadGroupsX, threadGroupsY, 1)
        else:
            self.RayTraceShader.DispatchRows(0, threadGroupsX, threadGroupsY, 1, self.rank, self.world_size)
        if self._additionMaterial is None:
            self._additionMaterial = Material("Hidden/AdditionShader")
        self._additionMaterial.SetFloat("_Sample", self._currentSample)
        Graphics.Blit(self._target, self._converged, self._additionMaterial)
        if destination is not None:
            Graphics.Blit(self._converged, destination)
        self._currentSample += 1
        self._frame += 1

    # RM:848-866
    def OnRenderImage(self, destination: RenderTexture | None = None):
        if self._treesNeedRebuilding:
            self._currentSample = 0
            self._treesNeedRebuilding = False
            if self._rayTraceObjects:
                self.RebuildObjectLists()
            self.RebuildTrees()
        self.SetShaderParameters()
        self.Render(destination)

    # RM:760-769: a camera move resets the running mean
    def ResetAccumulation(self):
        self._currentSample = 0

    # RM:761-763: F12 -> ScreenCapture.CaptureScreenshot("Screenshots/" + Time.time + "-" + _currentSample + ".png")
    def CaptureScreenshot(self, directory: str, time_seconds: float) -> str:
        import os
        from . import host_io
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, f"{time_seconds:g}-{self._currentSample}.png")
        host_io.write_png(path, self._converged.GetPixels())
        return path

    # RM:869-878: the editor gizmos become text dumps of the two object-level heaps (RayTraceDebug.DrawBVHTree) and of the normals (DrawNormals)
    def OnDrawGizmos(self):
        if self.rayDebug is not None:
            s = self.scene
            self.rayDebug.DrawBVHTree(s.mesh_bvh, self.tree_depth(len(s.mesh_objects)), 0)
            mesh_dump = getattr(self.rayDebug, "last_dump", None)
            self.rayDebug.DrawBVHTree(s.sphere_bvh, self.tree_depth(len(s.spheres)), 1)
            if len(s.mesh_objects):
                self.rayDebug.DrawNormals(s.mesh_objects, s.vertices, s.indices, s.normals)      # RM:875-876
            return mesh_dump, getattr(self.rayDebug, "last_dump", None)
        return None, None

    # ---- checkpoint / resume of a progressive accumulation (the reference keeps `_converged` only in GPU memory and loses it
    # on every reset, RM:189,766,843,852; a 1024-spp run such as BASELINE config 5 wants to survive a restart) ----
    def SaveCheckpoint(self, path: str):
        """The running mean and the counters that index the frame sequence -> one .npz (readback submits and waits)."""
        np.savez(path, converged=self._converged.GetPixels(), currentSample=self._currentSample, frame=self._frame,
                 frame_seed=self.frame_seed, size=(self.screen_width, self.screen_height))

    def LoadCheckpoint(self, path: str):
        """Resume: the next OnRenderImage blends frame `frame` with alpha 1 / (currentSample + 1) into the restored mean."""
        z = np.load(path, allow_pickle=False)
        w, h = (int(v) for v in z["size"])
        if (w, h) != (self.screen_width, self.screen_height):
            raise ValueError("checkpoint is of another resolution")
        if self._treesNeedRebuilding:                       # RM:850-859 would reset the sample counter: do the rebuild first
            self._treesNeedRebuilding = False
            if self._rayTraceObjects:
                self.RebuildObjectLists()
            self.RebuildTrees()
        self.InitRenderTexture()
        self._converged.SetPixels(z["converged"])
        self._currentSample, self._frame, self.frame_seed = int(z["currentSample"]), int(z["frame"]), int(z["frame_seed"])

    # RM:188-212
    def OnDisable(self):
        for b in (self._sphereBuffer, self._meshObjectBuffer, self._vertexBuffer, self._indexBuffer, self._normalBuffer,
                  self._sphereBVHBuffer, self._meshObjectBVHBuffer):
            if b is not None:
                b.Release()
        for t in (self._target, self._converged, self.SkyboxTexture):
            if t is not None:
                t.Release()
        self._target = self._converged = self.SkyboxTexture = None

    # ---- multi-GPU frame-end gather (no counterpart in the reference: it is single-GPU) -----------
    def gather_converged(self, dist, device):
        """One collective at frame end: every rank packs its strips of `_converged` into a dense device buffer
        (k_pack_rows), rank 0 gathers (RCCL; with the gloo backend the buffers are staged through host memory) and
        de-interleaves (k_pack_rows, reverse).  Returns the full image on rank 0 (numpy), else None.
        Synchronous by design (a convenience for tests and tools; bench.py pipelines the same steps on streams)."""
        import torch
        from . import strips
        from ._lib import UrtError
        if self._converged is None or not self._converged.handle:
            raise UrtError(2, "gather_converged: no accumulated image yet (render a frame first)")
        n_floats = strips.packed_rows(self.screen_height, self.world_size) * self.screen_width * 4
        mine = torch.zeros(n_floats, dtype=torch.float32, device=device)
        if mine.is_cuda:
            # the zero fill runs on torch's current stream, the pack kernel on the library's own stream: order them
            torch.cuda.current_stream(device).synchronize()
        self._converged.pack_rows(self.rank, self.world_size, mine.data_ptr())
        self.ctx.synchronize()
        staged = dist.get_backend() == "gloo"
        parts = strips.gather_to_root(dist, mine.cpu() if staged else mine, self.rank, self.world_size)
        if mine.is_cuda:
            torch.cuda.synchronize(device)
        if self.rank != 0:
            return None
        full = RenderTexture(self.ctx, self.screen_width, self.screen_height)
        keep = []
        for r, p in enumerate(parts):
            pd = p.to(device) if staged else p
            keep.append(pd)
            if pd.is_cuda:
                torch.cuda.current_stream(device).synchronize()             # the upload (torch's stream) before the unpack (library's stream)
            full.unpack_rows(r, self.world_size, pd.data_ptr())
        out = full.GetPixels()                                             # synchronises
        full.Release()
        return out
