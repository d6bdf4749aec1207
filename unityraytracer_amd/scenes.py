"""Deterministic synthetic inputs for the path-tracing hot path.

Everything here produces *inputs* in the reference's own buffer layouts (RayTraceMaster.cs:42-45,
738-745): `_MeshObjects` (112 B), `_Vertices`/`_Normals` (12 B), `_Indices` (4 B), `_Spheres` (56 B),
`_MeshBVH`/`_SphereBVH` (28 B implicit heaps), camera matrices in Unity `Matrix4x4` memory order, an
equirect RGBA32F sky.  The reference's assets (Unity built-in meshes, HDR skies, Stanford bunny) are
not in the reference tree (SURVEY.md §7 "Missing assets"), so the five BASELINE.json configurations
are synthesised here from a splitmix64 stream with fixed seeds.

Host-side scene preparation that the reference does in C# is restated here in numpy:
  * `compute_normals`     — RayTraceMaster.ComputeNormals (RM:340-368): per-vertex sum of the
                            un-normalised face cross products of every index slot whose vertex position
                            equals this vertex's position (weld across all meshes), then normalised;
  * `build_object_bvh`    — a clean top-down median builder that emits the same implicit-heap format
                            RM:405-722 produces (children 2i+1/2i+2, `index < 0` interior, filler nodes
                            all-zero with index -1, length 2^D - 1 with D = ceil(log2 n) + 1).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# --- reference byte layouts -----------------------------------------------------
PARAMS_DT = np.dtype([("color_albedo", "<f4", 3), ("color_specular", "<f4", 3), ("emission", "<f4", 3), ("smoothness", "<f4")])
MESHOBJECT_DT = np.dtype([("localToWorldMatrix", "<f4", 16), ("indices_offset", "<i4"), ("indices_count", "<i4"), ("lighting", PARAMS_DT)])
SPHERE_DT = np.dtype([("position", "<f4", 3), ("radius", "<f4"), ("lighting", PARAMS_DT)])
BVHNODE_DT = np.dtype([("vmin", "<f4", 3), ("vmax", "<f4", 3), ("index", "<i4")])
assert PARAMS_DT.itemsize == 40 and MESHOBJECT_DT.itemsize == 112 and SPHERE_DT.itemsize == 56 and BVHNODE_DT.itemsize == 28


class SplitMix64:
    """splitmix64 (Steele/Lea/Flood): the documented PRNG behind every synthetic input."""

    MASK = (1 << 64) - 1

    def __init__(self, seed: int):
        self.state = seed & self.MASK

    def next_u64(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & self.MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.MASK
        return z ^ (z >> 31)

    def value(self) -> float:
        """Uniform in [0,1) with 24 bits — stands in for UnityEngine.Random.value (RM:777-778)."""
        return (self.next_u64() >> 40) / 16777216.0            # 24 bits / 2^24: exact in float32 (and in the double it is returned as)

    def uniform(self, lo: float, hi: float) -> float:
        return lo + (hi - lo) * self.value()


def frame_uniforms(frame: int, seed: int = 0x5EED):
    """(_PixelOffset.x, _PixelOffset.y, _Seed) of frame `frame` (RM:777-778).  Frame 0 is the fixed
    fixture (0.5, 0.5, 0.5); later frames come from splitmix64(seed) — three values per frame."""
    if frame == 0:
        return 0.5, 0.5, 0.5
    # splitmix64's state is seed + k * gamma after k draws, so frame f's three values (draws 3f-2 .. 3f) are reached directly
    rng = SplitMix64(seed + (3 * (frame - 1)) * 0x9E3779B97F4A7C15)
    return rng.value(), rng.value(), rng.value()


# --- camera (Scene1.unity:1777-1779,1804-1805) -------------------------------------
def camera_matrices(width: int, height: int, position=(0.0, 1.0, -10.0), fov_deg: float = 81.0, near: float = 0.3,
                    far: float = 1000.0, yaw_deg: float = 0.0, pitch_deg: float = 0.0):
    """(_CameraToWorld, _CameraInverseProjection) as 16 floats each in Unity Matrix4x4 memory order
    (column-major).  cameraToWorld = T * R * diag(1,1,-1); projection is GL-style."""
    aspect = width / height
    f = 1.0 / math.tan(math.radians(fov_deg) * 0.5)
    cy, sy = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    cp, sp = math.cos(math.radians(pitch_deg)), math.sin(math.radians(pitch_deg))
    # Unity: yaw about +y, pitch about +x (positive pitch looks down), left-handed
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], dtype=np.float64)
    rot = ry @ rx
    c2w = np.eye(4, dtype=np.float64)
    c2w[:3, :3] = rot @ np.diag([1.0, 1.0, -1.0])
    c2w[:3, 3] = position
    proj = np.zeros((4, 4), dtype=np.float64)
    proj[0, 0] = f / aspect
    proj[1, 1] = f
    proj[2, 2] = -(far + near) / (far - near)
    proj[2, 3] = -2.0 * far * near / (far - near)
    proj[3, 2] = -1.0
    invp = np.linalg.inv(proj)
    return (np.ascontiguousarray(c2w.T.astype(np.float32).reshape(16)), np.ascontiguousarray(invp.T.astype(np.float32).reshape(16)))


# --- sky ----------------------------------------------------------------------------------------
def make_sky(width: int = 2048, height: int = 1024) -> np.ndarray:
    """Analytic equirect RGBA32F sky (H, W, 4), row 0 = bottom (v = 0): zenith-horizon gradient plus a
    sun lobe.  Stands in for Assets/Skyboxes/*.hdr, whose blobs are absent from the reference tree."""
    v = (np.arange(height, dtype=np.float64) + 0.5) / height          # 0 bottom .. 1 top
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    # RS:424-425: v = -acos(d.y)/pi (mod 1) => elevation angle from straight up = (1 - v) * pi
    polar = (1.0 - v)[:, None] * math.pi
    azim = (u[None, :] - 0.5) * 2.0 * math.pi
    dy = np.cos(polar)
    dx = np.sin(polar) * np.sin(azim)
    dz = np.sin(polar) * np.cos(azim)
    up = np.clip(dy, 0.0, 1.0)
    down = np.clip(-dy, 0.0, 1.0)
    horizon = np.array([0.80, 0.86, 0.95])
    zenith = np.array([0.18, 0.36, 0.85])
    nadir = np.array([0.25, 0.22, 0.20])
    col = horizon[None, None, :] * (1.0 - up[..., None]) + zenith[None, None, :] * up[..., None]
    col = col * (1.0 - down[..., None]) + nadir[None, None, :] * down[..., None]
    sun_dir = np.array([0.45, 0.55, -0.70])
    sun_dir /= np.linalg.norm(sun_dir)
    c = np.clip(dx * sun_dir[0] + dy * sun_dir[1] + dz * sun_dir[2], 0.0, 1.0)
    lobe = 12.0 * c ** 256 + 0.6 * c ** 8
    col = col + lobe[..., None] * np.array([1.0, 0.92, 0.75])[None, None, :]
    out = np.ones((height, width, 4), dtype=np.float32)
    out[..., :3] = col.astype(np.float32)
    return out


# --- meshes -------------------------------------------------------------------------------------
def uv_blob(slices: int, stacks: int, bumps: float = 0.12, phase: float = 0.0):
    """Displaced UV sphere without seam duplicates: slices*(stacks-1)+2 vertices, 2*slices*(stacks-1)
    triangles (S=200, T=175 -> 69,600, the Stanford-bunny-class config C3).  Winding is outward."""
    th = (np.arange(1, stacks, dtype=np.float64) / stacks) * math.pi                  # polar, rings
    ph = (np.arange(slices, dtype=np.float64) / slices) * 2.0 * math.pi

    def radius(t, p):
        return 1.0 + bumps * (np.sin(3.0 * t + phase) * np.sin(4.0 * p) + 0.6 * np.sin(7.0 * p + 1.3 + phase) * np.sin(5.0 * t) ** 2)

    T, P = np.meshgrid(th, ph, indexing="ij")
    R = radius(T, P)
    ring = np.stack([R * np.sin(T) * np.cos(P), R * np.cos(T), R * np.sin(T) * np.sin(P)], axis=-1).reshape(-1, 3)
    top = np.array([[0.0, radius(0.0, 0.0), 0.0]])
    bot = np.array([[0.0, -radius(math.pi, 0.0), 0.0]])
    verts = np.concatenate([top, ring, bot]).astype(np.float32)
    n_ring = stacks - 1
    bot_i = 1 + n_ring * slices

    def vid(r, s):
        return 1 + r * slices + (s % slices)

    tris = []
    s = np.arange(slices)
    s1 = (s + 1) % slices
    tris.append(np.stack([np.zeros(slices, dtype=np.int64), 1 + s1, 1 + s], axis=1))                  # top fan
    for r in range(n_ring - 1):
        a, b = 1 + r * slices + s, 1 + r * slices + s1
        c, d = 1 + (r + 1) * slices + s, 1 + (r + 1) * slices + s1
        tris.append(np.stack([a, b, c], axis=1))
        tris.append(np.stack([b, d, c], axis=1))
    last = 1 + (n_ring - 1) * slices
    tris.append(np.stack([last + s, last + s1, np.full(slices, bot_i)], axis=1))                    # bottom fan
    idx = np.concatenate(tris).astype(np.int32)
    return verts, _orient_outward(verts, idx)


def icosphere(level: int, bumps: float = 0.0, phase: float = 0.0):
    """Icosphere with 20 * 4^level triangles (level 6 -> 81,920), optionally displaced radially."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        es = np.sort(e, axis=1)
        uniq, inv = np.unique(es, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid])
        n = len(f)
        m01, m12, m20 = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                            np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)])
    if bumps:
        th = np.arccos(np.clip(v[:, 1], -1, 1))
        ph = np.arctan2(v[:, 2], v[:, 0])
        r = 1.0 + bumps * (np.sin(3.0 * th + phase) * np.sin(4.0 * ph) + 0.6 * np.sin(6.0 * ph + 1.3 + phase) * np.sin(5.0 * th) ** 2)
        v = v * r[:, None]
    verts = v.astype(np.float32)
    return verts, _orient_outward(verts, f.astype(np.int32))


def quad(p0, p1, p2, p3):
    """Two triangles (p0,p1,p2), (p0,p2,p3); the visible side is chosen by the caller via the order."""
    return np.array([p0, p1, p2, p3], dtype=np.float32), np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)


def front_facing(v0, v1, v2, direction):
    """True where RS:199-213 would NOT cull: det = dot(v1-v0, cross(dir, v2-v0)) >= EPSILON."""
    e1, e2 = v1 - v0, v2 - v0
    return np.einsum("...i,...i", e1, np.cross(direction, e2)) > 0


def _orient_outward(verts, idx):
    """Order each triangle so a ray travelling INWARD (towards the centroid) passes the back-face
    culling of RS:211."""
    v0, v1, v2 = verts[idx[:, 0]].astype(np.float64), verts[idx[:, 1]].astype(np.float64), verts[idx[:, 2]].astype(np.float64)
    centre = verts.astype(np.float64).mean(axis=0)
    inward = centre - (v0 + v1 + v2) / 3.0
    ok = front_facing(v0, v1, v2, inward)
    out = idx.copy()
    out[~ok] = out[~ok][:, [0, 2, 1]]
    return out


def trs(translate=(0, 0, 0), scale=(1, 1, 1), yaw_deg: float = 0.0) -> np.ndarray:
    """localToWorldMatrix as 16 floats, Unity Matrix4x4 memory order (column-major)."""
    c, s = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    sc = np.array(scale, dtype=np.float64) if np.ndim(scale) else np.array([scale] * 3, dtype=np.float64)
    m = np.eye(4, dtype=np.float64)
    m[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]) @ np.diag(sc)
    m[:3, 3] = translate
    return np.ascontiguousarray(m.T.astype(np.float32).reshape(16))


def compute_normals(vertices: np.ndarray, indices: np.ndarray) -> np.ndarray:
    """RayTraceMaster.ComputeNormals (RM:340-368) in O(V + I): for vertex i, sum over every index
    slot whose vertex POSITION equals vertices[i] (exact equality: the reference's EPSILON is
    3 * float.Epsilon ~ 4e-45, RM:14,351) of cross(v[i1]-v[i0], v[i2]-v[i0]) of that slot's triangle,
    accumulated in float32 in ascending slot order, then Vector3.Normalize (zero for |v| < 1e-5)."""
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
    if len(v) == 0:
        return np.zeros((0, 3), dtype=np.float32)
    _, group = np.unique(v.view([("x", "<f4"), ("y", "<f4"), ("z", "<f4")]).reshape(-1), return_inverse=True)
    group = group.reshape(-1)
    tri = idx.reshape(-1, 3)
    a, b, c = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
    face = np.cross(b - a, c - a).astype(np.float32)                     # one per triangle
    acc = np.zeros((group.max() + 1, 3), dtype=np.float32)
    slot_group = group[idx]                                              # ascending slot order
    np.add.at(acc, slot_group, np.repeat(face, 3, axis=0))
    n = acc[group]
    ln = np.sqrt((n.astype(np.float32) ** 2).sum(axis=1, dtype=np.float32))
    out = np.zeros_like(n)
    ok = ln > 1e-5
    out[ok] = n[ok] / ln[ok, None]
    return out.astype(np.float32)


# --- object-level BVH in the reference's implicit-heap format (RM:405-722 output contract) -----------
def build_object_bvh(lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
    """lo/hi: (n,3) world-space bounds per object.  Returns BVHNODE_DT[2^D - 1], D = ceil(log2 n) + 1
    (RM:683,705).  Interior nodes: index -1 and the union of their children; a leaf may sit above the
    last level when a subtree holds one object (the reference's "paired with itself" case, RM:661-665)."""
    n = len(lo)
    if n == 0:
        return np.zeros(0, dtype=BVHNODE_DT)
    depth = int(math.ceil(math.log2(n))) + 1 if n > 1 else 1
    nodes = np.zeros((1 << depth) - 1, dtype=BVHNODE_DT)
    nodes["index"] = -1
    lo = np.asarray(lo, dtype=np.float32)
    hi = np.asarray(hi, dtype=np.float32)
    cen = (lo.astype(np.float64) + hi.astype(np.float64)) * 0.5

    def rec(slot: int, ids: np.ndarray):
        nodes[slot]["vmin"] = lo[ids].min(axis=0)
        nodes[slot]["vmax"] = hi[ids].max(axis=0)
        if len(ids) == 1:
            nodes[slot]["index"] = int(ids[0])
            return
        ext = cen[ids].max(axis=0) - cen[ids].min(axis=0)
        ax = int(np.argmax(ext))
        order = ids[np.lexsort((ids, cen[ids, ax]))]
        half = (len(order) + 1) // 2
        rec(2 * slot + 1, order[:half])
        rec(2 * slot + 2, order[half:])

    rec(0, np.arange(n))
    return nodes


# --- scene container ------------------------------------------------------------------------------
@dataclass
class Scene:
    """The flattened lists RebuildObjectLists/RebuildTrees hand to CreateComputeBuffer (RM:738-745) plus
    the per-frame uniforms of SetShaderParameters (RM:772-795)."""
    name: str
    width: int
    height: int
    num_bounces: int
    num_rays: int
    mesh_objects: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=MESHOBJECT_DT))
    vertices: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), dtype=np.float32))
    indices: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    normals: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), dtype=np.float32))
    spheres: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=SPHERE_DT))
    mesh_bvh: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=BVHNODE_DT))
    sphere_bvh: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=BVHNODE_DT))
    sky: np.ndarray = field(default_factory=lambda: make_sky(256, 128))
    camera_to_world: np.ndarray = None
    camera_inverse_projection: np.ndarray = None
    pixel_offset: tuple = (0.5, 0.5)
    seed: float = 0.5

    def __post_init__(self):
        if self.camera_to_world is None:
            self.camera_to_world, self.camera_inverse_projection = camera_matrices(self.width, self.height)

    @property
    def n_triangles(self) -> int:
        return int(self.mesh_objects["indices_count"].sum() // 3) if len(self.mesh_objects) else 0

    def resized(self, width: int, height: int, **cam) -> "Scene":
        """Same scene at another resolution (camera aspect follows)."""
        import copy
        s = copy.copy(self)
        s.width, s.height = width, height
        s.camera_to_world, s.camera_inverse_projection = camera_matrices(width, height, **cam)
        return s


def _params(albedo, specular, emission, smoothness):
    p = np.zeros((), dtype=PARAMS_DT)
    p["color_albedo"], p["color_specular"], p["emission"], p["smoothness"] = albedo, specular, emission, smoothness
    return p


def _random_material(rng: SplitMix64, k: int):
    colour = (rng.value(), rng.value(), rng.value())
    metal = (k % 2) == 1
    smooth = rng.value()
    emissive = (k % 8) == 7
    e = rng.uniform(1.0, 4.0)
    emission = tuple(e * c for c in colour) if emissive else (0.0, 0.0, 0.0)
    return _params((0.0, 0.0, 0.0) if metal else colour, colour if metal else (0.04, 0.04, 0.04), emission, smooth)


def make_spheres(n: int, half_extent: float, seed: int) -> np.ndarray:
    """n analytic spheres: centres uniform in x,z in [-half_extent, half_extent], r in [0.25,1], y = r
    (resting on the ground plane), colours U[0,1], odd ones metallic, every 8th emissive."""
    rng = SplitMix64(seed)
    sp = np.zeros(n, dtype=SPHERE_DT)
    for k in range(n):
        r = rng.uniform(0.25, 1.0)
        x, z = rng.uniform(-half_extent, half_extent), rng.uniform(-half_extent, half_extent)
        sp[k]["position"] = (x, r, z)
        sp[k]["radius"] = r
        sp[k]["lighting"] = _random_material(rng, k)
    return sp


def sphere_bounds(spheres: np.ndarray):
    p, r = spheres["position"], spheres["radius"][:, None]
    return p - r, p + r


class MeshSceneBuilder:
    """Concatenates meshes the way RebuildObjectLists does (RM:296-320): vertices appended, indices
    offset by the first vertex, one MeshObject per mesh; normals computed over the concatenation."""

    def __init__(self):
        self.verts, self.idx, self.objs = [], [], []
        self.nv = 0
        self.ni = 0

    def add(self, verts, tris, matrix16, lighting):
        verts = np.asarray(verts, dtype=np.float32).reshape(-1, 3)
        flat = np.asarray(tris, dtype=np.int32).reshape(-1)
        mo = np.zeros((), dtype=MESHOBJECT_DT)
        mo["localToWorldMatrix"] = matrix16
        mo["indices_offset"] = self.ni
        mo["indices_count"] = len(flat)
        mo["lighting"] = lighting
        self.verts.append(verts)
        self.idx.append(flat + self.nv)
        self.objs.append(mo)
        self.nv += len(verts)
        self.ni += len(flat)

    def finish(self):
        vertices = np.concatenate(self.verts) if self.verts else np.zeros((0, 3), np.float32)
        indices = np.concatenate(self.idx).astype(np.int32) if self.idx else np.zeros(0, np.int32)
        mesh_objects = np.array(self.objs, dtype=MESHOBJECT_DT) if self.objs else np.zeros(0, MESHOBJECT_DT)
        normals = compute_normals(vertices, indices)
        lo, hi = mesh_bounds(mesh_objects, vertices, indices)
        return mesh_objects, vertices, indices, normals, build_object_bvh(lo, hi)


def world_vertices(mo, vertices, indices):
    m = np.asarray(mo["localToWorldMatrix"], dtype=np.float64).reshape(4, 4).T
    sl = indices[int(mo["indices_offset"]): int(mo["indices_offset"]) + int(mo["indices_count"])]
    v = vertices[sl].astype(np.float64)
    return v @ m[:3, :3].T + m[:3, 3]


def mesh_bounds(mesh_objects, vertices, indices):
    lo = np.zeros((len(mesh_objects), 3), np.float32)
    hi = np.zeros((len(mesh_objects), 3), np.float32)
    for k, mo in enumerate(mesh_objects):
        w = world_vertices(mo, vertices, indices)
        # outward rounding so the float32 box contains the float32 world vertices the kernel computes
        lo[k] = np.nextafter(w.min(axis=0).astype(np.float32), np.float32(-np.inf))
        hi[k] = np.nextafter(w.max(axis=0).astype(np.float32), np.float32(np.inf))
    return lo, hi


# --- the five BASELINE.json configurations --------------------------------------------------------
def config1(width=256, height=256, sky=None) -> Scene:
    """C1: 16 analytic spheres, 256x256, 1 bounce."""
    sp = make_spheres(16, 8.0, seed=0xC1)
    return Scene("C1-16spheres", width, height, 1, 1, spheres=sp, sphere_bvh=build_object_bvh(*sphere_bounds(sp)),
                 sky=sky if sky is not None else make_sky(512, 256))


def config2(width=1920, height=1080, sky=None) -> Scene:
    """C2: 64 spheres + ground plane, 1920x1080, 4 bounces, 1 spp."""
    sp = make_spheres(64, 16.0, seed=0xC2)
    return Scene("C2-64spheres", width, height, 4, 1, spheres=sp, sphere_bvh=build_object_bvh(*sphere_bounds(sp)),
                 sky=sky if sky is not None else make_sky())


def config3(width=1920, height=1080, slices=200, stacks=175, sky=None) -> Scene:
    """C3: one bunny-class mesh (UV blob, 2*S*(T-1) = 69,600 triangles) resting on y = 0, no spheres,
    1920x1080, 8 bounces — the configuration the headline metric is quoted on.  The mesh is ~4.8 units
    tall, 6 units in front of Scene1's camera, so that it fills about half the frame height and the
    triangle-BVH traversal (not the sky/ground early-outs) dominates the frame."""
    v, t = uv_blob(slices, stacks)
    b = MeshSceneBuilder()
    miny = float(v[:, 1].min())
    scale = 2.2
    b.add(v, t, trs(translate=(0.0, -miny * scale, -4.0), scale=scale), _params((0.75, 0.55, 0.35), (0.15, 0.15, 0.15), (0, 0, 0), 0.55))
    mo, vv, ii, nn, bvh = b.finish()
    return Scene(f"C3-blob{len(ii) // 3}", width, height, 8, 1, mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh,
                 sky=sky if sky is not None else make_sky())


def config3_dense(width=1920, height=1080, sky=None) -> Scene:
    """C3D: the C3 mesh in close-up (camera at (0, 2.4, -7.3), 0.8 units in front of the mesh's bounding box): every primary
    ray enters the triangle BVH and about half of them end on a triangle — the frame on which "during BVH traversal" can
    be measured.  Same mesh, sky, bounces and resolution as C3."""
    sc = config3(width, height, sky=sky)
    sc = sc.resized(width, height, position=(0.0, 2.4, -7.3))
    sc.name = sc.name.replace("C3-", "C3D-closeup-")
    return sc


def _cornell(b: MeshSceneBuilder, half=5.0, height=10.0, zc=0.0):
    x0, x1, y0, y1, z0, z1 = -half, half, 0.002, height, zc - half, zc + half
    white = _params((0.73, 0.73, 0.73), (0, 0, 0), (0, 0, 0), 0.1)
    red = _params((0.65, 0.05, 0.05), (0, 0, 0), (0, 0, 0), 0.1)
    green = _params((0.12, 0.45, 0.15), (0, 0, 0), (0, 0, 0), 0.1)
    light = _params((0.0, 0.0, 0.0), (0, 0, 0), (15.0, 15.0, 15.0), 0.0)
    centre = np.array([0.0, height * 0.5, zc])

    def inward(p0, p1, p2, p3, mat):
        v, t = quad(p0, p1, p2, p3)
        c = v.mean(axis=0)
        if not front_facing(v[0].astype(np.float64), v[1].astype(np.float64), v[2].astype(np.float64), (c - centre).astype(np.float64)):
            t = t[:, [0, 2, 1]]
        b.add(v, t, trs(), mat)

    inward((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), white)      # floor (2 mm above the ground plane)
    inward((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), white)      # ceiling
    inward((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), white)      # back
    inward((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), red)        # left
    inward((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), green)      # right
    l = half * 0.35
    inward((-l, y1 - 0.01, zc - l), (l, y1 - 0.01, zc - l), (l, y1 - 0.01, zc + l), (-l, y1 - 0.01, zc + l), light)


def config4(width=3840, height=2160, slices=250, stacks=201, sky=None) -> Scene:
    """C4: Cornell box (5 walls + emissive ceiling quad = 6 MeshObjects, 12 triangles) + 3 blobs of
    2*250*200 = 100,000 triangles each, 3840x2160, 8 bounces."""
    b = MeshSceneBuilder()
    _cornell(b)
    mats = [_params((0.8, 0.8, 0.3), (0.1, 0.1, 0.1), (0, 0, 0), 0.4), _params((0.0, 0.0, 0.0), (0.9, 0.9, 0.9), (0, 0, 0), 0.95),
            _params((0.3, 0.5, 0.9), (0.2, 0.2, 0.2), (0, 0, 0), 0.7)]
    for k, (x, z, s) in enumerate([(-2.4, 1.5, 1.5), (2.2, 0.5, 1.3), (0.0, -2.0, 1.0)]):
        v, t = uv_blob(slices, stacks, phase=0.7 * k)
        b.add(v, t, trs(translate=(x, -float(v[:, 1].min()) * s + 0.002, z), scale=s, yaw_deg=25.0 * k), mats[k])
    mo, vv, ii, nn, bvh = b.finish()
    return Scene(f"C4-cornell{len(ii) // 3}", width, height, 8, 1, mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh,
                 sky=sky if sky is not None else make_sky())


def config5(width=3840, height=2160, level=6, n_blobs=12, sky=None) -> Scene:
    """C5: 12 displaced icospheres x 81,920 triangles = 983,040 triangles, 3840x2160, 16 bounces
    (1024 progressive frames in the full run)."""
    rng = SplitMix64(0xC5)
    b = MeshSceneBuilder()
    for k in range(n_blobs):
        v, t = icosphere(level, bumps=0.10, phase=0.5 * k)
        s = rng.uniform(0.8, 1.6)
        x, z = (k % 4 - 1.5) * 4.2 + rng.uniform(-0.5, 0.5), (k // 4) * 4.5 - 3.0 + rng.uniform(-0.5, 0.5)
        b.add(v, t, trs(translate=(x, -float(v[:, 1].min()) * s, z), scale=s, yaw_deg=30.0 * k), _random_material(rng, k))
    mo, vv, ii, nn, bvh = b.finish()
    return Scene(f"C5-ico{len(ii) // 3}", width, height, 16, 1, mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh,
                 sky=sky if sky is not None else make_sky())


def mixed_test_scene(width=96, height=64, n_spheres=5, blob=(12, 9), sky=None) -> Scene:
    """Small scene with every primitive kind (rotated + scaled meshes, spheres, emitters) for parity tests."""
    rng = SplitMix64(0x7E57)
    b = MeshSceneBuilder()
    v, t = uv_blob(*blob)
    b.add(v, t, trs(translate=(-2.0, 1.3, 1.0), scale=(1.2, 1.0, 0.9), yaw_deg=33.0), _params((0.7, 0.4, 0.3), (0.2, 0.2, 0.2), (0, 0, 0), 0.6))
    v2, t2 = icosphere(1)
    b.add(v2, t2, trs(translate=(2.5, 1.0, -1.0), scale=1.0, yaw_deg=-20.0), _params((0.0, 0.0, 0.0), (0.85, 0.85, 0.85), (0, 0, 0), 0.9))
    qv, qt = quad((-1.5, 3.5, -0.5), (1.5, 3.5, -0.5), (1.5, 3.5, 2.5), (-1.5, 3.5, 2.5))
    if not front_facing(qv[0].astype(np.float64), qv[1].astype(np.float64), qv[2].astype(np.float64), np.array([0.0, 1.0, 0.0])):
        qt = qt[:, [0, 2, 1]]
    b.add(qv, qt, trs(), _params((0, 0, 0), (0, 0, 0), (6.0, 5.0, 4.0), 0.0))
    mo, vv, ii, nn, bvh = b.finish()
    sp = make_spheres(n_spheres, 5.0, seed=rng.next_u64())
    return Scene("mixed", width, height, 4, 1, mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh, spheres=sp,
                 sphere_bvh=build_object_bvh(*sphere_bounds(sp)), sky=sky if sky is not None else make_sky(128, 64))


def many_meshes_scene(width=128, height=80, n=300, level=0, sky=None) -> Scene:
    """n small MeshObjects (icospheres of 20 * 4^level triangles: each has interior BVH nodes) on a jittered grid: more
    object-level heap nodes (511 for n = 300) and more BVH roots than the library keeps in LDS — exercises the paths
    that fall back to global memory."""
    rng = SplitMix64(0xA11CE)
    b = MeshSceneBuilder()
    v, t = icosphere(level)
    side = int(math.ceil(math.sqrt(n)))
    for i in range(n):
        gx, gz = i % side, i // side
        x = (gx - side / 2.0) * 1.3 + 0.4 * (rng.value() - 0.5)
        z = (gz - side / 2.0) * 1.3 + 4.0 + 0.4 * (rng.value() - 0.5)
        r = 0.35 + 0.25 * rng.value()
        col = (rng.value(), rng.value(), rng.value())
        metal = i % 3 == 0
        lighting = _params((0, 0, 0) if metal else col, col if metal else (0.04, 0.04, 0.04), (2.0, 1.5, 1.0) if i % 17 == 0 else (0, 0, 0), rng.value())
        b.add(v, t, trs(translate=(x, r, z), scale=r, yaw_deg=360.0 * rng.value()), lighting)
    mo, vv, ii, nn, bvh = b.finish()
    return Scene(f"many-meshes{n}", width, height, 4, 1, mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh,
                 sky=sky if sky is not None else make_sky(128, 64))


CONFIGS = {"C1": config1, "C2": config2, "C3": config3, "C4": config4, "C5": config5, "C3D": config3_dense}


# --- the reference's own scenes (tests/golden/scene_*.json, mined from Assets/Scenes/*.unity by tests/golden/make_scene_fixtures.py) ---
def deep_chain_scene(width=96, height=64, n=40, ratio=3.0, grow=1.8) -> Scene:
    """One MeshObject whose triangles stand behind each other at geometrically growing distances (x ratio) and sizes (x grow): every SAH
    split peels off the far end, so with one triangle per leaf the BVH is a chain as deep as it has triangles — and a camera ray descends
    into the near (interior) child at every level with the far leaf pushed: the stack holds one entry per level.
    Exercises the depth of the traversal stacks (tests/test_gpu_parity.py)."""
    vs, ts = [], []
    for k in range(n):
        z = -8.0 + ratio ** k
        s = 0.2 * grow ** k
        base = len(vs)
        vs += [(-s + 0.01 * k, 1 - s, z), (0.01 * k, 1 + s, z), (s + 0.01 * k, 1 - s, z)]     # counter-clockwise seen from the camera (-z)
        ts.append((base, base + 1, base + 2))
    b = MeshSceneBuilder()
    b.add(np.array(vs, np.float32), np.array(ts, np.int32), trs(), _params((0.8, 0.6, 0.4), (0.1, 0.1, 0.1), (0, 0, 0), 0.3))
    mo, vv, ii, nn, bvh = b.finish()
    return Scene("deep-chain", width, height, 3, 1, mesh_objects=mo, vertices=vv, indices=ii, normals=nn, mesh_bvh=bvh, sky=make_sky(64, 32))


def trs_quat(translate=(0, 0, 0), quat=(0, 0, 0, 1), scale=(1, 1, 1)) -> np.ndarray:
    """Matrix4x4.TRS(position, rotation, scale) as 16 floats in Unity memory order (column-major); quat = (x, y, z, w)."""
    x, y, z, w = (float(c) for c in quat)
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)
    m = np.eye(4, dtype=np.float64)
    m[:3, :3] = r @ np.diag(np.array(scale, dtype=np.float64))
    m[:3, 3] = translate
    return np.ascontiguousarray(m.T.astype(np.float32).reshape(16))


def unity_builtin_mesh(kind: str):
    """Stand-ins for Unity's built-in meshes (the editor's default resources are not in the reference's tree): same shape, extent
    and triangle count as the originals — cube 12, quad 2, plane 200, cylinder 80 (20 sides, radius 0.5, height 2), capsule 832
    (radius 0.5, height 2), sphere 768 — with one vertex per position (the reference welds normals by position anyway, RM:351)
    and triangles ordered so that rays from outside pass the back-face culling of RS:211.  Vertex order inside the originals is
    unknown; nothing in the path depends on it except the order of exact ties."""
    if kind == "cube":
        v = np.array([[x, y, z] for x in (-0.5, 0.5) for y in (-0.5, 0.5) for z in (-0.5, 0.5)], dtype=np.float32)
        faces = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
        t = np.array([[a, b, c] for a, b, c, d in faces] + [[a, c, d] for a, b, c, d in faces], dtype=np.int32)
        return v, _orient_outward(v, t)
    if kind == "quad":                                     # unit quad in the xy plane, visible from -z
        v, t = quad((-0.5, -0.5, 0), (0.5, -0.5, 0), (0.5, 0.5, 0), (-0.5, 0.5, 0))
        if not front_facing(v[0].astype(np.float64), v[1].astype(np.float64), v[2].astype(np.float64), np.array([0.0, 0.0, 1.0])):
            t = t[:, [0, 2, 1]]
        return v, t
    if kind == "plane":                                    # 10 x 10 units, 10 x 10 cells, visible from +y
        n = 11
        v = np.array([[-5 + i, 0, -5 + j] for j in range(n) for i in range(n)], dtype=np.float32)
        t = []
        for j in range(n - 1):
            for i in range(n - 1):
                a, b, c, d = j * n + i, j * n + i + 1, (j + 1) * n + i + 1, (j + 1) * n + i
                t += [[a, b, c], [a, c, d]]
        t = np.array(t, dtype=np.int32)
        if not front_facing(v[t[0, 0]].astype(np.float64), v[t[0, 1]].astype(np.float64), v[t[0, 2]].astype(np.float64), np.array([0.0, -1.0, 0.0])):
            t = t[:, [0, 2, 1]]
        return v, t
    if kind in ("cylinder", "capsule", "sphere"):
        sides = {"cylinder": 20, "capsule": 26, "sphere": 24}[kind]
        rings = []                                         # (y, radius) from the bottom pole to the top pole, poles excluded
        if kind == "cylinder":
            rings = [(-1.0, 0.5), (1.0, 0.5)]
        elif kind == "capsule":                            # two hemispheres of radius 0.5 around y = -0.5 / +0.5: 7 quad rows + a fan each, one row between
            for k in range(1, 9):
                a = math.pi / 2 * k / 8
                rings.append((-0.5 - 0.5 * math.cos(a), 0.5 * math.sin(a)))
            for k in range(8, 0, -1):
                a = math.pi / 2 * k / 8
                rings.append((0.5 + 0.5 * math.cos(a), 0.5 * math.sin(a)))
        else:                                              # 17 latitudes: 15 quad rows + two fans = 2 * 24 * 16 = 768 triangles
            for k in range(1, 17):
                a = math.pi * k / 17
                rings.append((-0.5 * math.cos(a), 0.5 * math.sin(a)))
        ybot, ytop = (-1.0, 1.0) if kind != "sphere" else (-0.5, 0.5)
        verts = [[0.0, ybot, 0.0]]
        for y, r in rings:
            for s_ in range(sides):
                a = 2 * math.pi * s_ / sides
                verts.append([r * math.cos(a), y, r * math.sin(a)])
        verts.append([0.0, ytop, 0.0])
        top = len(verts) - 1
        t = []
        for s_ in range(sides):
            t.append([0, 1 + s_, 1 + (s_ + 1) % sides])
            base = 1 + (len(rings) - 1) * sides
            t.append([top, base + (s_ + 1) % sides, base + s_])
        for q in range(len(rings) - 1):
            for s_ in range(sides):
                a, b = 1 + q * sides + s_, 1 + q * sides + (s_ + 1) % sides
                c, d = a + sides, b + sides
                t += [[a, c, d], [a, d, b]]
        verts = np.array(verts, dtype=np.float32)
        return verts, _orient_outward(verts, np.array(t, dtype=np.int32))
    raise ValueError(f"no stand-in for Unity mesh {kind!r}")


def from_unity_fixture(fixture, width: int, height: int, sky=None) -> Scene:
    """A scene of the reference itself, from its mined fixture (a dict or a path to tests/golden/scene_*.json): the ENABLED
    RayTraceObjects in the order the fixture lists them, the scene's camera (position, pitch/yaw from its quaternion, field of view)
    and its numBounces / numRays.  Flattened like RebuildObjectLists (RM:262-336): spheres from the collider radius x largest
    lossy scale (RO:33), meshes concatenated with their localToWorldMatrix."""
    import json
    if not isinstance(fixture, dict):
        fixture = json.load(open(fixture))
    b = MeshSceneBuilder()
    sph = []
    for o in fixture["objects"]:
        if not o["enabled"]:
            continue
        lighting = _params(tuple(o["albedoColor"]), tuple(o["specularColor"]), tuple(o["emissionColor"]), float(o["smoothness"]))
        if o["type"] == "sphere":
            s_ = np.zeros((), dtype=SPHERE_DT)
            s_["position"], s_["radius"], s_["lighting"] = tuple(o["position"]), float(o["radius"]), lighting
            sph.append(s_)
        else:
            v, t = unity_builtin_mesh(o["mesh"])
            b.add(v, t, trs_quat(o["position"], o["rotation"], o["scale"]), lighting)
    mo, vv, ii, nn, bvh = b.finish()
    spheres = np.array(sph, dtype=SPHERE_DT) if sph else np.zeros(0, SPHERE_DT)
    cam = fixture["camera"]
    x, y, z, w = cam["rotation"]
    fwd = (2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y))           # rotation applied to (0, 0, 1)
    yaw, pitch = math.degrees(math.atan2(fwd[0], fwd[2])), -math.degrees(math.asin(max(-1.0, min(1.0, fwd[1]))))
    sc = Scene(f"unity-{fixture['source'].split('.')[0]}", width, height, int(fixture["numBounces"]), int(fixture["numRays"]), mesh_objects=mo, vertices=vv,
               indices=ii, normals=nn, mesh_bvh=bvh if len(mo) else np.zeros(0, BVHNODE_DT), spheres=spheres,
               sphere_bvh=build_object_bvh(*sphere_bounds(spheres)) if len(spheres) else np.zeros(0, BVHNODE_DT),
               sky=sky if sky is not None else make_sky(512, 256))
    name = sc.name
    sc = sc.resized(width, height, position=tuple(cam["position"]), fov_deg=float(cam["field_of_view"]), near=float(cam["near"]), far=float(cam["far"]),
                    yaw_deg=yaw, pitch_deg=pitch)
    sc.name = name
    return sc
