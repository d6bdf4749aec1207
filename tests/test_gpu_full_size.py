"""GPU: the BASELINE.json configurations at their FULL sizes, bit-exact against the oracle (BVH-culled mode on the product's
BVH, all host threads).  C2/C3 are 1920x1080, C4/C5 3840x2160 with up to 983,040 triangles and 16 bounces; plus
size-independent properties at full size: traversal counters equal event for event, and a frame hash that does not depend on
the kernel mode."""
import hashlib
import os

import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import RayTraceMaster, scenes

pytestmark = pytest.mark.gpu


def threads():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except Exception:
        pass
    return max(1, min(n, 32))


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5"])
def test_full_size_config_bit_exact(gpu_ctx, cfg):
    sc = scenes.CONFIGS[cfg]()
    gpu_ctx.set_option("kernel_mode", 3)
    gpu_ctx.set_option("count_stats", 1)
    gpu_ctx.reset_counters()
    m = RayTraceMaster(gpu_ctx, sc)
    m.OnRenderImage()
    img = m._target.GetPixels()
    gc = gpu_ctx.counters()
    gpu_ctx.set_option("count_stats", 0)
    # the oracle walks the tree the product REALLY used (default builder: the host's SAH below 200,000 triangles, the GPU's from there on —
    # C4 and C5), read back from the device, so that the traversal counters can be compared event for event whichever builder ran
    o = pyoracle.Oracle(sc)
    if len(sc.mesh_objects):
        nodes, tri, root, _ = gpu_ctx.read_scene_blas(len(sc.mesh_objects))
        o.set_blas(nodes, tri, root)
        assert gpu_ctx.launch_info()["blas_builder"] == (3 if sc.n_triangles >= 200000 else 0), gpu_ctx.launch_info()
    ref, oc = o.render(mode=1, threads=threads(), counters=True)
    same = np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    if not same:
        d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
        bad = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
        raise AssertionError(f"{cfg}: {bad} of {sc.width * sc.height} pixels differ, max |d| = {np.nanmax(d):.3e} (north_star tolerance 1e-4)")
    for k in ("rays", "tlas_nodes", "blas_nodes", "tri_tests", "sphere_tests", "hit_tri", "hit_sphere", "hit_ground", "hit_sky"):
        assert gc[k] == oc[k], (cfg, k, gc[k], oc[k])
    assert gc["pixels"] == sc.width * sc.height and gc["watchdog_trips"] == 0
    # the frame does not depend on the kernel mode (same hash from the per-pixel and the persistent-regeneration kernels)
    h3 = hashlib.sha256(img.tobytes()).hexdigest()
    for mode in ((0, 2, 4, 5) if cfg in ("C2", "C3") else (2, 4, 5)):
        gpu_ctx.set_option("kernel_mode", mode)
        m._frame = 0
        m._currentSample = 0
        m.OnRenderImage()
        assert hashlib.sha256(m._target.GetPixels().tobytes()).hexdigest() == h3, (cfg, mode)
    gpu_ctx.set_option("kernel_mode", 3)
    m.OnDisable()
    if not len(sc.mesh_objects):
        return
    # ---- checks that do NOT share the product's BVH (VERDICT r1 "what's weak" 2 / ADVICE r1): a triangle lost by the
    # product's builder or wrongly culled by its traversal would be lost on both sides above ----
    # (1) the oracle walking its OWN, independently built median-split BVH: identical pixels (counters may differ)
    own = pyoracle.Oracle(sc)
    own.build_own_blas()
    ref_own = own.render(mode=1, threads=threads())
    assert np.array_equal(img.view(np.uint32), ref_own.view(np.uint32)), f"{cfg}: GPU frame differs from the oracle on its own BVH"
    # (2) literal brute force over every triangle (RS:243-266, oracle mode 0) on a 32x16 crop through EVERY MeshObject
    for k, (x0, y0) in enumerate(crops_through_meshes(sc, 32, 16)):
        bf = own.render(rect=(x0, y0, x0 + 32, y0 + 16), mode=0, threads=threads())
        assert np.array_equal(img[y0:y0 + 16, x0:x0 + 32].view(np.uint32), bf.view(np.uint32)), f"{cfg}: brute-force crop {k} at ({x0},{y0}) differs"


def crops_through_meshes(sc, cw, ch):
    """Lower-left corners of one cw x ch crop per MeshObject, centred on the projection of the MeshObject's bounding-box
    centre (SURVEY.md A.2 camera conventions), clamped to the frame; MeshObjects behind the camera are skipped."""
    lo, hi = scenes.mesh_bounds(sc.mesh_objects, sc.vertices, sc.indices)
    c2w = np.asarray(sc.camera_to_world, np.float64).reshape(4, 4).T          # column-major storage
    proj = np.linalg.inv(np.asarray(sc.camera_inverse_projection, np.float64).reshape(4, 4).T)
    w2c = np.linalg.inv(c2w)
    out = []
    for a, b in zip(lo, hi):
        p = w2c @ np.append((np.asarray(a, np.float64) + np.asarray(b, np.float64)) / 2, 1.0)
        q = proj @ p
        if q[3] <= 0:
            continue
        u, v = q[0] / q[3], q[1] / q[3]
        x = int(np.clip((u + 1) / 2 * sc.width - cw / 2, 0, sc.width - cw))
        y = int(np.clip((v + 1) / 2 * sc.height - ch / 2, 0, sc.height - ch))
        if (x, y) not in out:
            out.append((x, y))
    return out
