"""GPU: the boundary is a plain C ABI.  examples/frame_loop.c — the reference's frame protocol in C99 against include/urt.h, no
Python, no C++ — is compiled with gcc, linked against libunityraytracer_amd.so and run; its accumulated image must be bit-identical
to the same protocol driven through the Python mirror (same buffers, same uniforms) AND to the scalar oracle's running mean, and
its PNG byte-identical."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle
from unityraytracer_amd import LIB_PATH, RayTraceMaster, host_io, host_scene, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_example_scene():
    """The scene of examples/frame_loop.c, built with the same float32 values."""
    W, H = 160, 96
    sp = np.zeros(4, dtype=scenes.SPHERE_DT)
    sx, sr = (-3.0, -1.0, 1.2, 3.2), (0.9, 0.6, 0.8, 0.5)
    mats = [((0.8, 0.2, 0.2), (0.04, 0.04, 0.04), 0.0, 0.2), ((0, 0, 0), (0.9, 0.9, 0.9), 0.0, 0.95), ((0.2, 0.7, 0.3), (0.3, 0.3, 0.3), 0.0, 0.6), ((0, 0, 0), (0, 0, 0), 3.0, 0.0)]
    for k in range(4):
        sp[k]["position"] = (sx[k], sr[k], (k % 2) * 1.5)
        sp[k]["radius"] = sr[k]
        a, s, e, sm = mats[k]
        sp[k]["lighting"] = scenes._params(a, s, (e, e, e), sm)
    v = np.array([[-4, 0, 4], [4, 0, 4], [4, 3, 4], [-4, 3, 4]], np.float32)
    idx = np.array([0, 2, 1, 0, 3, 2], np.int32)
    mo = np.zeros(1, dtype=scenes.MESHOBJECT_DT)
    mo[0]["localToWorldMatrix"] = np.eye(4, dtype=np.float32).T.reshape(16)
    mo[0]["indices_offset"], mo[0]["indices_count"] = 0, 6
    mo[0]["lighting"] = scenes._params((0.6, 0.6, 0.7), (0.1, 0.1, 0.1), (0, 0, 0), 0.3)
    yy = (np.arange(32, dtype=np.float32) / np.float32(31))[:, None]
    sky = np.zeros((32, 64, 4), np.float32)
    sky[..., 0] = np.float32(0.3) + np.float32(0.5) * yy
    sky[..., 1] = np.float32(0.4) + np.float32(0.5) * yy
    sky[..., 2] = np.float32(0.6) + np.float32(0.4) * yy
    sky[..., 3] = 1.0
    f = np.float32
    near, far, th, aspect = f(0.3), f(1000.0), f(0.85408069), f(W) / f(H)
    invp = np.zeros(16, np.float32)
    invp[0], invp[5], invp[14] = aspect * th, th, -1.0
    invp[11] = (near - far) / (f(2.0) * far * near)
    invp[15] = (far + near) / (f(2.0) * far * near)
    c2w = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 1, -10, 1], np.float32)
    sc = scenes.Scene("c-example", W, H, 6, 1, mesh_objects=mo, vertices=v, indices=idx, normals=host_scene.compute_normals(v, idx), spheres=sp,
                      mesh_bvh=host_scene.build_object_bvh(host_scene.mesh_leaf_bounds(mo, v, idx)),
                      sphere_bvh=host_scene.build_object_bvh(host_scene.sphere_leaf_bounds(sp), pairing=True), sky=sky,
                      camera_to_world=c2w, camera_inverse_projection=invp)
    return sc


def test_c99_host_matches_python_host(gpu_ctx, tmp_path):
    exe = str(tmp_path / "frame_loop")
    libdir = os.path.dirname(LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "frame_loop.c"), "-o", exe,
                    "-L" + libdir, "-lunityraytracer_amd", "-Wl,-rpath," + libdir], check=True)
    frames = 21
    out = str(tmp_path / "c_frame")
    r = subprocess.run([exe, str(frames), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"{frames} frames of 160x96" in r.stdout and "watchdog 0" in r.stdout
    assert f"in 1 launches ({frames} dispatches)" in r.stdout, r.stdout    # the per-frame present (RM:819) did not break the batch
    got = np.fromfile(out + ".rgba32f", dtype=np.float32).reshape(96, 160, 4)
    sc = c_example_scene()
    m = RayTraceMaster(gpu_ctx, sc)
    for _ in range(frames):
        m.OnRenderImage()
    want = m._converged.GetPixels()
    m.OnDisable()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    o = pyoracle.Oracle(sc)                                      # and both equal the scalar oracle's running mean
    ref = np.zeros((96, 160, 4), np.float32)
    for f in range(frames):
        ox, oy, seed = scenes.frame_uniforms(f)
        o.set_frame((ox, oy), seed)
        ref = pyoracle.accumulate(o.render(mode=0, threads=8), ref, f)       # mode 0: the reference's literal loops, no triangle BVH
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    ref_png = str(tmp_path / "py.png")
    host_io.write_png(ref_png, want)
    assert open(out + ".png", "rb").read() == open(ref_png, "rb").read()
    got8 = np.fromfile(out + ".rgba8", dtype=np.uint8).reshape(96, 160, 4)     # urt_texture_read_begin_format(URT_FORMAT_RGBA8_SRGB) from C
    assert np.array_equal(got8, host_io.encode_srgb8(want))
    assert (got[..., :3] > 0).mean() > 0.9                      # a real picture, not zeros
