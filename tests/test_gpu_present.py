"""GPU: the last hop of RM:819 `Graphics.Blit(_converged, destination)` for a destination in host memory — the image converted ON THE GPU to
the destination's format (csrc/present.hip) and read back through the pipelined readback (include/urt.h urt_texture_read_begin_format).

  RGBA8 sRGB  == urt_host_encode_srgb8 (the PNG writer's bytes) for every float around every step of the encoder, the specials, and a
                 rendered frame;
  RGBA16F     == IEEE round-to-nearest-even (numpy's float32 -> float16), bit for bit;
  tickets of different formats in flight together; a converted ticket cannot be ended as floats; unknown formats are refused."""
import ctypes as C

import numpy as np
import pytest

from unityraytracer_amd import RayTraceMaster, RenderTexture, UrtError, host_io, scenes
from test_host_io import present_probe_image

pytestmark = pytest.mark.gpu


def half_bits_equal(got, src):
    want = src.astype(np.float16)
    nan = np.isnan(want)
    return np.array_equal(np.isnan(got), nan) and np.array_equal(got.view(np.uint16)[~nan], want.view(np.uint16)[~nan])


def test_encoders_on_every_step_and_the_specials(gpu_ctx):
    img = present_probe_image()
    h, w = img.shape[:2]
    t = RenderTexture(gpu_ctx, w, h)
    t.SetPixels(img)
    got8 = t.ReadEnd(t.ReadBegin("RGBA8_SRGB"))
    assert got8.dtype == np.uint8 and got8.shape == (h, w, 4)
    want8 = host_io.encode_srgb8(img)
    assert np.array_equal(got8, want8), int((got8 != want8).sum())
    with np.errstate(over="ignore"):
        got16 = t.ReadEnd(t.ReadBegin("RGBA16F"))
        assert got16.dtype == np.float16 and half_bits_equal(got16, img)
    got32 = t.ReadEnd(t.ReadBegin("RGBA32F"))
    assert np.array_equal(got32.view(np.uint32), img.view(np.uint32))
    # float16's own ties and range ends, every exponent
    rng = np.random.default_rng(5)
    b = rng.integers(0, 1 << 32, size=(h, w, 4), dtype=np.uint64).astype(np.uint32)
    b[0, :, 0] = (np.arange(w, dtype=np.uint32) << 13) | 0x38000FFF                    # just below / at / above the half-way points
    b[1, :, 0] = (np.arange(w, dtype=np.uint32) << 13) | 0x38001000
    b[2, :, 0] = (np.arange(w, dtype=np.uint32) << 13) | 0x38001001
    raw = b.view(np.float32)
    t.SetPixels(raw)
    with np.errstate(over="ignore", invalid="ignore"):
        assert half_bits_equal(t.ReadEnd(t.ReadBegin("RGBA16F")), raw)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(t.ReadEnd(t.ReadBegin("RGBA8_SRGB")), host_io.encode_srgb8(raw))
    t.Release()


def test_present_formats_of_a_rendered_frame_in_flight_together(gpu_ctx):
    sc = scenes.mixed_test_scene(168, 96)
    gpu_ctx.set_option("kernel_mode", 3)
    for fpl in (1, 0):
        gpu_ctx.set_option("frames_per_launch", fpl)
        try:
            m = RayTraceMaster(gpu_ctx, sc)
            want = []
            for _ in range(6):
                m.OnRenderImage()
                want.append(m._converged.GetPixels())
            m.OnDisable()
            m = RayTraceMaster(gpu_ctx, sc)
            fmts = ["RGBA8_SRGB", "RGBA16F", "RGBA32F", "RGBA8_SRGB", "RGBA8_SRGB", "RGBA16F"]
            tickets, got = [], []
            for i in range(6):
                m.OnRenderImage()
                tickets.append(m._converged.ReadBegin(fmts[i]))
                if len(tickets) == 3:
                    got.append(m._converged.ReadEnd(tickets.pop(0)))
            while tickets:
                got.append(m._converged.ReadEnd(tickets.pop(0)))
            for i in range(6):
                if fmts[i] == "RGBA8_SRGB":
                    assert np.array_equal(got[i], host_io.encode_srgb8(want[i])), (fpl, i)
                elif fmts[i] == "RGBA16F":
                    assert half_bits_equal(got[i], want[i]), (fpl, i)
                else:
                    assert np.array_equal(got[i].view(np.uint32), want[i].view(np.uint32)), (fpl, i)
            # a converted ticket is not a float image; unknown formats are refused; the ticket stays usable after the refusal
            tk = m._converged.ReadBegin("RGBA8_SRGB")
            p = C.POINTER(C.c_float)()
            assert gpu_ctx.lib.urt_texture_read_end(gpu_ctx._h, C.c_uint64(tk), C.byref(p)) == 1
            assert b"urt_texture_read_end_format" in gpu_ctx.lib.urt_last_error(gpu_ctx._h)
            assert np.array_equal(m._converged.ReadEnd(tk), host_io.encode_srgb8(want[5]))
            t2 = C.c_uint64()
            assert gpu_ctx.lib.urt_texture_read_begin_format(gpu_ctx._h, m._converged.handle, 7, C.byref(t2)) == 1
            m.OnDisable()
        finally:
            gpu_ctx.set_option("frames_per_launch", 0)
