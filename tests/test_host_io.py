"""Radiance .hdr loader and frame writers (csrc/host_io.cpp through the C ABI).  The reference's sky
assets are absent from its tree, so the loader is pinned on synthetic RGBE files written here (flat and new-style RLE) —
"parity unpinned" against Unity's own importer.  CPU only."""
import struct
import zlib

import numpy as np
import pytest

from unityraytracer_amd import UrtError, host_io


def float_to_rgbe(img):
    """Ward's float2rgbe on an (H, W, 3) array -> (H, W, 4) uint8."""
    m = img.max(axis=2)
    out = np.zeros(img.shape[:2] + (4,), np.uint8)
    ok = m > 1e-32
    mant, exp = np.frexp(m[ok])
    scale = mant * 256.0 / m[ok]
    out[ok, :3] = (img[ok] * scale[:, None]).astype(np.uint8)
    out[ok, 3] = (exp + 128).astype(np.uint8)
    return out


def rgbe_to_float(rgbe):
    f = np.ldexp(1.0, rgbe[..., 3].astype(np.int32) - 136)
    out = rgbe[..., :3].astype(np.float64) * f[..., None]
    out[rgbe[..., 3] == 0] = 0
    return out.astype(np.float32)


def rle_channel(vals):
    out, i = bytearray(), 0
    while i < len(vals):
        run = 1
        while i + run < len(vals) and run < 127 and vals[i + run] == vals[i]:
            run += 1
        if run >= 4:
            out += bytes([128 + run, vals[i]]); i += run
        else:
            j = i
            while j < len(vals) and j - i < 128:
                k = 1
                while j + k < len(vals) and k < 4 and vals[j + k] == vals[j]:
                    k += 1
                if k >= 4:
                    break
                j += 1
            out += bytes([j - i]) + bytes(vals[i:j]); i = j
    return bytes(out)


def write_hdr(path, rgbe_top_down, rle):
    h, w = rgbe_top_down.shape[:2]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nEXPOSURE=1.0\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
        for row in rgbe_top_down:
            if rle:
                f.write(bytes([2, 2, w >> 8, w & 255]))
                for ch in range(4):
                    f.write(rle_channel(row[:, ch].tolist()))
            else:
                f.write(row.tobytes())


@pytest.mark.parametrize("rle", [False, True])
def test_load_hdr_decodes_rgbe_and_flips_rows(built_library, tmp_path, rle):
    rng = np.random.default_rng(3)
    img = (rng.uniform(0, 1, (12, 40, 3)) ** 4 * 50).astype(np.float32)
    img[3, 5:25] = img[3, 5]                                   # long runs for the RLE path
    img[0, :3] = 0
    rgbe = float_to_rgbe(img.astype(np.float64))
    p = str(tmp_path / ("sky_rle.hdr" if rle else "sky_flat.hdr"))
    write_hdr(p, rgbe, rle)
    got = host_io.load_hdr(p)
    assert got.shape == (12, 40, 4) and (got[..., 3] == 1).all()
    want = rgbe_to_float(rgbe)[::-1]                            # file rows are top-down, library rows bottom-up
    assert np.array_equal(got[..., :3], want)
    err = np.abs(got[..., :3] - img[::-1])
    assert (err <= img[::-1].max(axis=2, keepdims=True) / 128 + 1e-6).all()   # shared exponent: 8 bits relative to the largest channel


def test_load_hdr_rejects_other_files(built_library, tmp_path):
    p = tmp_path / "bad.hdr"
    p.write_bytes(b"P6\n2 2\n255\n" + bytes(12))
    with pytest.raises(UrtError):
        host_io.load_hdr(str(p))
    with pytest.raises(UrtError):
        host_io.load_hdr(str(tmp_path / "missing.hdr"))
    q = tmp_path / "trunc.hdr"
    q.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 4\n" + bytes(10))
    with pytest.raises(UrtError):
        host_io.load_hdr(str(q))


def test_write_pfm_roundtrip(built_library, tmp_path):
    rng = np.random.default_rng(5)
    img = rng.normal(size=(7, 9, 4)).astype(np.float32)
    p = str(tmp_path / "f.pfm")
    host_io.write_pfm(p, img)
    data = open(p, "rb").read()
    header, rest = data.split(b"\n-1.0\n", 1)
    assert header == b"PF\n9 7"
    back = np.frombuffer(rest, "<f4").reshape(7, 9, 3)
    assert np.array_equal(back, img[..., :3])                  # bottom row first, exactly the library's row order


def test_write_png_is_a_valid_srgb_png(built_library, tmp_path):
    img = np.zeros((5, 6, 4), np.float32)
    img[..., 0] = np.linspace(0, 1, 6)[None, :]
    img[0, :, 1] = 1.0                                          # bottom row green
    img[4, 0] = (np.nan, -1.0, 7.0, 1.0)                        # NaN and out-of-range values clamp
    p = str(tmp_path / "shot.png")
    host_io.write_png(p, img)
    data = open(p, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body)
        chunks[tag] = body
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (w, h, depth, ctype) == (6, 5, 8, 2)
    raw = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(5, 1 + 6 * 3)
    assert (raw[:, 0] == 0).all()
    px = raw[:, 1:].reshape(5, 6, 3)
    assert (px[4, :, 1] == 255).all()                           # our row 0 (bottom) is the LAST PNG row
    assert px[0, 0].tolist() == [0, 0, 255]                     # NaN -> 0, negative -> 0, > 1 -> 255 (top-left = our row 4)
    lin = np.linspace(0, 1, 6)
    srgb = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)
    assert np.abs(px[2, :, 0].astype(int) - np.round(srgb * 255).astype(int)).max() <= 1
    from PIL import Image                                       # and an independent decoder agrees
    assert np.array_equal(np.asarray(Image.open(p)), px)


def present_probe_image():
    """Floats around every step of the 8-bit sRGB encoder (+-300 representable neighbours of each code's first float), the specials
    (NaN, +-inf, +-0, denormals, negatives, huge) and a uniform random fill; (rows, 601, 4) float32, every channel exercised."""
    first = host_io.srgb8_first_floats()
    bits = first[1:].view(np.uint32).astype(np.int64)
    near = (bits[:, None] + np.arange(-300, 301)[None, :]).astype(np.uint32).view(np.float32)           # (255, 601)
    rng = np.random.default_rng(11)
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-45, 1e-40, -1e-40, -1.0, 1.0, np.nextafter(np.float32(1), np.float32(0)), 65504.0, 65520.0,
                        1e30, 6.1e-5, 5.96e-8, 2.98e-8, 2.9802325e-08, 0.5, 1.0 / 510.0, 253.5 / 255.0], np.float32)
    img = np.zeros((255 + 8, 601, 4), np.float32)
    img[:255, :, 0] = near
    img[:255, :, 1] = near[::-1]
    img[:255, :, 2] = np.roll(near, 7, axis=0)
    img[:255, :, 3] = (np.arange(255 * 601, dtype=np.float64).reshape(255, 601) / (255 * 601 - 1)).astype(np.float32)   # alpha ramp: every UNORM8 code
    img[255:] = rng.uniform(-0.25, 1.5, (8, 601, 4)).astype(np.float32)
    img[255, :special.size, :] = special[:, None]
    img[256, :255, 3] = ((np.arange(255) + 0.5) / 255.0).astype(np.float32)                                   # alpha at the rounding ties
    return img


def test_srgb8_encoder_and_its_step_table(built_library, tmp_path):
    """urt_host_encode_srgb8 (the PNG writer's pixel encoding; what RenderTexture.ReadBegin("RGBA8_SRGB") delivers from the GPU) against
    (a) an independent double-precision restatement of the sRGB transfer function — equal except where float pow() lands within
    rounding of a code boundary (at most four of the 601 floats nearest to each boundary, none elsewhere); (b) the PNG writer's
    own bytes; (c) the step table urt_host_srgb8_first_floats: code(x) == number of steps at or below x, for the neighbours of every step."""
    img = present_probe_image()
    got = host_io.encode_srgb8(img)
    x = img[..., :3].astype(np.float64)
    with np.errstate(invalid="ignore"):
        s = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(np.maximum(x, 0), 1 / 2.4) - 0.055)
        want = np.where(np.isnan(x) | (x <= 0), 0, np.where(x >= 1, 255, np.floor(s * 255 + 0.5))).astype(np.int64)
    d = np.abs(got[..., :3].astype(np.int64) - want)
    # float32 evaluation (powf, the multiply-add, x 255) moves a boundary by a few representable floats against the double evaluation
    assert d.max() <= 1 and int((d[:255, :, 0] != 0).sum(axis=1).max()) <= 4, (int(d.max()), int((d[:255, :, 0] != 0).sum(axis=1).max()))
    assert int((d[257:] != 0).sum()) <= 3                       # 10,818 random values: the chance to sit within four floats of a boundary is 1e-4 each
    a = img[..., 3].astype(np.float64)
    with np.errstate(invalid="ignore"):
        v = (img[..., 3] * np.float32(255.0)).astype(np.float64)   # the product is rounded to float32 first (urt_host_encode_srgb8)
        wa = np.where(np.isnan(a) | (a <= 0), 0, np.where(a >= 1, 255, np.floor(v + 0.5))).astype(np.int64)
    assert np.array_equal(got[..., 3].astype(np.int64), wa)
    p = str(tmp_path / "probe.png")
    host_io.write_png(p, img)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(p))[::-1], got[..., :3])     # PNG rows run top to bottom
    first = host_io.srgb8_first_floats()
    assert first[0] == -np.inf and np.all(np.diff(first[1:]) > 0) and first[255] < 1.0
    codes = np.searchsorted(first[1:], img[..., :3], side="right")           # steps at or below x (NaN sorts last: handled below)
    codes = np.where(np.isnan(img[..., :3]), 0, codes)
    assert np.array_equal(codes.astype(np.uint8), got[..., :3])


def test_mitchell_resize_and_sky_import_limit(built_library, tmp_path):
    """urt_host_resize_rgba: constant images stay constant (normalised weights), a linear ramp stays linear away from the edges,
    energy is preserved on a 2:1 downscale; load_sky applies the importer's maxTextureSize (Assets/Skyboxes/*.hdr.meta:36)."""
    const = np.full((12, 20, 4), 0.75, np.float32)
    assert np.allclose(host_io.resize(const, 7, 5), 0.75, atol=1e-6)
    ramp = np.zeros((8, 64, 4), np.float32)
    ramp[..., 0] = np.arange(64, dtype=np.float32)[None, :]
    small = host_io.resize(ramp, 32, 8)
    want = (np.arange(32) * 2 + 0.5).astype(np.float32)                       # centres of the 2-texel cells
    assert np.allclose(small[4, 4:-4, 0], want[4:-4], atol=1e-3)
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 4, (64, 128, 4)).astype(np.float32)
    half = host_io.resize(img, 64, 32)
    assert abs(float(half.mean()) - float(img.mean())) < 0.02
    # a 96x48 "sky" with the limit set to 32: 32x16 comes back, row 0 still the bottom row
    yy = np.linspace(0.1, 3.0, 48, dtype=np.float32)[::-1, None]                # file rows are top first: brightest at the top of the FILE
    rgb = np.repeat(np.repeat(yy[:, :, None], 96, axis=1), 3, axis=2)
    path = str(tmp_path / "big.hdr")
    write_hdr(path, float_to_rgbe(rgb.astype(np.float64)), rle=False)
    sky = host_io.load_sky(path, max_texture_size=32)
    assert sky.shape == (16, 32, 4)
    assert sky[0, 5, 0] < sky[-1, 5, 0]                                        # bottom row (file's last) is the dim one
    assert host_io.load_sky(path, max_texture_size=4096).shape == (48, 96, 4)  # under the limit: untouched
