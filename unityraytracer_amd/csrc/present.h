// present.h — encoding of an RGBA32F image into the format of a host's `destination` (csrc/present.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace urtd {

// formats of urt_texture_read_begin_format (include/urt.h URT_FORMAT_*)
static constexpr int kFormatRGBA32F = 0, kFormatRGBA8sRGB = 1, kFormatRGBA16F = 2;
static constexpr int kSrgbCodes = 256;

// bytes per pixel of a format, 0 = unknown format
size_t format_pixel_bytes(int format);
// src (pixels x float4) -> dst in `format`; srgb_first = device table of kSrgbCodes floats: srgb_first[k] = the smallest float whose
// 8-bit sRGB code (csrc/host_io.cpp urt_host_encode_srgb8) is >= k  (needed for kFormatRGBA8sRGB only)
hipError_t launch_encode(const float4* src, void* dst, size_t pixels, int format, const float* srgb_first, hipStream_t st);

}  // namespace urtd
