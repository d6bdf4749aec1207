// present.hip — the last hop of RM:819 `Graphics.Blit(_converged, destination)` for a host whose `destination` lives in host memory or
// on another device: the image is converted to the destination's format ON THE GPU, so that 8 MB (RGBA8) or 17 MB (RGBA16F) cross the
// PCIe bus per 1080p frame instead of 33 MB of RGBA32F.
//
//   RGBA8 sRGB   the project renders in linear colour space (ProjectSettings/ProjectSettings.asset:50 m_ActiveColorSpace: 1), so the blit
//                into an 8-bit back buffer applies the sRGB transfer function: colour = the code of csrc/host_io.cpp's PNG writer (what
//                RM:762's screenshot holds), alpha = UNORM8 (linear).  The code is a monotone step function of the float, so the kernel
//                does not evaluate pow(): it counts, by binary search in a 256-entry table the host derived from ITS encoder, how many
//                codes' first floats lie at or below the value — the same bytes as the host encoder for every float, by construction
//                (tests/test_gpu_present.py sweeps the neighbourhood of every step).
//   RGBA16F      IEEE round-to-nearest-even conversion (v_cvt_f16_f32; a camera with allowHDR renders into ARGBHalf).
//
// HBM-bound: 16 B read + 4 / 8 B written per pixel, one thread per pixel, coalesced dwordx4 loads.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "present.h"

namespace urtd {

namespace {

__device__ __forceinline__ unsigned int srgb_code(const float* __restrict__ first, float x) {
  // number of k in [1, 255] with first[k] <= x  (first[] ascending; NaN compares false everywhere -> 0, like the host encoder)
  unsigned int lo = 0;
#pragma unroll
  for (unsigned int step = 128; step > 0; step >>= 1)
    if (first[lo + step] <= x) lo += step;
  return lo;
}

__device__ __forceinline__ unsigned int unorm8(float a) {
  if (!(a > 0.0f)) return 0;
  if (a >= 1.0f) return 255;
  float v = a * 255.0f, f = floorf(v);                // lround: halves away from zero; v - f is exact
  return (unsigned int)(v - f >= 0.5f ? f + 1.0f : f);
}

__global__ __launch_bounds__(256) void k_encode_srgb8(const float4* __restrict__ src, uint32_t* __restrict__ dst, size_t pixels, const float* __restrict__ first_g) {
  static_assert(kSrgbCodes == 256, "one table entry per thread of the 256-thread block");
  __shared__ float first[kSrgbCodes];
  first[threadIdx.x] = first_g[threadIdx.x];
  __syncthreads();
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels) return;
  float4 p = src[i];
  dst[i] = srgb_code(first, p.x) | (srgb_code(first, p.y) << 8) | (srgb_code(first, p.z) << 16) | (unorm8(p.w) << 24);
}

__global__ __launch_bounds__(256) void k_encode_half(const float4* __restrict__ src, uint2* __restrict__ dst, size_t pixels) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels) return;
  float4 p = src[i];
  uint2 o;
  o.x = (unsigned int)__half_as_ushort(__float2half_rn(p.x)) | ((unsigned int)__half_as_ushort(__float2half_rn(p.y)) << 16);
  o.y = (unsigned int)__half_as_ushort(__float2half_rn(p.z)) | ((unsigned int)__half_as_ushort(__float2half_rn(p.w)) << 16);
  dst[i] = o;
}

}  // namespace

size_t format_pixel_bytes(int format) {
  return format == kFormatRGBA32F ? 16 : format == kFormatRGBA8sRGB ? 4 : format == kFormatRGBA16F ? 8 : 0;
}

hipError_t launch_encode(const float4* src, void* dst, size_t pixels, int format, const float* srgb_first, hipStream_t st) {
  if (pixels == 0) return hipSuccess;
  if (pixels > ((size_t)1 << 31) * 256) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((pixels + 255) / 256)), block(256);
  if (format == kFormatRGBA8sRGB) {
    if (!srgb_first) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_encode_srgb8, grid, block, 0, st, src, (uint32_t*)dst, pixels, srgb_first);
  } else if (format == kFormatRGBA16F) {
    hipLaunchKernelGGL(k_encode_half, grid, block, 0, st, src, (uint2*)dst, pixels);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace urtd
