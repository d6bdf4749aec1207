// valu_rate.hip — how many wave64 VALU instructions per cycle does one SIMD of this chip issue?  (the trace kernel's
// SIMD cycles equal 4 x its VALU instructions; is 4 cycles per instruction the ceiling, or 2?)
// Independent v_fma_f32 / v_min_f32 / v_cndmask chains, W waves per SIMD, no memory traffic.  Prints wave-instructions per cycle per SIMD
// (clock from wall_clock64 at 100 MHz against s_memtime is not needed: we time with HIP events and take the device clock rate).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate scripts/micro/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 1.0000001f, c = 1e-7f;
  for (int i = 0; i < iters; i++) {
#pragma unroll 8
    for (int u = 0; u < 8; u++) {
      if (KIND == 0) {            // 8 independent fma per round
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
      } else if (KIND == 1) {     // min / max mix (the slab test's diet)
        a0 = fminf(a0, a1 + c); a1 = fmaxf(a1, a2 - c); a2 = fminf(a2, a3 + c); a3 = fmaxf(a3, a4 - c);
        a4 = fminf(a4, a5 + c); a5 = fmaxf(a5, a6 - c); a6 = fminf(a6, a7 + c); a7 = fmaxf(a7, a0 - c);
      } else {                    // compare + select (the scheduler's diet)
        a0 = a0 > a1 ? a2 : a3; a1 = a1 > a2 ? a3 : a4; a2 = a2 > a3 ? a4 : a5; a3 = a3 > a4 ? a5 : a6;
        a4 = a4 > a5 ? a6 : a7; a5 = a5 > a6 ? a7 : a0; a6 = a6 > a7 ? a0 : a1; a7 = a7 > a0 ? a1 : a2;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
  int dev = 0; hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
  int cus = p.multiProcessorCount; double mhz = p.clockRate / 1000.0;
  printf("%s: %d CUs, clock %.0f MHz (nominal)\n", p.name, cus, mhz);
  float* out; hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int kind = 0; kind < 3; kind++)
    for (int wps : {1, 2, 4, 5, 8}) {                   // waves per SIMD = workgroups of 256 threads (4 waves = 1 per SIMD) per CU
      int blocks = cus * wps;
      auto launch = [&]() {
        if (kind == 0) hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        else if (kind == 1) hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        else hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double insts_per_wave = (double)iters * 64.0 * (kind == 1 ? 2.0 : kind == 2 ? 2.0 : 1.0);   // kind 1: add + min/max, kind 2: cmp + cndmask per statement
      double wave_insts_per_simd = insts_per_wave * wps;
      double cycles = ms * 1e-3 * mhz * 1e6;
      printf("kind %d (%s), %d waves/SIMD: %.3f ms, %.3f wave-instructions per cycle per SIMD (nominal clock) = one per %.2f cycles\n", kind,
             kind == 0 ? "fma" : kind == 1 ? "add+min/max" : "cmp+cndmask", wps, ms, wave_insts_per_simd / cycles, cycles / wave_insts_per_simd);
    }
  return 0;
}
