// valu_table.hip — issue cost of individual wave64 VALU instructions on this chip (SIMD cycles per instruction, all SIMDs loaded with
// W waves each; independent chains, no memory traffic).  Complements valu_rate.hip (compiler-generated mixes): here every kind is ONE
// instruction, written in inline assembly, eight copies per round.  The copies form ONE dependent chain per wave (each reads the
// previous result): the 1-wave column is the instruction's latency, the 8-wave column its issue cost (latency hidden by the other waves).
//   hipcc --offload-arch=gfx950 -O3 -o valu_table scripts/micro/valu_table.hip && ./valu_table
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(S) S S S S S S S S

enum Kind { K_FMA, K_MUL, K_ADD, K_MAX, K_MIN3, K_MED3, K_PK_FMA, K_PK_MUL, K_CMP_VCC, K_CMP_SGPR, K_CND_VCC, K_CND_SGPR, K_CMP_CND, K_AND, K_LSHL_ADD, K_ADD3, K_XOR3,
            K_CVT_I2F, K_RCP, K_SQRT, K_SIN, K_EXP, K_LOG, K_MUL_LO, K_MAD_U24, K_MOV, K_READLANE, K_BFE, K_CMP_CLASS, K_MAX_I32, K_MIN_MAX_PAIR, K_COUNT };
static const char* kNames[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_min3_f32", "v_med3_f32", "v_pk_fma_f32", "v_pk_mul_f32",
                               "v_cmp_gt_f32 vcc", "v_cmp_gt_f32 sgpr", "v_cndmask_b32 (vcc)", "v_cndmask_b32 (sgpr)", "v_cmp + v_cndmask (pair, per instr)", "v_and_b32", "v_lshl_add_u32",
                               "v_add3_u32", "v_xor3 / v_bitop3", "v_cvt_f32_i32", "v_rcp_f32", "v_sqrt_f32", "v_sin_f32", "v_exp_f32", "v_log_f32", "v_mul_lo_u32", "v_mad_u32_u24",
                               "v_mov_b32", "v_readlane_b32", "v_bfe_u32", "v_cmp_class_f32", "v_max_i32", "v_min_f32 + v_max_f32 (per instr)"};

template <int KIND>
__global__ __launch_bounds__(256) void k_one(float* out, int iters, float seed) {
  float a = seed + threadIdx.x, b = a + 1.5f, c = 1e-7f;
  double pa = (double)a, pb = 1.0000001, pc = 1e-9;
  int ia = (int)threadIdx.x, ib = 7;
  int s = 0;
  for (int i = 0; i < iters; i++) {
    if (KIND == K_FMA) asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
    if (KIND == K_MUL) asm volatile(REP8("v_mul_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
    if (KIND == K_ADD) asm volatile(REP8("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(c));
    if (KIND == K_MAX) asm volatile(REP8("v_max_f32 %0, %0, %1\n") : "+v"(a) : "v"(c));
    if (KIND == K_MIN3) asm volatile(REP8("v_min3_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
    if (KIND == K_MED3) asm volatile(REP8("v_med3_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
    if (KIND == K_PK_FMA) asm volatile(REP8("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(pa) : "v"(pb), "v"(pc));
    if (KIND == K_PK_MUL) asm volatile(REP8("v_pk_mul_f32 %0, %0, %1\n") : "+v"(pa) : "v"(pb));
    if (KIND == K_CMP_VCC) asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %1\n") : : "v"(a), "v"(b) : "vcc");
    if (KIND == K_CMP_SGPR) asm volatile(REP8("v_cmp_gt_f32 s[20:21], %0, %1\n") : : "v"(a), "v"(b) : "s20", "s21");
    if (KIND == K_CND_VCC) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n" REP8("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(b), "v"(c) : "vcc");
    if (KIND == K_CND_SGPR) asm volatile("v_cmp_gt_f32 s[20:21], %1, %2\n s_nop 4\n" REP8("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n") : "+v"(a) : "v"(b), "v"(c) : "s20", "s21");
    if (KIND == K_CMP_CND) asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %1, vcc\n") : "+v"(a) : "v"(b), "v"(c) : "vcc");
    if (KIND == K_AND) asm volatile(REP8("v_and_b32 %0, %0, %1\n") : "+v"(ia) : "v"(ib));
    if (KIND == K_LSHL_ADD) asm volatile(REP8("v_lshl_add_u32 %0, %0, 1, %1\n") : "+v"(ia) : "v"(ib));
    if (KIND == K_ADD3) asm volatile(REP8("v_add3_u32 %0, %0, %1, %1\n") : "+v"(ia) : "v"(ib));
    if (KIND == K_XOR3) asm volatile(REP8("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96\n") : "+v"(ia) : "v"(ib));
    if (KIND == K_CVT_I2F) asm volatile(REP8("v_cvt_f32_i32 %0, %1\n") : "+v"(a) : "v"(ia));
    if (KIND == K_RCP) asm volatile(REP8("v_rcp_f32 %0, %0\n") : "+v"(a));
    if (KIND == K_SQRT) asm volatile(REP8("v_sqrt_f32 %0, %0\n") : "+v"(a));
    if (KIND == K_SIN) asm volatile(REP8("v_sin_f32 %0, %0\n") : "+v"(a));
    if (KIND == K_EXP) asm volatile(REP8("v_exp_f32 %0, %0\n") : "+v"(a));
    if (KIND == K_LOG) asm volatile(REP8("v_log_f32 %0, %0\n") : "+v"(a));
    if (KIND == K_MUL_LO) asm volatile(REP8("v_mul_lo_u32 %0, %0, %1\n") : "+v"(ia) : "v"(ib));
    if (KIND == K_MAD_U24) asm volatile(REP8("v_mad_u32_u24 %0, %0, %1, %1\n") : "+v"(ia) : "v"(ib));
    if (KIND == K_MOV) asm volatile(REP8("v_mov_b32 %0, %1\n") : "+v"(ia) : "v"(ib));
    if (KIND == K_READLANE) asm volatile(REP8("v_readlane_b32 %0, %1, 3\n") : "+s"(s) : "v"(ia));
    if (KIND == K_BFE) asm volatile(REP8("v_bfe_u32 %0, %0, 1, 20\n") : "+v"(ia));
    if (KIND == K_CMP_CLASS) asm volatile(REP8("v_cmp_class_f32 vcc, %0, %1\n") : : "v"(a), "v"(ib) : "vcc");
    if (KIND == K_MAX_I32) asm volatile(REP8("v_max_i32 %0, %0, %1\n") : "+v"(ia) : "v"(ib));
    if (KIND == K_MIN_MAX_PAIR) asm volatile(REP8("v_min_f32 %0, %0, %1\n v_max_f32 %0, %0, %2\n") : "+v"(a) : "v"(b), "v"(c));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + (float)pa + (float)ia + (float)s;
}

template <int KIND>
void run(float* out, int cus, double mhz, hipEvent_t e0, hipEvent_t e1) {
  if constexpr (KIND < K_COUNT) {
    const int iters = 20000;
    printf("%-38s", kNames[KIND]);
    for (int wps : {1, 2, 5, 8}) {
      int blocks = cus * wps;
      hipLaunchKernelGGL(k_one<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_one<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double per_round = (KIND == K_CMP_CND || KIND == K_MIN_MAX_PAIR) ? 16.0 : 8.0;
      double wave_insts_per_simd = (double)iters * per_round * wps;
      double cycles = ms * 1e-3 * mhz * 1e6;
      printf("  %d w/SIMD: %5.2f", wps, cycles / wave_insts_per_simd);
    }
    printf("   cycles per instruction\n");
    run<KIND + 1>(out, cus, mhz, e0, e1);
  }
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount; double mhz = p.clockRate / 1000.0;
  printf("%s: %d CUs, clock %.0f MHz (nominal); SIMD cycles per wave64 instruction\n", p.name, cus, mhz);
  float* out; hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  run<0>(out, cus, mhz, e0, e1);
  return 0;
}
