// qnodes.hip — the traversal's derived copies of the triangle-BVH nodes: centre / half-extent boxes (k_center: what every trace kernel reads)
// and, optionally, 32-byte quantized nodes for the traversal loop of the default trace kernel.
//
// Measured (profiles/r03_logs/r3_ab_vote_xload.log): the loop is sensitive to the number of vector-memory instructions per node
// step — two extra dwordx4 loads per step cost +14..19 % frame time.  A 64-byte node (two child boxes as 12 floats + two child
// codes) takes four; the form built here takes two:
//     dwords 0-2: child 0  lo.x | lo.y << 16,  lo.z | hi.x << 16,  hi.y | hi.z << 16      (16-bit grid coordinates)
//     dwords 3-5: child 1, the same            dwords 6-7: the two child codes, unchanged
// The grid is ONE frame for the whole forest: origin = the lower corner of the union of all root boxes, cell = extent / 65531 per
// axis.  A box is rounded OUTWARD and widened by two more cells, which covers the rounding of the traversal's own arithmetic
// (kernels.hip qnode_eval_flat: t = fma(2^23 + q, cell / d, B) with B folding origin, ray and the 2^23 offset: error <= 0.5 cell).
// Quantized boxes only CULL — conservatively; the Moller-Trumbore tests decide the hits — so pixels are the float nodes' pixels;
// visit counts differ slightly (the counting instantiation of the kernel walks the float nodes, like the oracle).
// qbuf[0] = origin.xyz, quality (smallest MeshObject extent in cells);  qbuf[1] = cell.xyz, 0;  node n at qbuf[2 + 2 n].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/urt_math.h"
#include "qnodes.h"

using namespace urt;

namespace {

__device__ __forceinline__ int as_i(float f) { return __builtin_bit_cast(int, f); }
__device__ __forceinline__ float as_f(unsigned int i) { return __builtin_bit_cast(float, i); }

// one workgroup: union of the root boxes of every MeshObject whose root is an interior node -> the frame
__global__ __launch_bounds__(256) void k_qframe(const float4* __restrict__ nodes, int n_nodes, const int32_t* __restrict__ mesh_root, int n_meshes,
                                                float4* __restrict__ qbuf) {
  __shared__ float slo[3][256], shi[3][256];
  float lo[3] = {URT_INF, URT_INF, URT_INF}, hi[3] = {-URT_INF, -URT_INF, -URT_INF};
  for (int m = threadIdx.x; m < n_meshes; m += 256) {
    int r = mesh_root[m];
    if (r < 0 || r >= n_nodes) continue;                     // a single leaf or empty: no nodes
    float4 q0 = nodes[4 * (size_t)r], q1 = nodes[4 * (size_t)r + 1], q2 = nodes[4 * (size_t)r + 2];
    const float l[2][3] = {{q0.x, q0.y, q0.z}, {q1.z, q1.w, q2.x}}, h[2][3] = {{q0.w, q1.x, q1.y}, {q2.y, q2.z, q2.w}};
    for (int c = 0; c < 2; c++)
      for (int k = 0; k < 3; k++) {
        if (l[c][k] <= h[c][k] && f_abs(l[c][k]) < URT_INF && f_abs(h[c][k]) < URT_INF) { lo[k] = f_min(lo[k], l[c][k]); hi[k] = f_max(hi[k], h[c][k]); }
      }
  }
  for (int k = 0; k < 3; k++) { slo[k][threadIdx.x] = lo[k]; shi[k][threadIdx.x] = hi[k]; }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int k = 0; k < 3; k++) { slo[k][threadIdx.x] = f_min(slo[k][threadIdx.x], slo[k][threadIdx.x + s]); shi[k][threadIdx.x] = f_max(shi[k][threadIdx.x], shi[k][threadIdx.x + s]); }
    __syncthreads();
  }
  __shared__ float cell[3], org[3];
  if (threadIdx.x == 0) {
    for (int k = 0; k < 3; k++) {
      float a = slo[k][0], b = shi[k][0];
      if (!(a <= b)) { a = 0.0f; b = 0.0f; }
      float ext = b - a;
      org[k] = a;
      cell[k] = f_max(ext * (1.0f / 65531.0f), 1e-30f) * 1.0000002f;      // (a hair more than extent / 65531: the top plane lands below 65533)
    }
    qbuf[1] = make_float4(cell[0], cell[1], cell[2], 0.0f);
  }
  __syncthreads();
  // quality: the smallest MeshObject, in cells along its longest axis (a mesh of a few cells would be walked almost exhaustively)
  float q = URT_INF;
  for (int m = threadIdx.x; m < n_meshes; m += 256) {
    int r = mesh_root[m];
    if (r < 0 || r >= n_nodes) continue;
    float4 q0 = nodes[4 * (size_t)r], q1 = nodes[4 * (size_t)r + 1], q2 = nodes[4 * (size_t)r + 2];
    float ex = f_max(q0.w, q2.y) - f_min(q0.x, q1.z), ey = f_max(q1.x, q2.z) - f_min(q0.y, q1.w), ez = f_max(q1.y, q2.w) - f_min(q0.z, q2.x);
    float cells = f_max(f_max(ex / cell[0], ey / cell[1]), ez / cell[2]);
    if (cells == cells) q = f_min(q, cells);
  }
  slo[0][threadIdx.x] = q;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) slo[0][threadIdx.x] = f_min(slo[0][threadIdx.x], slo[0][threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) qbuf[0] = make_float4(org[0], org[1], org[2], slo[0][0]);
}

__device__ __forceinline__ unsigned int q_lo(float x, float org, float inv) {      // outward, two cells of margin, into [0, 65535]
  float g = f_floor((x - org) * inv) - 2.0f;
  if (!(g >= 0.0f)) g = 0.0f;                                 // (NaN -> 0: conservative)
  if (g > 65535.0f) g = 65535.0f;
  return (unsigned int)g;
}
__device__ __forceinline__ unsigned int q_hi(float x, float org, float inv) {
  float g = f_floor((x - org) * inv) + 3.0f;                  // ceil + 2, without a ceil
  if (!(g <= 65535.0f)) g = 65535.0f;                         // (NaN -> 65535)
  if (g < 0.0f) g = 0.0f;
  return (unsigned int)g;
}

__global__ __launch_bounds__(256) void k_quantize(const float4* __restrict__ nodes, int n_nodes, float4* __restrict__ qbuf) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_nodes) return;
  const float4 f0 = qbuf[0], f1 = qbuf[1];
  const float org[3] = {f0.x, f0.y, f0.z}, inv[3] = {1.0f / f1.x, 1.0f / f1.y, 1.0f / f1.z};
  float4 q0 = nodes[4 * (size_t)n], q1 = nodes[4 * (size_t)n + 1], q2 = nodes[4 * (size_t)n + 2], q3 = nodes[4 * (size_t)n + 3];
  const float l[2][3] = {{q0.x, q0.y, q0.z}, {q1.z, q1.w, q2.x}}, h[2][3] = {{q0.w, q1.x, q1.y}, {q2.y, q2.z, q2.w}};
  unsigned int w[6];
  for (int c = 0; c < 2; c++) {
    unsigned int a[3], b[3];
    bool empty = false;
    for (int k = 0; k < 3; k++) { a[k] = q_lo(l[c][k], org[k], inv[k]); b[k] = q_hi(h[c][k], org[k], inv[k]); empty = empty || l[c][k] > h[c][k]; }
    if (empty) { for (int k = 0; k < 3; k++) { a[k] = 65535u; b[k] = 0u; } }      // an inverted box stays inverted: never entered
    w[3 * c] = a[0] | (a[1] << 16); w[3 * c + 1] = a[2] | (b[0] << 16); w[3 * c + 2] = b[1] | (b[2] << 16);
  }
  qbuf[2 + 2 * (size_t)n] = make_float4(as_f(w[0]), as_f(w[1]), as_f(w[2]), as_f(w[3]));
  qbuf[2 + 2 * (size_t)n + 1] = make_float4(as_f(w[4]), as_f(w[5]), q3.x, q3.y);
}

// The traversal's copy of the nodes: every child box as centre / half extent (include/urt_math.h box_center_form), child codes unchanged.
//   q0 = c0.xyz, h0.x   q1 = h0.yz, c1.xy   q2 = c1.z, h1.xyz   q3 = child0, child1, 0, 0
__global__ __launch_bounds__(256) void k_center(const float4* __restrict__ nodes, int n_nodes, float4* __restrict__ cnodes) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_nodes) return;
  float4 q0 = nodes[4 * (size_t)n], q1 = nodes[4 * (size_t)n + 1], q2 = nodes[4 * (size_t)n + 2], q3 = nodes[4 * (size_t)n + 3];
  const float l0[3] = {q0.x, q0.y, q0.z}, h0[3] = {q0.w, q1.x, q1.y}, l1[3] = {q1.z, q1.w, q2.x}, h1[3] = {q2.y, q2.z, q2.w};
  float c0[3], e0[3], c1[3], e1[3];
  box_center_form(l0, h0, c0, e0);
  box_center_form(l1, h1, c1, e1);
  cnodes[4 * (size_t)n] = make_float4(c0[0], c0[1], c0[2], e0[0]);
  cnodes[4 * (size_t)n + 1] = make_float4(e0[1], e0[2], c1[0], c1[1]);
  cnodes[4 * (size_t)n + 2] = make_float4(c1[2], e1[0], e1[1], e1[2]);
  cnodes[4 * (size_t)n + 3] = q3;
}

}  // namespace

namespace urtd {

hipError_t center_nodes(const float4* nodes, int n_nodes, float4* cnodes, hipStream_t st) {
  if (n_nodes <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_center, dim3((n_nodes + 255) / 256), dim3(256), 0, st, nodes, n_nodes, cnodes);
  return hipGetLastError();
}

hipError_t quantize_nodes(const float4* nodes, int n_nodes, const int32_t* mesh_root, int n_meshes, float4* qbuf, hipStream_t st) {
  if (n_nodes <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_qframe, dim3(1), dim3(256), 0, st, nodes, n_nodes, mesh_root, n_meshes, qbuf);
  hipLaunchKernelGGL(k_quantize, dim3((n_nodes + 255) / 256), dim3(256), 0, st, nodes, n_nodes, qbuf);
  return hipGetLastError();
}

}  // namespace urtd
