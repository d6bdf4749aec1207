// cullflags.hip — verification pass behind the object-level cull (include/urt_math.h tlas_cull; kernels.hip front_masked / trace_front).
//
// The reference intersects every MeshObject whose heap leaf is popped after the first hit leaf box (RS:294-326, `tests` never reset) —
// also objects whose box the ray misses.  Skipping such an object is only sound if the leaf's box (the SCENE's data: `_MeshBVH`, RM:148-152,
// which the library does not build) really contains the object's triangles.  So after every build / refit of the triangle records this
// pass checks exactly that on the GPU, one thread per triangle, and clears the cull word of every leaf that fails; leaves of heaps
// made by the reference's own builder (tight or literal bounds of the transformed vertices, RM:405-457) pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cullflags.h"

namespace urtd {

namespace {

__global__ __launch_bounds__(256) void k_cull_check(float4* __restrict__ mesh_tlas, int n_nodes, const int32_t* __restrict__ mesh_leaf, int n_meshes,
                                                    const float4* __restrict__ tri_verts, int n_tris) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_tris) return;
  float4 r0 = tri_verts[3 * (size_t)k], r1 = tri_verts[3 * (size_t)k + 1], r2 = tri_verts[3 * (size_t)k + 2];
  int m = __builtin_bit_cast(int, r1.w);
  if (m < 0 || m >= n_meshes) return;
  int node = mesh_leaf[m];
  if (node < 0 || node >= n_nodes) return;
  float4 a = mesh_tlas[2 * node], b = mesh_tlas[2 * node + 1];
  float M = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fabsf(a.z)), fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fabsf(b.z)));
  float tol = M * 9.5367431640625e-7f;                                    // 2^-20
  float px[3] = {r0.x, r0.x + r1.x, r0.x + r2.x}, py[3] = {r0.y, r0.y + r1.y, r0.y + r2.y}, pz[3] = {r0.z, r0.z + r1.z, r0.z + r2.z};
  bool inside = true;
  for (int v = 0; v < 3; v++)
    inside = inside && px[v] >= a.x - tol && px[v] <= b.x + tol && py[v] >= a.y - tol && py[v] <= b.y + tol && pz[v] >= a.z - tol && pz[v] <= b.z + tol;   // (NaN: not inside)
  if (!inside) ((int*)mesh_tlas)[(2 * node + 1) * 4 + 3] = 0;            // benign race: every writer stores 0
}

__global__ void k_cull_mask(const float4* __restrict__ mesh_tlas, int n_nodes, int* __restrict__ walk_cull_mask) {
  int i = threadIdx.x;
  int w = i < n_nodes ? __builtin_bit_cast(int, mesh_tlas[2 * i + 1].w) : 0;
  for (int off = 32; off > 0; off >>= 1) w |= __shfl_xor(w, off, 64);
  if (i == 0) *walk_cull_mask = w;
}

}  // namespace

hipError_t update_cull_flags(float4* mesh_tlas, int n_nodes, const int32_t* mesh_leaf, int n_meshes, const float4* tri_verts, int n_tris,
                             int* walk_cull_mask, hipStream_t st) {
  if (n_nodes <= 0 || n_meshes <= 0) return hipSuccess;
  if (n_tris > 0) {
    hipLaunchKernelGGL(k_cull_check, dim3((unsigned)((n_tris + 255) / 256)), dim3(256), 0, st, mesh_tlas, n_nodes, mesh_leaf, n_meshes, tri_verts, n_tris);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (walk_cull_mask) {
    if (n_nodes > 64) return hipErrorInvalidValue;                         // (a walk table exists for heaps of <= 31 nodes only)
    hipLaunchKernelGGL(k_cull_mask, dim3(1), dim3(64), 0, st, (const float4*)mesh_tlas, n_nodes, walk_cull_mask);
    return hipGetLastError();
  }
  return hipSuccess;
}

}  // namespace urtd
