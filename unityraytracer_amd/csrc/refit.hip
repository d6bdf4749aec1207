// refit.hip — GPU refit of the triangle BVHs of MOVED MeshObjects (dynamic scenes).
//
// The reference's protocol for a moving object is "re-upload everything" (RayTraceMaster.cs:215-230 -> 262-336: any
// Register/Unregister or transform change rebuilds all lists and calls SetData on all seven buffers).  When all that changed is the
// localToWorldMatrix of some MeshObjects, their triangle BVHs keep their TOPOLOGY (a good SAH tree stays a good tree under a rigid
// or affine motion): only the world-space triangle records (RS:244-246 applied once per upload) and the boxes have
// to follow.  Both are recomputed here, in place, on the GPU, from device-resident copies of _Vertices / _Indices:
//   k_refit_tris   one lane per leaf-order triangle of a moved MeshObject: the normative mul(localToWorld, float4(v, 1)) of its three
//                  vertices with the NEW matrix -> the 48-byte record (v0 | slot, e1 | mesh, e2 | 0) — bit-identical to what a full
//                  rebuild writes (blas_builder.cpp `records`) — and the MeshObject's largest |coordinate| (the box pad's scale);
//   k_refit_level  bottom-up, one launch per tree level: a node's two child boxes from the new records (leaf children) or from the
//                  level below (interior children), written with the builder's pad of 2^-16 x largest |coordinate| (blas_builder.cpp).
// Which nodes belong to which MeshObject, every node's parent and its depth are derived once per full scene preparation (k_parents,
// k_node_mesh, k_depth) from the node array itself, so the refit works on trees of either builder (host SAH, GPU LBVH).
// Pixels do not depend on the boxes (they only cull; the Moller-Trumbore test decides): tests/test_gpu_refit.py compares moved
// scenes with a full rebuild and with the oracle bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/urt_math.h"
#include "refit.h"

using namespace urt;

namespace {

__device__ __forceinline__ int as_i(float f) { return __builtin_bit_cast(int, f); }
__device__ __forceinline__ float as_f(int i) { return __builtin_bit_cast(float, i); }

__global__ __launch_bounds__(256) void k_parents(const float4* __restrict__ nodes, int n_nodes, int32_t* __restrict__ parent) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_nodes) return;
  float4 q3 = nodes[4 * (size_t)n + 3];
  int c0 = as_i(q3.x), c1 = as_i(q3.y);
  if (c0 >= 0 && c0 < n_nodes) parent[c0] = n;
  if (c1 >= 0 && c1 < n_nodes) parent[c1] = n;
}

// every node's MeshObject: a node with a leaf child reads it from that leaf's first triangle and hands it up its ancestor chain
// (all ancestors belong to the same MeshObject; concurrent walkers write the same value, a walker stops where another has been)
__global__ __launch_bounds__(256) void k_node_mesh(const float4* __restrict__ nodes, const float4* __restrict__ tri_verts, int n_nodes,
                                                   const int32_t* __restrict__ parent, int32_t* node_mesh) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_nodes) return;
  float4 q3 = nodes[4 * (size_t)n + 3];
  int c0 = as_i(q3.x), c1 = as_i(q3.y);
  int leaf = c0 < 0 ? c0 : c1;
  if (leaf >= 0) return;
  uint32_t first = (~(uint32_t)leaf) >> 3;
  int m = as_i(tri_verts[3 * (size_t)first + 1].w);
  for (int p = n; p >= 0; p = parent[p]) {
    if (__hip_atomic_load(&node_mesh[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == m) break;
    __hip_atomic_store(&node_mesh[p], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void k_refit_tris(float4* __restrict__ tri_verts, int n_tris, const float* __restrict__ vertices,
                                                    const int32_t* __restrict__ indices, const float* __restrict__ matrices,
                                                    const int32_t* __restrict__ moved, unsigned int* __restrict__ ext) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = k < n_tris;
  float4 r0 = make_float4(0, 0, 0, 0), r1 = r0;
  if (in_range) { r0 = tri_verts[3 * (size_t)k]; r1 = tri_verts[3 * (size_t)k + 1]; }
  const int slot = as_i(r0.w), m = in_range ? as_i(r1.w) : -1;
  const bool mine = in_range && moved[m];
  const float* M = matrices + 16 * (size_t)(mine ? m : 0);
  v3 w[3];
  float e = 0.0f;
  if (mine) for (int j = 0; j < 3; j++) {
    const float* v = vertices + 3 * (size_t)indices[slot + j];
    w[j] = mul_m4(M, v[0], v[1], v[2], 1.0f);                                  // RS:244-246
    float a = f_abs(w[j].x); if (a < URT_INF) e = f_max(e, a);                // finite coordinates only, like the builders
    a = f_abs(w[j].y); if (a < URT_INF) e = f_max(e, a);
    a = f_abs(w[j].z); if (a < URT_INF) e = f_max(e, a);
  }
  if (mine) {
    v3 e1 = w[1] - w[0], e2 = w[2] - w[0];                                      // RS:201-202
    tri_verts[3 * (size_t)k] = make_float4(w[0].x, w[0].y, w[0].z, r0.w);
    tri_verts[3 * (size_t)k + 1] = make_float4(e1.x, e1.y, e1.z, r1.w);
    tri_verts[3 * (size_t)k + 2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
  }
  // largest |coordinate| per MeshObject (non-negative floats order like their bit patterns).  Leaf-order neighbours nearly always
  // belong to one MeshObject: one atomic per wave then (a million same-address atomics would serialise for ~10 ms)
  unsigned int eb = __float_as_uint(e);
  const int m0 = __shfl(m, 0, 64);
  if (__ballot(m != m0) == 0) {
    for (int off = 32; off > 0; off >>= 1) eb = max(eb, (unsigned int)__shfl_xor((int)eb, off, 64));
    if ((threadIdx.x & 63) == 0 && m0 >= 0 && moved[m0]) atomicMax(&ext[m0], eb);
  } else if (mine) atomicMax(&ext[m], eb);
}

struct Box { float lo[3], hi[3]; };

__device__ __forceinline__ Box leaf_box(const float4* __restrict__ tri_verts, int32_t code) {
  uint32_t c = ~(uint32_t)code;
  uint32_t first = c >> 3, cnt = (c & 7u) + 1u;
  Box b;
  for (int k = 0; k < 3; k++) { b.lo[k] = URT_INF; b.hi[k] = -URT_INF; }
  for (uint32_t t = 0; t < cnt; t++) {
    float4 r0 = tri_verts[3 * (size_t)(first + t)], r1 = tri_verts[3 * (size_t)(first + t) + 1], r2 = tri_verts[3 * (size_t)(first + t) + 2];
    const float p[3][3] = {{r0.x, r0.y, r0.z}, {r0.x + r1.x, r0.y + r1.y, r0.z + r1.z}, {r0.x + r2.x, r0.y + r2.y, r0.z + r2.z}};
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) { b.lo[k] = f_min(b.lo[k], p[j][k]); b.hi[k] = f_max(b.hi[k], p[j][k]); }
  }
  return b;
}

// depth of every node below its root (roots 0): the refit sweeps the levels bottom-up, one launch per level
__global__ __launch_bounds__(256) void k_depth(int n_nodes, const int32_t* __restrict__ parent, int32_t* __restrict__ depth) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_nodes) return;
  int d = 0;
  for (int p = parent[n]; p >= 0; p = parent[p]) d++;
  depth[n] = d;
}

// One level of the bottom-up refit.  cbox: per node 4 float4 = {child0.lo, child0.hi, child1.lo, child1.hi}, UNPADDED (the pad is added
// where a box is written into a node, as blas_builder.cpp does: a box is the union of its triangles, never of padded boxes).  A leaf
// child's box comes from its triangle records, an interior child's from the cbox entries the level below has just written.
__global__ __launch_bounds__(256) void k_refit_level(float4* __restrict__ nodes, int n_nodes, const float4* __restrict__ tri_verts,
                                                     const int32_t* __restrict__ depth, int level, const int32_t* __restrict__ node_mesh,
                                                     const int32_t* __restrict__ moved, const unsigned int* __restrict__ ext, float4* cbox) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_nodes || depth[n] != level) return;
  const int m = node_mesh[n];
  if (m < 0 || !moved[m]) return;
  const float pad = __uint_as_float(ext[m]) * 1.52587890625e-5f + 1e-30f;       // blas_builder.cpp: 2^-16 x largest |coordinate| of the MeshObject
  float4 q3 = nodes[4 * (size_t)n + 3];
  const int c[2] = {as_i(q3.x), as_i(q3.y)};
  float4 lo[2], hi[2];
  for (int k = 0; k < 2; k++) {
    if (c[k] < 0) {
      Box b = leaf_box(tri_verts, c[k]);
      lo[k] = make_float4(b.lo[0], b.lo[1], b.lo[2], 0.0f); hi[k] = make_float4(b.hi[0], b.hi[1], b.hi[2], 0.0f);
    } else {
      float4 l0 = cbox[4 * (size_t)c[k]], h0 = cbox[4 * (size_t)c[k] + 1], l1 = cbox[4 * (size_t)c[k] + 2], h1 = cbox[4 * (size_t)c[k] + 3];
      lo[k] = make_float4(f_min(l0.x, l1.x), f_min(l0.y, l1.y), f_min(l0.z, l1.z), 0.0f);
      hi[k] = make_float4(f_max(h0.x, h1.x), f_max(h0.y, h1.y), f_max(h0.z, h1.z), 0.0f);
    }
    cbox[4 * (size_t)n + 2 * k] = lo[k];
    cbox[4 * (size_t)n + 2 * k + 1] = hi[k];
  }
  nodes[4 * (size_t)n] = make_float4(lo[0].x - pad, lo[0].y - pad, lo[0].z - pad, hi[0].x + pad);
  nodes[4 * (size_t)n + 1] = make_float4(hi[0].y + pad, hi[0].z + pad, lo[1].x - pad, lo[1].y - pad);
  nodes[4 * (size_t)n + 2] = make_float4(lo[1].z - pad, hi[1].x + pad, hi[1].y + pad, hi[1].z + pad);
}

}  // namespace

namespace urtd {

hipError_t refit_prepare(const float4* nodes, int n_nodes, const float4* tri_verts, int32_t* parent, int32_t* node_mesh, int32_t* depth, hipStream_t st) {
  if (n_nodes <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(parent, 0xff, sizeof(int32_t) * (size_t)n_nodes, st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(node_mesh, 0xff, sizeof(int32_t) * (size_t)n_nodes, st);
  if (e != hipSuccess) return e;
  int nb = (n_nodes + 255) / 256;
  hipLaunchKernelGGL(k_parents, dim3(nb), dim3(256), 0, st, nodes, n_nodes, parent);
  hipLaunchKernelGGL(k_node_mesh, dim3(nb), dim3(256), 0, st, nodes, tri_verts, n_nodes, (const int32_t*)parent, node_mesh);
  hipLaunchKernelGGL(k_depth, dim3(nb), dim3(256), 0, st, n_nodes, (const int32_t*)parent, depth);
  return hipGetLastError();
}

hipError_t refit_moved(float4* nodes, int n_nodes, float4* tri_verts, int n_tris, const float* vertices, const int32_t* indices,
                       const int32_t* depth, int max_level, const int32_t* node_mesh, const float* matrices, const int32_t* moved,
                       unsigned int* ext, int n_meshes, float4* cbox, hipStream_t st) {
  if (n_tris <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(ext, 0, sizeof(unsigned int) * (size_t)n_meshes, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_refit_tris, dim3((n_tris + 255) / 256), dim3(256), 0, st, tri_verts, n_tris, vertices, indices, matrices, moved, ext);
  for (int level = max_level; level >= 0 && n_nodes > 0; level--)               // bottom-up: a level reads what the level below wrote (stream order)
    hipLaunchKernelGGL(k_refit_level, dim3((n_nodes + 255) / 256), dim3(256), 0, st, nodes, n_nodes, (const float4*)tri_verts, depth, level, node_mesh,
                       moved, (const unsigned int*)ext, cbox);
  return hipGetLastError();
}

}  // namespace urtd
