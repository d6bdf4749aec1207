// qnodes.h — derived copies of the triangle-BVH nodes: centre / half-extent form and 32-byte quantized form (see qnodes.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace urtd {

// qbuf: 2 + 2 * n_nodes float4 (frame, then one 32-byte node per float node, same numbering).  Derived on the GPU from the float
// nodes — after a build and after every refit.
hipError_t quantize_nodes(const float4* nodes, int n_nodes, const int32_t* mesh_root, int n_meshes, float4* qbuf, hipStream_t st);

// cnodes: n_nodes x 4 float4, the (centre, half extent) form the trace kernels traverse (DevScene::blas_cnodes) — derived on the GPU from
// the [lo, hi] nodes after a build and after every refit.
hipError_t center_nodes(const float4* nodes, int n_nodes, float4* cnodes, hipStream_t st);

}  // namespace urtd
