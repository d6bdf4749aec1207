// qnodes.h — 32-byte quantized copies of the triangle-BVH nodes (see qnodes.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace urtd {

// qbuf: 2 + 2 * n_nodes float4 (frame, then one 32-byte node per float node, same numbering).  Derived on the GPU from the float
// nodes — after a build and after every refit.
hipError_t quantize_nodes(const float4* nodes, int n_nodes, const int32_t* mesh_root, int n_meshes, float4* qbuf, hipStream_t st);

}  // namespace urtd
