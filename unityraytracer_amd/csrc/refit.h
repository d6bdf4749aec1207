// refit.h — GPU refit of the triangle BVHs of moved MeshObjects (see refit.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace urtd {

// Once per full scene preparation: parent[n] (or -1 for a root), node_mesh[n] (MeshObject id) and depth[n] of every node, derived on the GPU
// from the node array and the triangle records of either builder.
hipError_t refit_prepare(const float4* nodes, int n_nodes, const float4* tri_verts, int32_t* parent, int32_t* node_mesh, int32_t* depth, hipStream_t st);

// MeshObjects with moved[m] != 0 take matrices[16 m ..] (Unity memory order) as their new localToWorldMatrix: their leaf-order
// triangle records are re-derived from vertices / indices (device copies of _Vertices / _Indices) and their node boxes refitted
// bottom-up (one launch per level, max_level = the deepest interior level), in place.  ext: n_meshes words of scratch; cbox: 4 float4 per
// node of scratch.
hipError_t refit_moved(float4* nodes, int n_nodes, float4* tri_verts, int n_tris, const float* vertices, const int32_t* indices,
                       const int32_t* depth, int max_level, const int32_t* node_mesh, const float* matrices, const int32_t* moved,
                       unsigned int* ext, int n_meshes, float4* cbox, hipStream_t st);

}  // namespace urtd
