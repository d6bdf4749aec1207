// cullflags.h — which object-level heap leaves may be culled (see cullflags.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace urtd {

// mesh_tlas: the device copy of the packed mesh heap (2 float4 per node: vmin.xyz, index | vmax.xyz, cull word).  The host packs the cull
// word of every ELIGIBLE leaf as a non-zero value (context.cpp pack_tlas: the leaf's position bit of the masked walk, or 1); this pass
// clears it for every leaf whose box does not contain all triangles of its MeshObject (records v0, v0 + e1, v0 + e2, with a slack of
// 2^-20 of the box's largest |coordinate|), and then ORs the surviving words into *walk_cull_mask (or null: heap without a walk table).
// mesh_leaf[m] = the heap node whose index is MeshObject m, or < 0 (none, or several: not eligible).
hipError_t update_cull_flags(float4* mesh_tlas, int n_nodes, const int32_t* mesh_leaf, int n_meshes, const float4* tri_verts, int n_tris,
                             int* walk_cull_mask, hipStream_t st);

}  // namespace urtd
