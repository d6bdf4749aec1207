// blas_builder.h — host-side triangle-BVH builder (see blas_builder.cpp)
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/urt_types.h"

namespace urtd {

static constexpr int kBlasNodeFloats = 16;
static constexpr int32_t kEmptyMeshRoot = 0x7fffffff;
static constexpr int32_t kTopOrderNodes = 256;   // node indices [0, this) are the top of the forest in breadth-first order

struct BlasResult {
  std::vector<float> nodes;            // 16 floats per node (urt_device.h)
  std::vector<int32_t> tri_slot;       // leaf order -> index slot i (RS:243)
  std::vector<int32_t> tri_mesh;       // leaf order -> MeshObject id
  std::vector<float> tri_verts;        // 12 floats per triangle: v0|slot, e1|mesh, e2|0
  std::vector<float> tri_norms;        // 12 floats per triangle: n0|0, n1|0, n2|0
  std::vector<int32_t> mesh_root;      // per MeshObject: node index, leaf code (<0) or kEmptyMeshRoot
  std::vector<int32_t> mesh_first_tri; // per MeshObject: first leaf-order slot
  int max_depth = 0;                   // deepest level (root = 1): bound on the traversal stack
};

// largest number of triangles the SAH builder puts in a leaf (1..8, default 4); process-wide
void set_blas_leaf_max(int n);
int get_blas_leaf_max();

// mesh_objects: n_meshes records of 112 bytes (urt_MeshObject).  Returns false and sets err when the
// buffers are inconsistent (an index outside _Vertices, a MeshObject range outside _Indices).
bool build_blas(const uint8_t* mesh_objects, int n_meshes, const float* vertices, int n_vertices, const int32_t* indices,
                int n_indices, const float* normals, int n_normals, BlasResult& out, std::string& err);

}  // namespace urtd
