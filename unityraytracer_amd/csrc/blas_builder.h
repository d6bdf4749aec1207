// blas_builder.h — host-side triangle-BVH builder (see blas_builder.cpp)
#pragma once
#include <stdint.h>
#include <mutex>
#include <string>
#include <unordered_map>
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

// Per-MeshObject results kept between builds of one context.  The reference re-uploads EVERY buffer whenever any object
// moves (RM:262-336 -> SetData); most MeshObjects are unchanged then, and their BVH — a function of the leaf size, the
// localToWorld matrix and the object-space positions of the mesh's index slots only — is reused instead of rebuilt.
struct BlasCacheEntry {
  uint64_t check = 0;                  // second, independent hash of the same inputs
  std::vector<float> nodes;            // mesh-local node numbering
  std::vector<int32_t> tri_rel;        // leaf order -> index slot relative to the mesh's indices_offset
  int32_t root = kEmptyMeshRoot;
  int max_depth = 0;
  uint64_t last_use = 0;
};
struct BlasCache {
  std::unordered_map<uint64_t, BlasCacheEntry> map;
  std::mutex lock;
  uint64_t generation = 0;
  uint64_t hits = 0, builds = 0;       // MeshObjects reused / built since the context was created
};

// largest number of triangles the SAH builder puts in a leaf (1..8, default 4); process-wide
void set_blas_leaf_max(int n);
int get_blas_leaf_max();

// mesh_objects: n_meshes records of 112 bytes (urt_MeshObject).  Returns false and sets err when the
// buffers are inconsistent (an index outside _Vertices, a MeshObject range outside _Indices).
bool build_blas(const uint8_t* mesh_objects, int n_meshes, const float* vertices, int n_vertices, const int32_t* indices,
                int n_indices, const float* normals, int n_normals, BlasResult& out, std::string& err, BlasCache* cache = nullptr);

}  // namespace urtd
