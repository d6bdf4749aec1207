// lbvh.h — GPU triangle-BVH build (see lbvh.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace urtd {

struct LbvhInput {
  // DEVICE copies of the reference-layout buffers exactly as SetData delivered them
  const uint8_t* mesh_objects = nullptr; int n_meshes = 0;      // 112-byte records (RM:82-86)
  const float* vertices = nullptr; int n_vertices = 0;          // _Vertices
  const int32_t* indices = nullptr; int n_indices = 0;          // _Indices
  const float* normals = nullptr; int n_normals = 0;            // _Normals (may be null)
  // HOST copies of each MeshObject's (indices_offset, indices_count): they lay out the per-mesh segments
  const int32_t* h_offsets = nullptr; const int32_t* h_counts = nullptr;
  int leaf_max = 2;                                             // triangles per leaf (1..8)
  bool sah = false;                                             // builder 3: binned SAH on the GPU, level by level (lbvh.hip k_sah_*): the host builder's algorithm
  int depth_slack = 6;                                          // builder 2: levels beyond a median tree of the biggest MeshObject that lopsided radix splits may use
  bool depth_budget = false;                                    // builder 2: the radix tree built top-down, lopsided splits halved once the depth budget would be exceeded (lbvh.hip k_td_level)
};

struct LbvhOutput {
  // device arrays in the layouts of urt_device.h; owned by the caller after a successful build (hipFree each of `allocs`)
  float4* nodes = nullptr; int n_nodes = 0;
  float4* tri_verts = nullptr; float4* tri_norms = nullptr; int n_tris = 0;
  int32_t* mesh_root = nullptr;                                 // device, one per MeshObject
  std::vector<int32_t> h_mesh_root;                             // the same on the host
  int max_depth = 0;                                            // deepest level (root = 1, leaves included)
  std::vector<void*> allocs;
};

// Builds one LBVH per MeshObject on the GPU (Morton sort + Karras hierarchy + bottom-up fit), in the node / triangle-record
// formats the trace kernels read.  Synchronises `st` before returning (the sizes come back to the host).  Returns a urt status
// code (URT_OK, URT_ERR_SCENE for inconsistent buffers, URT_ERR_HIP / URT_ERR_OUT_OF_MEMORY) and sets err.
int lbvh_build(const LbvhInput& in, hipStream_t st, LbvhOutput& out, std::string& err);

}  // namespace urtd
