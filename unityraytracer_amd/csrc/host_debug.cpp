// host_debug.cpp — the reference's debug log and BVH inspection, outside the Unity editor.
// Pure C++, no GPU.
//
//   urt_host_log                RayTraceDebug.Log (RD:25-36): append one line to a text file when level <= debugLevel
//   urt_host_log_scene_counts   the five count lines RebuildObjectLists writes (RM:331-335)
//   urt_host_log_tree_report    the two tree reports RebuildTrees writes (RM:731-735): amount, depth, complete length
//                               2^depth - 1, real length
//   urt_host_dump_bvh           text stand-in for the gizmo walk RayTraceDebug.DrawBVH (RD:92-117) over the implicit heap
//                               (children 2i+1 / 2i+2, `depth` levels from the root): one line per node with the label the
//                               gizmo prints — "(position in tree list, index of object)" (RD:108) —, its box and, when a test
//                               segment is given, whether RD's CPU slab test (RD:70-89, EPSILON = float.Epsilon) hits it
//                               (the gizmo paints those boxes black, RD:99-103)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/urt.h"

namespace {

std::string g_dbg_error;
int dbg_fail(int code, const std::string& m) { g_dbg_error = m; return code; }

// RD:70-89 literally: `end` is turned into a direction first; EPSILON is float.Epsilon (1.4e-45), not the shader's 1e-8
bool rd_intersect_bvh_node(const float* start, const float* end, const urt_BVHNode& node) {
  const float kRdEpsilon = 1.401298464324817e-45f;
  const float kFloatMax = 3.40282347e+38f;
  float t_min = -kFloatMax, t_max = kFloatMax;
  for (int i = 0; i < 3; i++) {
    float d = end[i] - start[i];
    float t1 = (node.vmin[i] - start[i]) / (d + kRdEpsilon);
    float t2 = (node.vmax[i] - start[i]) / (d + kRdEpsilon);
    t_min = std::fmax(t_min, std::fmin(t1, t2));
    t_max = std::fmin(t_max, std::fmax(t1, t2));
  }
  return t_max >= t_min;
}

void dump_rec(FILE* f, const urt_BVHNode* nodes, int n_nodes, int depth, int level, int index, const float* start, const float* end,
              int* lines) {
  if (depth <= 0) return;                                  // RD:93
  if (index >= n_nodes) return;                            // (List<> indexing would throw; a truncated list simply ends here)
  const urt_BVHNode& nd = nodes[index];
  bool hit = start && end && rd_intersect_bvh_node(start, end, nd);
  std::fprintf(f, "%*s(%d, %d) min (%.9g, %.9g, %.9g) max (%.9g, %.9g, %.9g) centre (%.9g, %.9g, %.9g)%s\n", 2 * level, "", index,
               nd.index, nd.vmin[0], nd.vmin[1], nd.vmin[2], nd.vmax[0], nd.vmax[1], nd.vmax[2], (nd.vmin[0] + nd.vmax[0]) / 2.0f,
               (nd.vmin[1] + nd.vmax[1]) / 2.0f, (nd.vmin[2] + nd.vmax[2]) / 2.0f, hit ? " [ray]" : "");
  (*lines)++;
  dump_rec(f, nodes, n_nodes, depth - 1, level + 1, index * 2 + 1, start, end, lines);   // RD:112-113
  dump_rec(f, nodes, n_nodes, depth - 1, level + 1, index * 2 + 2, start, end, lines);
}

}  // namespace

extern "C" {

const char* urt_host_debug_last_error(void) { return g_dbg_error.c_str(); }

int urt_host_log(const char* path, int debug_level, int level, const char* text) {
  if (!path || !text) return dbg_fail(URT_ERR_INVALID_ARGUMENT, "urt_host_log: NULL argument");
  if (level > debug_level) return 1;                       // RD:27: filtered, nothing written (the reference returns 1 too)
  FILE* f = std::fopen(path, "a");                          // RD:30: StreamWriter(path, append: true)
  if (!f) return dbg_fail(URT_ERR_INVALID_ARGUMENT, std::string("urt_host_log: cannot open ") + path);
  std::fputs(text, f);
  std::fputc('\n', f);                                      // WriteLine
  std::fclose(f);
  return URT_OK;
}

int urt_host_log_scene_counts(const char* path, int debug_level, int n_spheres, int n_mesh_objects, int n_vertices, int n_indices,
                              int n_normals) {
  const char* names[5] = {"# of Spheres: ", "# of Mesh Objects: ", "# of Vertices: ", "# of Indices: ", "# of Normals: "};   // RM:331-335
  const int vals[5] = {n_spheres, n_mesh_objects, n_vertices, n_indices, n_normals};
  for (int k = 0; k < 5; k++) {
    int rc = urt_host_log(path, debug_level, 2, (std::string(names[k]) + std::to_string(vals[k])).c_str());
    if (rc != URT_OK) return rc;
  }
  return URT_OK;
}

int urt_host_log_tree_report(const char* path, int debug_level, int n_mesh_objects, int mesh_depth, int mesh_real_length,
                             int n_spheres, int sphere_depth, int sphere_real_length) {
  // RM:731-732: (int) Mathf.Round(Mathf.Pow(2.0f, depth)) - 1
  auto complete = [](int depth) { return (int)std::lround(std::pow(2.0f, (float)depth)) - 1; };
  std::string a = "[MESH OBJECTS] \n > Amount: " + std::to_string(n_mesh_objects) + "\n > Depth: " + std::to_string(mesh_depth) +
                  "\n > Complete Length: " + std::to_string(complete(mesh_depth)) + "\n > Real Length: " + std::to_string(mesh_real_length);
  std::string b = "[SPHERES] \n > Amount: " + std::to_string(n_spheres) + "\n > Depth: " + std::to_string(sphere_depth) +
                  "\n > Complete Length: " + std::to_string(complete(sphere_depth)) + "\n > Real Length: " + std::to_string(sphere_real_length);
  int rc = urt_host_log(path, debug_level, 2, a.c_str());
  if (rc != URT_OK) return rc;
  return urt_host_log(path, debug_level, 2, b.c_str());
}

int urt_host_dump_bvh(const char* path, const urt_BVHNode* nodes, int n_nodes, int depth, const float* ray_start3, const float* ray_end3,
                      int* out_lines) {
  if (out_lines) *out_lines = 0;
  if (!path || (n_nodes > 0 && !nodes) || n_nodes < 0 || depth < 0) return dbg_fail(URT_ERR_INVALID_ARGUMENT, "urt_host_dump_bvh: bad arguments");
  if ((ray_start3 == nullptr) != (ray_end3 == nullptr)) return dbg_fail(URT_ERR_INVALID_ARGUMENT, "urt_host_dump_bvh: give both ends of the test segment or neither");
  FILE* f = std::fopen(path, "w");
  if (!f) return dbg_fail(URT_ERR_INVALID_ARGUMENT, std::string("urt_host_dump_bvh: cannot open ") + path);
  int lines = 0;
  if (n_nodes > 0) dump_rec(f, nodes, n_nodes, depth, 0, 0, ray_start3, ray_end3, &lines);
  std::fclose(f);
  if (out_lines) *out_lines = lines;
  return URT_OK;
}

/* Text stand-in for RayTraceDebug.DrawNormals (RD:165-183): for every index slot of every MeshObject, the base point the gizmo draws its
 * white sphere at — localToWorldMatrix.MultiplyPoint3x4(_vertices[_indices[i]]) — and the end of its blue line —
 * MultiplyPoint3x4(_vertices[_indices[i]] + _normals[_indices[i]] * 0.1f).  One line per slot: "mesh slot base -> tip". */
int urt_host_dump_normals(const char* path, const void* mesh_objects, int n_meshes, const float* vertices, int n_vertices, const int32_t* indices,
                          int n_indices, const float* normals, int n_normals, int* out_lines) {
  if (out_lines) *out_lines = 0;
  if (!path || n_meshes < 0 || (n_meshes > 0 && (!mesh_objects || !vertices || !indices || !normals)))
    return dbg_fail(URT_ERR_INVALID_ARGUMENT, "urt_host_dump_normals: bad arguments");
  FILE* f = std::fopen(path, "w");
  if (!f) return dbg_fail(URT_ERR_INVALID_ARGUMENT, std::string("urt_host_dump_normals: cannot open ") + path);
  int lines = 0;
  auto mul_point = [](const float* m, float x, float y, float z, float* o) {      // Matrix4x4.MultiplyPoint3x4: m (column-major) * (x, y, z, 1)
    for (int r = 0; r < 3; r++) o[r] = m[r] * x + m[4 + r] * y + m[8 + r] * z + m[12 + r];
  };
  for (int k = 0; k < n_meshes; k++) {
    urt_MeshObject mo;
    std::memcpy(&mo, (const char*)mesh_objects + (size_t)k * sizeof(urt_MeshObject), sizeof mo);
    for (long i = mo.indices_offset; i < (long)mo.indices_offset + mo.indices_count; i++) {
      if (i < 0 || i >= n_indices) { std::fclose(f); return dbg_fail(URT_ERR_SCENE, "urt_host_dump_normals: MeshObject index range outside _Indices"); }
      int vi = indices[i];
      if (vi < 0 || vi >= n_vertices || vi >= n_normals) { std::fclose(f); return dbg_fail(URT_ERR_SCENE, "urt_host_dump_normals: index outside _Vertices / _Normals"); }
      const float* v = vertices + 3 * (size_t)vi;
      const float* n = normals + 3 * (size_t)vi;
      float base[3], tip[3];
      mul_point(mo.localToWorldMatrix, v[0], v[1], v[2], base);
      mul_point(mo.localToWorldMatrix, v[0] + n[0] * 0.1f, v[1] + n[1] * 0.1f, v[2] + n[2] * 0.1f, tip);
      std::fprintf(f, "%d %ld (%.9g, %.9g, %.9g) -> (%.9g, %.9g, %.9g)\n", k, i, base[0], base[1], base[2], tip[0], tip[1], tip[2]);
      lines++;
    }
  }
  std::fclose(f);
  if (out_lines) *out_lines = lines;
  return URT_OK;
}

}  // extern "C"
