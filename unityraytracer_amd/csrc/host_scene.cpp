// host_scene.cpp — host-side scene preparation that RayTraceMaster.cs does in C# before it uploads buffers
// (SURVEY.md §8f rows f1 and f2).  Pure C++ behind the C ABI, no GPU needed.
//
//   urt_host_compute_normals     RayTraceMaster.ComputeNormals (RM:340-368) in O(V + I) instead of O(V * I)
//   urt_host_mesh_leaf_bounds    SetupBVHLeaves(List<MeshObject>) (RM:405-433), literal incl. its quirks (A.7), or tight
//   urt_host_sphere_leaf_bounds  SetupBVHLeaves(List<Sphere>) (RM:436-455), literal (inverted boxes) or normalised
//   urt_host_build_object_bvh    the OUTPUT CONTRACT of CreateBVH (RM:681-722): implicit heap, 2^D - 1 nodes; the pairing
//                                heuristic of RM:510-678 depends on .NET's unstable List.Sort and is not reproducible
//                                (SURVEY.md A.7), so the tree itself is built by a deterministic median split
//
// Arithmetic follows UnityEngine's float32 Vector3 helpers: separate multiplies and adds (no fma; this file is
// compiled with -ffp-contract=off), left-to-right sums, Vector3.Normalize = v / sqrt(x*x + y*y + z*z) or zero when the
// magnitude is <= 1e-5.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/urt.h"

namespace {

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }   // Vector3.Cross
inline V3 normalize(V3 v) {                                                                                        // Vector3.Normalize
  float mag = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  if (mag > 1e-5f) return {v.x / mag, v.y / mag, v.z / mag};
  return {0.0f, 0.0f, 0.0f};
}
inline V3 ld(const float* p) { return {p[0], p[1], p[2]}; }

struct Key {
  uint32_t a, b, c;
  bool operator==(const Key& o) const { return a == o.a && b == o.b && c == o.c; }
};
struct KeyHash {
  size_t operator()(const Key& k) const {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ k.a;
    h = (h ^ (h >> 32)) * 0xBF58476D1CE4E5B9ull + k.b;
    h = (h ^ (h >> 29)) * 0x94D049BB133111EBull + k.c;
    return (size_t)(h ^ (h >> 31));
  }
};
inline uint32_t bits_no_negzero(float f) {           // -0 == +0 for the reference's (a - b).sqrMagnitude test
  uint32_t u; std::memcpy(&u, &f, 4);
  return u == 0x80000000u ? 0u : u;
}

// MultiplyPoint3x4 on a Unity Matrix4x4 stored column-major (m[col*4+row]): rows evaluated left to right, mul then add
inline V3 multiply_point_3x4(const float* m, V3 p) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
          m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}

std::string g_host_error;
int host_fail(int code, const std::string& msg) { g_host_error = msg; return code; }

}  // namespace

extern "C" {

const char* urt_host_last_error(void) { return g_host_error.c_str(); }

}  // extern "C"

// ---- the reference's own object-level BVH builder, restated: SetupBVHRankList / PairBVHBounds / JoinBVH / CreateBVH (RM:459-722) ----
// Literal, quirks included:
//  * BVHNode equality is UnityEngine's APPROXIMATE Vector3 == (squared distance < 9.99999944e-11) on both corners plus the
//    index (RM:153-155); `nodes.FindIndex(x => x == other)` is the first such node;
//  * the "forbidden" test measures the distance of a bystander's centre from the line through the ORIGIN along the vector
//    between the two centres (RM:546), and every bystander it catches FLIPS the sign of the distance (RM:550);
//  * `bestPairing = pairing` (RM:670) aliases the list that every outer iteration clears, so the volume comparison is dead and
//    the candidate built from start index n-1 always wins — all n candidates are still built here, as written;
//  * a lone tree is joined with an empty list under a copy of its own root (RM:661-665): that object's id then also sits on an
//    interior position, where the traversal treats it as a leaf (RS:311-318) and never descends further.
// The one thing that cannot be restated is the ORDER OF TIES in ranking.Sort (RM:561): List<T>.Sort is an unstable introsort
// whose tie order is the runtime's business.  Ties keep insertion order here (std::stable_sort) — "parity unpinned" for
// scenes with exactly equal pair distances; pixels do not depend on the heap's shape (tests/test_host_scene.py).
namespace {

typedef std::vector<urt_BVHNode> Tree;

inline bool unity_v3_equal(const float* a, const float* b) {          // UnityEngine.Vector3 operator ==
  float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
  return x * x + y * y + z * z < 9.99999944e-11f;
}
inline bool node_equal(const urt_BVHNode& a, const urt_BVHNode& b) {  // RM:153-155
  return unity_v3_equal(b.vmax, a.vmax) && unity_v3_equal(b.vmin, a.vmin) && b.index == a.index;
}
inline int find_index(const std::vector<urt_BVHNode>& nodes, const urt_BVHNode& x) {
  for (size_t i = 0; i < nodes.size(); i++) if (node_equal(nodes[i], x)) return (int)i;
  return -1;
}
inline float magnitude(V3 v) { return (float)std::sqrt((double)(v.x * v.x + v.y * v.y + v.z * v.z)); }   // Vector3.Magnitude
inline float sqr_magnitude(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
inline V3 centre(const urt_BVHNode& n) { return {(n.vmax[0] + n.vmin[0]) / 2.0f, (n.vmax[1] + n.vmin[1]) / 2.0f, (n.vmax[2] + n.vmin[2]) / 2.0f}; }
inline float unity_sign(float f) { return f >= 0.0f ? 1.0f : -1.0f; }  // Mathf.Sign

// RM:459-505
Tree join_bvh(const urt_BVHNode& parent, const Tree* left, const Tree* right) {
  double ll = left->empty() ? -INFINITY : std::log((double)left->size()) / std::log(2.0);
  double lr = right->empty() ? -INFINITY : std::log((double)right->size()) / std::log(2.0);
  int depth = (int)std::ceil((float)std::max((float)ll, (float)lr)) + 1;      // Mathf.CeilToInt(Mathf.Max(Mathf.Log(..,2), ..)) + 1
  if (depth <= 1) depth = 2;
  if (right->size() > left->size()) std::swap(left, right);
  Tree list;
  list.push_back(parent);
  size_t start = 0, sub = 1;
  urt_BVHNode filler;
  std::memset(&filler, 0, sizeof filler); filler.index = -1;
  for (int i = 1; i < depth; i++) {
    for (size_t k = 0; k < sub; k++) list.push_back(start + k < left->size() ? (*left)[start + k] : filler);   // (left is complete in every call CreateBVH makes)
    for (size_t k = 0; k < sub; k++) list.push_back(start + k < right->size() ? (*right)[start + k] : filler);
    start += sub; sub *= 2;
  }
  return list;
}

// RM:510-595
std::vector<std::vector<urt_BVHNode>> setup_rank_list(const std::vector<Tree>& trees) {
  std::vector<urt_BVHNode> nodes;
  for (const Tree& t : trees) nodes.push_back(t[0]);
  std::vector<std::vector<urt_BVHNode>> rank_list;
  struct Rank { int index; double dis; };
  for (const urt_BVHNode& chosen : nodes) {
    std::vector<Rank> ranking;
    for (const urt_BVHNode& other : nodes) {
      if (node_equal(chosen, other)) continue;                            // chosen != other
      V3 a = {chosen.vmax[0] - other.vmin[0], chosen.vmax[1] - other.vmin[1], chosen.vmax[2] - other.vmin[2]};
      V3 b = {chosen.vmin[0] - other.vmax[0], chosen.vmin[1] - other.vmax[1], chosen.vmin[2] - other.vmax[2]};
      double distance = (double)std::min(sqr_magnitude(a), sqr_magnitude(b));
      V3 test = sub(centre(chosen), centre(other));
      if (magnitude(test) != 0) {
        for (const urt_BVHNode& by : nodes) {
          if (node_equal(by, chosen) || node_equal(by, other)) continue;
          float line_dis = magnitude(cross(centre(by), test)) / magnitude(test);
          V3 diag = {by.vmax[0] - by.vmin[0], by.vmax[1] - by.vmin[1], by.vmax[2] - by.vmin[2]};
          if (line_dis <= magnitude(diag) / 2.0f) distance *= -1;
        }
      }
      ranking.push_back({find_index(nodes, other), distance});
    }
    std::stable_sort(ranking.begin(), ranking.end(), [](const Rank& x, const Rank& y) {   // RM:561-580 as a strict "comes before"
      if (x.dis == y.dis) return false;
      if (unity_sign((float)x.dis) != unity_sign((float)y.dis)) return !(x.dis < y.dis);      // the negative ("forbidden") one goes last
      if (x.dis < 0 || y.dis < 0) return -x.dis < -y.dis;
      return x.dis < y.dis;
    });
    std::vector<urt_BVHNode> cur;
    for (const Rank& r : ranking) cur.push_back(nodes[(size_t)r.index]);
    rank_list.push_back(cur);
  }
  return rank_list;
}

// RM:598-678
void pair_bvh_bounds(std::vector<Tree>& trees) {
  std::vector<std::vector<urt_BVHNode>> rank_list = setup_rank_list(trees);
  std::vector<Tree> pairing;                        // `bestPairing` is only ever an alias of this list
  std::vector<urt_BVHNode> nodes;
  for (const Tree& t : trees) nodes.push_back(t[0]);
  const int num_tests = (int)nodes.size();
  const Tree empty;
  for (int i = 0; i < num_tests; i++) {
    pairing.clear();
    std::vector<char> paired(nodes.size(), 0);
    for (int j = i; j < i + num_tests; j++) {
      int index = j % (int)trees.size();
      if (paired[(size_t)index]) continue;
      for (size_t k = 0; k < rank_list[(size_t)index].size(); k++) {
        int other = find_index(nodes, rank_list[(size_t)index][k]);
        if (other >= 0 && !paired[(size_t)other]) {
          paired[(size_t)index] = 1; paired[(size_t)other] = 1;
          const urt_BVHNode& p = nodes[(size_t)index]; const urt_BVHNode& q = nodes[(size_t)other];
          urt_BVHNode parent;
          for (int c = 0; c < 3; c++) {             // RM:642-647: min / max over BOTH corners of both boxes
            parent.vmin[c] = std::min(std::min(p.vmin[c], q.vmin[c]), std::min(p.vmax[c], q.vmax[c]));
            parent.vmax[c] = std::max(std::max(p.vmin[c], q.vmin[c]), std::max(p.vmax[c], q.vmax[c]));
          }
          parent.index = -1;
          pairing.push_back(join_bvh(parent, &trees[(size_t)index], &trees[(size_t)other]));
          break;
        }
      }
      if (!paired[(size_t)index]) pairing.push_back(join_bvh(nodes[(size_t)index], &trees[(size_t)index], &empty));   // RM:661-665
    }
  }
  trees = pairing;                                  // RM:677 (the candidate of the last start index)
}

}  // namespace

extern "C" {

int urt_host_build_object_bvh_pairing(const urt_BVHNode* leaves, int n_objects, urt_BVHNode* out_nodes, int capacity) {
  int len = urt_host_object_bvh_length(n_objects);
  if (n_objects < 0 || (n_objects && (!leaves || !out_nodes)) || capacity < len) return host_fail(URT_ERR_INVALID_ARGUMENT, "CreateBVH: bad arguments");
  if (n_objects == 0) return URT_OK;                 // (the reference throws on an empty list, A.7; an empty list is "no buffer" here)
  try {
    int depth = 1;
    while ((1 << (depth - 1)) < n_objects) depth++;  // RM:683,705: Mathf.CeilToInt(Mathf.Log(n, 2)) + 1
    std::vector<Tree> trees((size_t)n_objects);
    for (int i = 0; i < n_objects; i++) trees[(size_t)i].push_back(leaves[i]);
    for (int i = 0; i < depth - 1; i++) pair_bvh_bounds(trees);
    if (trees.empty() || (int)trees[0].size() != len) return host_fail(URT_ERR_SCENE, "CreateBVH: the pairing did not end in one complete tree");
    std::memcpy(out_nodes, trees[0].data(), sizeof(urt_BVHNode) * (size_t)len);
    return URT_OK;
  } catch (...) { return host_fail(URT_ERR_OUT_OF_MEMORY, "CreateBVH: allocation failed"); }
}

int urt_host_compute_normals(const float* vertices, int n_vertices, const int32_t* indices, int n_indices, float* out_normals) {
  if (n_vertices < 0 || n_indices < 0 || n_indices % 3 != 0) return host_fail(URT_ERR_INVALID_ARGUMENT, "ComputeNormals: bad counts");
  if ((n_vertices && (!vertices || !out_normals)) || (n_indices && !indices)) return host_fail(URT_ERR_INVALID_ARGUMENT, "ComputeNormals: NULL array");
  try {
    // RM:351 welds index slots whose vertex satisfies (v - v_i).sqrMagnitude <= EPSILON = 3 * float.Epsilon (4.2e-45): that is
    // plain equality (-0 == +0) EXCEPT for differences so small that their squares underflow.  Two DIFFERENT positions can only
    // pass when, on every axis, their coordinates are equal or both of magnitude below 2^-50 (distinct floats of magnitude
    // >= 2^-50 differ by >= 2^-73, whose square 1.1e-44 already exceeds EPSILON; numerical zeros like sin(pi) * r = 1e-17
    // are common in real meshes, so this is not a corner to ignore).
    // Step 1: exact groups (O(V + I) position hash).
    std::unordered_map<Key, int, KeyHash> groups;
    groups.reserve((size_t)n_vertices * 2);
    std::vector<int> group_of((size_t)n_vertices);
    std::vector<int> rep;                                   // one vertex per group
    for (int i = 0; i < n_vertices; i++) {
      Key k{bits_no_negzero(vertices[3 * i]), bits_no_negzero(vertices[3 * i + 1]), bits_no_negzero(vertices[3 * i + 2])};
      auto it = groups.find(k);
      if (it == groups.end()) { it = groups.emplace(k, (int)groups.size()).first; rep.push_back(i); }
      group_of[(size_t)i] = it->second;
    }
    const size_t ng = groups.size();
    // Step 2: groups that weld with OTHER groups.  Candidates share a coarse key (coordinates below 2^-50 mapped to 0);
    // inside a coarse bucket the reference's float test decides, pair by pair (the relation is not transitive).
    const float kTiny = 8.8817842e-16f;                     // 2^-50
    const float kEps = 3.0f * 1.401298464e-45f;              // RM:14
    std::vector<std::vector<int>> partners(0);
    std::vector<int> partner_slot(ng, -1);                   // group -> index into `partners`, or -1 (the common case)
    {
      std::unordered_map<Key, std::vector<int>, KeyHash> coarse;
      for (size_t g = 0; g < ng; g++) {
        const float* p = vertices + 3 * (size_t)rep[g];
        bool has_small = false;
        uint32_t kk[3];
        for (int c = 0; c < 3; c++) { bool small = std::fabs(p[c]) < kTiny; has_small = has_small || small; kk[c] = small ? 0u : bits_no_negzero(p[c]); }
        if (has_small) coarse[Key{kk[0], kk[1], kk[2]}].push_back((int)g);
      }
      for (auto& kv : coarse) {
        const std::vector<int>& b = kv.second;
        if (b.size() < 2) continue;
        for (size_t x = 0; x < b.size(); x++)
          for (size_t y = x + 1; y < b.size(); y++) {
            V3 d = sub(ld(vertices + 3 * (size_t)rep[(size_t)b[x]]), ld(vertices + 3 * (size_t)rep[(size_t)b[y]]));
            if (d.x * d.x + d.y * d.y + d.z * d.z <= kEps) {
              for (int side = 0; side < 2; side++) {
                int g = side ? b[y] : b[x], o = side ? b[x] : b[y];
                if (partner_slot[(size_t)g] < 0) { partner_slot[(size_t)g] = (int)partners.size(); partners.emplace_back(); }
                partners[(size_t)partner_slot[(size_t)g]].push_back(o);
              }
            }
          }
      }
    }
    std::vector<V3> acc(ng, V3{0, 0, 0});
    std::vector<std::vector<int>> slots(partners.size());    // ascending index slots of the groups that have partners
    // every index slot j (ascending, the order of the reference's LINQ query) adds the un-normalised normal of ITS
    // triangle to the group of the vertex it refers to (RM:355-362)
    for (int start = 0; start + 2 < n_indices; start += 3) {
      int i0 = indices[start], i1 = indices[start + 1], i2 = indices[start + 2];
      if (i0 < 0 || i0 >= n_vertices || i1 < 0 || i1 >= n_vertices || i2 < 0 || i2 >= n_vertices)
        return host_fail(URT_ERR_SCENE, "ComputeNormals: index outside _Vertices");
      V3 a = ld(vertices + 3 * i0), b = ld(vertices + 3 * i1), c = ld(vertices + 3 * i2);
      V3 face = cross(sub(b, a), sub(c, a));
      const int vs[3] = {i0, i1, i2};
      for (int j = 0; j < 3; j++) {
        int g = group_of[(size_t)vs[j]];
        V3& s = acc[(size_t)g]; s = add(s, face);
        if (partner_slot[(size_t)g] >= 0) slots[(size_t)partner_slot[(size_t)g]].push_back(start + j);
      }
    }
    // groups with partners: re-sum over the union of the slot lists in ascending slot order (float adds do not commute)
    for (size_t g = 0; g < ng; g++) {
      int ps = partner_slot[g];
      if (ps < 0) continue;
      std::vector<int> all = slots[(size_t)ps];
      for (int o : partners[(size_t)ps]) { const std::vector<int>& so = slots[(size_t)partner_slot[(size_t)o]]; all.insert(all.end(), so.begin(), so.end()); }
      std::sort(all.begin(), all.end());
      V3 s{0, 0, 0};
      for (int j : all) {
        int start = j - j % 3;
        V3 a = ld(vertices + 3 * (size_t)indices[start]);
        s = add(s, cross(sub(ld(vertices + 3 * (size_t)indices[start + 1]), a), sub(ld(vertices + 3 * (size_t)indices[start + 2]), a)));
      }
      acc[g] = s;
    }
    for (int i = 0; i < n_vertices; i++) {
      V3 n = normalize(acc[(size_t)group_of[(size_t)i]]);
      out_normals[3 * i] = n.x; out_normals[3 * i + 1] = n.y; out_normals[3 * i + 2] = n.z;
    }
    return URT_OK;
  } catch (...) { return host_fail(URT_ERR_OUT_OF_MEMORY, "ComputeNormals: allocation failed"); }
}

int urt_host_mesh_leaf_bounds(const void* mesh_objects, int n_meshes, const float* vertices, int n_vertices, const int32_t* indices,
                              int n_indices, int literal, urt_BVHNode* out_leaves) {
  if (n_meshes < 0 || (n_meshes && (!mesh_objects || !out_leaves))) return host_fail(URT_ERR_INVALID_ARGUMENT, "SetupBVHLeaves: bad arguments");
  for (int m = 0; m < n_meshes; m++) {
    urt_MeshObject mo;
    std::memcpy(&mo, (const uint8_t*)mesh_objects + (size_t)m * sizeof mo, sizeof mo);
    long off = mo.indices_offset, cnt = mo.indices_count;
    if (off < 0 || cnt < 0 || off + cnt > n_indices) return host_fail(URT_ERR_SCENE, "SetupBVHLeaves: MeshObject range outside _Indices");
    auto point = [&](long slot, V3& out) -> bool {
      int vi = indices[slot];
      if (vi < 0 || vi >= n_vertices) return false;
      out = multiply_point_3x4(mo.localToWorldMatrix, ld(vertices + 3 * (size_t)vi));
      return true;
    };
    V3 lo, hi;
    long first;
    if (literal) {
      // RM:415-416: starts from _vertices[_indices[0]] — the GLOBAL first index, through THIS mesh's matrix — and
      // RM:421: refines from offset + 1, so the mesh's own first index slot is skipped (A.7)
      if (n_indices <= 0 || !point(0, lo)) return host_fail(URT_ERR_SCENE, "SetupBVHLeaves: _Indices[0] invalid");
      hi = lo;
      first = off + 1;
    } else {
      if (cnt == 0) { std::memset(&out_leaves[m], 0, sizeof(urt_BVHNode)); out_leaves[m].index = m; continue; }
      if (!point(off, lo)) return host_fail(URT_ERR_SCENE, "SetupBVHLeaves: index outside _Vertices");
      hi = lo;
      first = off + 1;
    }
    for (long i = first; i < off + cnt; i++) {
      V3 t;
      if (!point(i, t)) return host_fail(URT_ERR_SCENE, "SetupBVHLeaves: index outside _Vertices");
      lo = {std::min(lo.x, t.x), std::min(lo.y, t.y), std::min(lo.z, t.z)};      // Mathf.Min / Mathf.Max
      hi = {std::max(hi.x, t.x), std::max(hi.y, t.y), std::max(hi.z, t.z)};
    }
    out_leaves[m].vmin[0] = lo.x; out_leaves[m].vmin[1] = lo.y; out_leaves[m].vmin[2] = lo.z;
    out_leaves[m].vmax[0] = hi.x; out_leaves[m].vmax[1] = hi.y; out_leaves[m].vmax[2] = hi.z;
    out_leaves[m].index = m;                                                     // RM:417
  }
  return URT_OK;
}

int urt_host_sphere_leaf_bounds(const void* spheres, int n_spheres, int literal, urt_BVHNode* out_leaves) {
  if (n_spheres < 0 || (n_spheres && (!spheres || !out_leaves))) return host_fail(URT_ERR_INVALID_ARGUMENT, "SetupBVHLeaves: bad arguments");
  for (int i = 0; i < n_spheres; i++) {
    urt_Sphere s;
    std::memcpy(&s, (const uint8_t*)spheres + (size_t)i * sizeof s, sizeof s);
    // RM:445-446: vmin = position - (-r,-r,-r), vmax = position - (r,r,r): INVERTED; harmless for the slab test and the unions
    float a[3], b[3];
    for (int k = 0; k < 3; k++) { a[k] = s.position[k] - (-s.radius); b[k] = s.position[k] - s.radius; }
    for (int k = 0; k < 3; k++) {
      out_leaves[i].vmin[k] = literal ? a[k] : std::min(a[k], b[k]);
      out_leaves[i].vmax[k] = literal ? b[k] : std::max(a[k], b[k]);
    }
    out_leaves[i].index = i;
  }
  return URT_OK;
}

int urt_host_object_bvh_length(int n_objects) {       // RM:683,705: depth = ceil(log2 n) + 1, length 2^depth - 1
  if (n_objects <= 0) return 0;
  int depth = 1;
  while ((1 << (depth - 1)) < n_objects) depth++;
  return (1 << depth) - 1;
}

int urt_host_build_object_bvh(const urt_BVHNode* leaves, int n_objects, urt_BVHNode* out_nodes, int capacity) {
  int len = urt_host_object_bvh_length(n_objects);
  if (n_objects < 0 || (n_objects && (!leaves || !out_nodes)) || capacity < len) return host_fail(URT_ERR_INVALID_ARGUMENT, "CreateBVH: bad arguments");
  try {
    for (int i = 0; i < len; i++) { std::memset(&out_nodes[i], 0, sizeof(urt_BVHNode)); out_nodes[i].index = -1; }   // filler (RM:490-494)
    if (n_objects == 0) return URT_OK;
    struct Item { float c[3]; int leaf; };
    std::vector<Item> items((size_t)n_objects);
    for (int i = 0; i < n_objects; i++) {
      for (int k = 0; k < 3; k++) items[(size_t)i].c[k] = 0.5f * leaves[i].vmin[k] + 0.5f * leaves[i].vmax[k];
      items[(size_t)i].leaf = i;
    }
    struct Job { int slot, lo, hi; };
    std::vector<Job> jobs{{0, 0, n_objects}};
    while (!jobs.empty()) {
      Job j = jobs.back(); jobs.pop_back();
      urt_BVHNode& nd = out_nodes[j.slot];
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, clo[3], chi[3];
      std::memcpy(clo, lo, sizeof lo); std::memcpy(chi, hi, sizeof hi);
      for (int q = j.lo; q < j.hi; q++) {
        const urt_BVHNode& lf = leaves[items[(size_t)q].leaf];
        for (int k = 0; k < 3; k++) {
          // union over min/max of both corners, as the reference's parent boxes do (RM:642-647)
          lo[k] = std::min(lo[k], std::min(lf.vmin[k], lf.vmax[k]));
          hi[k] = std::max(hi[k], std::max(lf.vmin[k], lf.vmax[k]));
          clo[k] = std::min(clo[k], items[(size_t)q].c[k]); chi[k] = std::max(chi[k], items[(size_t)q].c[k]);
        }
      }
      if (j.hi - j.lo == 1) { nd = leaves[items[(size_t)j.lo].leaf]; continue; }      // a leaf keeps its own (possibly inverted) box and index
      for (int k = 0; k < 3; k++) { nd.vmin[k] = lo[k]; nd.vmax[k] = hi[k]; }
      nd.index = -1;
      int ax = 0;
      if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
      if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
      std::sort(items.begin() + j.lo, items.begin() + j.hi, [ax](const Item& a, const Item& b) {
        return a.c[ax] < b.c[ax] || (a.c[ax] == b.c[ax] && a.leaf < b.leaf);
      });
      int half = (j.hi - j.lo + 1) / 2;
      jobs.push_back({2 * j.slot + 1, j.lo, j.lo + half});
      jobs.push_back({2 * j.slot + 2, j.lo + half, j.hi});
    }
    return URT_OK;
  } catch (...) { return host_fail(URT_ERR_OUT_OF_MEMORY, "CreateBVH: allocation failed"); }
}

}  // extern "C"
