// blas_builder.cpp — host-side triangle-BVH ("BLAS") builder.
//
// The reference has no triangle-level acceleration structure: IntersectMeshObject tests every triangle
// of a mesh per ray (RayTraceShader.compute:243-266), which is why it is "very slow for complex
// scenes" (reference README.md:2).  This builder turns the reference's own buffers (_Vertices,
// _Indices, _MeshObjects as uploaded through urt_buffer_set_data) into one binned-SAH BVH per
// MeshObject over WORLD-SPACE triangles: the per-ray `mul(localToWorldMatrix, float4(v,1))` of
// RS:244-246 is applied once here with the same normative formula (urt::mul_m4), so the triangle
// test on the GPU sees bit-identical vertices.
//
// Output (layouts in urt_device.h): 64-byte nodes holding both children's boxes, leaf-ordered
// triangle records (v0, e1 = v1-v0, e2 = v2-v0, index slot, mesh id) and leaf-ordered vertex normals.
// Child boxes are padded by 2^-16 * (largest |coordinate| of the mesh) so that the slab test can never
// cull a triangle that the float32 Moller-Trumbore test would accept (DESIGN.md "BLAS traversal").
#include "experiments.h"
#include "blas_builder.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <limits>
#include <thread>

#include "../../include/urt_math.h"

namespace urtd {

namespace {

struct Prim {
  float lo[3], hi[3], c[3];
  int32_t slot;   // index slot i of RS:243 (global position in _Indices, multiple of 3 from the mesh offset)
};

struct Box {
  float lo[3], hi[3];
  void reset() { for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; } }
  void grow(const float* l, const float* h) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); } }
  float half_area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
  }
};

#ifndef URT_SAH_BINS
#define URT_SAH_BINS 32
#endif
constexpr int kBins = URT_SAH_BINS;
// Two triangles per leaf: a leaf trip of the traversal loop is then always ONE round of two Moller-Trumbore tests (no second round for
// the lanes with bigger leaves), and node steps are the cheap body since the centre-form slab test.  Measured with the final kernels
// (profiles/r03_logs/r3_sweep_leaf_max.log): 4 -> 2 triangles: C3 -3.0 %, C3D -3.0 %, C4 -2.2 %, C5 -4.2 % frame time; 1: +5 ... +10 %.
constexpr int kLeafMaxDefault = 2;
int leaf_max_from_env() {
  const char* e = std::getenv("URT_BLAS_LEAF_MAX");
  int v = e ? std::atoi(e) : kLeafMaxDefault;
  return v >= 1 && v <= 8 ? v : kLeafMaxDefault;
}
int g_leaf_max = leaf_max_from_env();   // SAH leaves (tunable: URT_BLAS_LEAF_MAX at load time / urt_set_option "blas_leaf_max")
constexpr int kLeafHardMax = 8;  // encoding limit (3 bits)
constexpr int kForkMinPrims = 8192;   // subtrees at least this big may get a thread of their own

struct Builder {
  std::vector<Prim> prims;           // owned by the MeshObject's root builder ...
  Prim* P = nullptr;                 // ... and shared (disjoint index ranges) with the builders of its big subtrees
  int fork_levels = 0;               // levels below which big subtrees may still be built on their own thread ...
  std::atomic<int>* spare = nullptr; // ... when one of the build's threads is idle (shared counter)
  // output of ONE mesh, in mesh-local numbering (node indices and leaf-order triangle slots start at 0): meshes are built
  // independently (in parallel) and concatenated afterwards
  std::vector<float> nodes;
  std::vector<int32_t> tri_slot;
  int32_t root = kEmptyMeshRoot;
  float pad = 0;
  int max_depth = 0;
  std::string err;

  int32_t make_leaf(int lo, int hi) {
    uint32_t first = (uint32_t)tri_slot.size();
    for (int q = lo; q < hi; q++) tri_slot.push_back(P[q].slot);
    uint32_t code = (first << 3) | (uint32_t)(hi - lo - 1);
    return (int32_t)~code;
  }

  // append a subtree built elsewhere (its own local numbering) and return its root's code in THIS builder's numbering
  int32_t append(const Builder& sub, int32_t code) {
    int32_t node_base = (int32_t)(nodes.size() / kBlasNodeFloats);
    uint32_t tri_base = (uint32_t)tri_slot.size();
    auto rebase = [&](int32_t c) -> int32_t {
      if (c >= 0) return c + node_base;
      uint32_t u = ~(uint32_t)c;
      return (int32_t)~((((u >> 3) + tri_base) << 3) | (u & 7u));
    };
    size_t at = nodes.size();
    nodes.insert(nodes.end(), sub.nodes.begin(), sub.nodes.end());
    for (size_t k = at; k < nodes.size(); k += kBlasNodeFloats)
      for (int c = 0; c < 2; c++) nodes[k + 12 + c] = urt::bits_f((uint32_t)rebase((int32_t)urt::f_bits(nodes[k + 12 + c])));
    tri_slot.insert(tri_slot.end(), sub.tri_slot.begin(), sub.tri_slot.end());
    max_depth = std::max(max_depth, sub.max_depth);
    return rebase(code);
  }

  // returns the child code for prims[lo,hi) and writes its (padded) box
  int32_t build(int lo, int hi, int depth, Box& box) {
    max_depth = std::max(max_depth, depth);
    int n = hi - lo;
    box.reset();
    Box cb; cb.reset();
    for (int q = lo; q < hi; q++) { box.grow(P[q].lo, P[q].hi); cb.grow(P[q].c, P[q].c); }
    if (n <= g_leaf_max) return make_leaf(lo, hi);

    // binned SAH over the three axes
    int best_axis = -1, best_bin = -1;
    float best_cost = std::numeric_limits<float>::infinity();
    for (int ax = 0; ax < 3; ax++) {
      float ext = cb.hi[ax] - cb.lo[ax];
      if (!(ext > 0)) continue;
      int cnt[kBins] = {0};
      Box bb[kBins];
      for (auto& b : bb) b.reset();
      float scale = (float)kBins / ext;
      for (int q = lo; q < hi; q++) {
        int b = (int)((P[q].c[ax] - cb.lo[ax]) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        cnt[b]++; bb[b].grow(P[q].lo, P[q].hi);
      }
      float right_area[kBins]; int right_cnt[kBins];
      Box acc; acc.reset(); int c = 0;
      for (int b = kBins - 1; b > 0; b--) { acc.grow(bb[b].lo, bb[b].hi); c += cnt[b]; right_area[b] = acc.half_area(); right_cnt[b] = c; }
      acc.reset(); c = 0;
      for (int b = 0; b < kBins - 1; b++) {
        acc.grow(bb[b].lo, bb[b].hi); c += cnt[b];
        if (c == 0 || right_cnt[b + 1] == 0) continue;
        float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
      }
    }
    int mid;
    if (best_axis >= 0) {
      float ext = cb.hi[best_axis] - cb.lo[best_axis];
      float scale = (float)kBins / ext;
      float clo = cb.lo[best_axis];
      int ax = best_axis, bin = best_bin;
      Prim* it = std::partition(P + lo, P + hi, [=](const Prim& p) {
        int b = (int)((p.c[ax] - clo) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        return b <= bin;
      });
      mid = (int)(it - P);
    } else {
      mid = lo;   // all centroids coincide
    }
    if (mid == lo || mid == hi || depth > 56) {
      if (n <= kLeafHardMax && best_axis < 0) return make_leaf(lo, hi);
      mid = (lo + hi) / 2;   // median by current order: keeps the tree finite for degenerate input
    }
    return emit(lo, mid, hi, depth);
  }

  // the node over prims[lo, mid) | prims[mid, hi): builds both subtrees (on another thread when the node is big and one is idle) and writes their padded boxes
  int32_t emit(int lo, int mid, int hi, int depth) {
    const int n = hi - lo;
    int32_t me = (int32_t)(nodes.size() / kBlasNodeFloats);
    nodes.resize(nodes.size() + kBlasNodeFloats, 0.0f);
    Box b0, b1;
    int32_t c0, c1;
    bool fork = false;
    if (fork_levels > 0 && n >= kForkMinPrims && spare) {
      if (spare->fetch_sub(1) > 0) fork = true; else spare->fetch_add(1);
    }
    if (fork) {
      // big node near the root: its two subtrees are built concurrently in builders of their own and appended in the order
      // a sequential build would have produced them (node order and leaf order are unchanged)
      Builder L, R;
      L.P = R.P = P; L.pad = R.pad = pad; L.fork_levels = R.fork_levels = fork_levels - 1; L.spare = R.spare = spare;
      int32_t l0 = 0, l1 = 0;
      std::thread t([&]() { l0 = L.build(lo, mid, depth + 1, b0); });
      l1 = R.build(mid, hi, depth + 1, b1);
      t.join();
      spare->fetch_add(1);
      c0 = append(L, l0);
      c1 = append(R, l1);
    } else {
      c0 = build(lo, mid, depth + 1, b0);
      c1 = build(mid, hi, depth + 1, b1);
    }
    float* nd = nodes.data() + (size_t)me * kBlasNodeFloats;
    for (int k = 0; k < 3; k++) {
      nd[k] = b0.lo[k] - pad; nd[3 + k] = b0.hi[k] + pad;
      nd[6 + k] = b1.lo[k] - pad; nd[9 + k] = b1.hi[k] + pad;
    }
    nd[12] = urt::bits_f((uint32_t)c0);
    nd[13] = urt::bits_f((uint32_t)c1);
    return me;
  }
};

}  // namespace

void set_blas_leaf_max(int n) { g_leaf_max = std::min(std::max(n, 1), kLeafHardMax); }
int get_blas_leaf_max() { return g_leaf_max; }

bool build_blas(const uint8_t* mesh_objects, int n_meshes, const float* vertices, int n_vertices, const int32_t* indices,
                int n_indices, const float* normals, int n_normals, BlasResult& out, std::string& err, BlasCache* cache) {
  out = BlasResult();
  out.mesh_root.assign((size_t)n_meshes, kEmptyMeshRoot);
  out.mesh_first_tri.assign((size_t)n_meshes, 0);
  // one independent build per MeshObject, spread over the host's cores; concatenated in MeshObject order afterwards, so the
  // result does not depend on the number of threads
  std::vector<Builder> builds((size_t)n_meshes);
  int n_threads_total = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if (n_indices < 30000) n_threads_total = 1;                       // small scenes: not worth starting threads
  if (const char* e = std::getenv("URT_BLAS_THREADS")) { int v = std::atoi(e); if (v >= 1) n_threads_total = std::min(v, 64); }   // tests
  // threads that are not busy with a MeshObject of their own go to the big subtrees of the ones still being built
  const int fork_levels = n_threads_total > 1 ? 4 : 0;
  std::atomic<int> spare(0);
  auto build_one = [&](int m) {
    Builder& B = builds[(size_t)m];
    urt_MeshObject mo;
    std::memcpy(&mo, mesh_objects + (size_t)m * sizeof(urt_MeshObject), sizeof mo);
    long off = mo.indices_offset, cnt = mo.indices_count;
    if (off < 0 || cnt < 0 || off + cnt > n_indices) {
      B.err = "MeshObject " + std::to_string(m) + ": indices_offset/count outside _Indices";
      return;
    }
    // validate the index slots; with a cache, hash what the BVH depends on in the same pass
    uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x9e3779b97f4a7c15ull;
    auto mix = [&](uint32_t w) {
      h1 = (h1 ^ w) * 0x100000001b3ull;
      h2 = (h2 ^ ((uint64_t)w * 0xff51afd7ed558ccdull)); h2 = (h2 << 27 | h2 >> 37) * 0xc4ceb9fe1a85ec53ull + 0x52dce729ull;
    };
    mix((uint32_t)g_leaf_max); mix((uint32_t)(cnt / 3));
    for (int k = 0; k < 16; k++) mix(urt::f_bits(mo.localToWorldMatrix[k]));
    for (long i = off; i + 2 < off + cnt; i += 3)
      for (int j = 0; j < 3; j++) {
        int32_t vi = indices[i + j];
        if (vi < 0 || vi >= n_vertices || (normals && vi >= n_normals)) {
          B.err = "_Indices[" + std::to_string(i + j) + "] = " + std::to_string(vi) + " is outside _Vertices/_Normals";
          return;
        }
        if (cache) { const float* v = vertices + 3 * (size_t)vi; mix(urt::f_bits(v[0])); mix(urt::f_bits(v[1])); mix(urt::f_bits(v[2])); }
      }
    if (cache) {
      std::lock_guard<std::mutex> g(cache->lock);
      auto it = cache->map.find(h1);
      if (it != cache->map.end() && it->second.check == h2) {
        BlasCacheEntry& e = it->second;
        e.last_use = cache->generation;
        cache->hits++;
        B.nodes = e.nodes;
        B.tri_slot.resize(e.tri_rel.size());
        for (size_t k = 0; k < e.tri_rel.size(); k++) B.tri_slot[k] = e.tri_rel[k] + (int32_t)off;
        B.root = e.root; B.max_depth = e.max_depth;
        return;
      }
    }
    float ext = 0;
    B.prims.reserve((size_t)(cnt / 3));
    for (long i = off; i + 2 < off + cnt; i += 3) {
      Prim p; p.slot = (int32_t)i;
      for (int k = 0; k < 3; k++) { p.lo[k] = std::numeric_limits<float>::infinity(); p.hi[k] = -p.lo[k]; }
      for (int j = 0; j < 3; j++) {
        const float* v = vertices + 3 * (size_t)indices[i + j];
        urt::v3 w = urt::mul_m4(mo.localToWorldMatrix, v[0], v[1], v[2], 1.0f);
        const float wv[3] = {w.x, w.y, w.z};
        for (int k = 0; k < 3; k++) {
          p.lo[k] = std::min(p.lo[k], wv[k]); p.hi[k] = std::max(p.hi[k], wv[k]);
          if (std::isfinite(wv[k])) ext = std::max(ext, std::fabs(wv[k]));
        }
      }
      for (int k = 0; k < 3; k++) p.c[k] = 0.5f * p.lo[k] + 0.5f * p.hi[k];
      B.prims.push_back(p);
    }
    if (B.prims.empty()) return;
    B.pad = ext * 1.52587890625e-5f + 1e-30f;
    Box root;
    B.P = B.prims.data();
    B.fork_levels = fork_levels; B.spare = &spare;
    B.root = B.build(0, (int)B.prims.size(), 1, root);
    std::vector<Prim>().swap(B.prims);
    if (cache) {
      BlasCacheEntry e;
      e.check = h2; e.nodes = B.nodes; e.root = B.root; e.max_depth = B.max_depth;
      e.tri_rel.resize(B.tri_slot.size());
      for (size_t k = 0; k < B.tri_slot.size(); k++) e.tri_rel[k] = B.tri_slot[k] - (int32_t)off;
      std::lock_guard<std::mutex> g(cache->lock);
      e.last_use = cache->generation;
      cache->builds++;
      cache->map[h1] = std::move(e);
    }
  };
  if (cache) { std::lock_guard<std::mutex> g(cache->lock); cache->generation++; }
  {
    int n_threads = std::min(n_threads_total, std::max(1, n_meshes));
    std::atomic<int> next_mesh(0);
    spare.store(n_threads_total - n_threads);                  // threads beyond one per MeshObject worker
    auto worker = [&]() { for (int m; (m = next_mesh.fetch_add(1)) < n_meshes;) build_one(m); spare.fetch_add(1); };   // an idle worker's slot is up for grabs
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; t++) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }
  for (int m = 0; m < n_meshes; m++) {
    Builder& B = builds[(size_t)m];
    if (!B.err.empty()) { err = B.err; return false; }         // the first MeshObject in error, as a sequential build reports
    int32_t node_base = (int32_t)(out.nodes.size() / kBlasNodeFloats);
    uint32_t tri_base = (uint32_t)out.tri_slot.size();
    out.mesh_first_tri[(size_t)m] = (int32_t)tri_base;
    auto rebase = [&](int32_t code) -> int32_t {
      if (code == kEmptyMeshRoot) return code;
      if (code >= 0) return code + node_base;
      uint32_t c = ~(uint32_t)code;
      return (int32_t)~((((c >> 3) + tri_base) << 3) | (c & 7u));
    };
    size_t at = out.nodes.size();
    out.nodes.insert(out.nodes.end(), B.nodes.begin(), B.nodes.end());
    for (size_t k = at; k < out.nodes.size(); k += kBlasNodeFloats)
      for (int c = 0; c < 2; c++) out.nodes[k + 12 + c] = urt::bits_f((uint32_t)rebase((int32_t)urt::f_bits(out.nodes[k + 12 + c])));
    out.tri_slot.insert(out.tri_slot.end(), B.tri_slot.begin(), B.tri_slot.end());
    out.tri_mesh.insert(out.tri_mesh.end(), B.tri_slot.size(), m);
    out.mesh_root[(size_t)m] = rebase(B.root);
    out.max_depth = std::max(out.max_depth, B.max_depth);
    B = Builder();
  }
  if (cache) {                       // keep only what this scene uses
    std::lock_guard<std::mutex> g(cache->lock);
    for (auto it = cache->map.begin(); it != cache->map.end();) { if (it->second.last_use != cache->generation) it = cache->map.erase(it); else ++it; }
  }
  // Renumber the interior nodes so that the first kTopOrderNodes indices are the TOP of the forest in breadth-first order
  // (all roots, then their children, ...): the phase-scheduled kernel keeps nodes [0, T) in LDS and walks them before a ray
  // joins the wave-wide traversal loop (kernels.hip trace_front).  Deeper nodes keep their depth-first order (subtree
  // locality).  Only indices change: every ray visits the same nodes in the same order as before.
  {
    size_t nn = out.nodes.size() / kBlasNodeFloats;
    std::vector<int32_t> new_of(nn, -1);
    std::vector<int32_t> queue;
    for (int m = 0; m < n_meshes; m++) { int32_t r = out.mesh_root[(size_t)m]; if (r >= 0 && r != kEmptyMeshRoot) queue.push_back(r); }
    size_t head = 0; int32_t next_id = 0;
    while (head < queue.size() && next_id < kTopOrderNodes) {
      int32_t o = queue[head++];
      new_of[(size_t)o] = next_id++;
      const float* nd = out.nodes.data() + (size_t)o * kBlasNodeFloats;
      for (int c = 0; c < 2; c++) { int32_t ch = (int32_t)urt::f_bits(nd[12 + c]); if (ch >= 0) queue.push_back(ch); }
    }
    for (size_t o = 0; o < nn; o++) if (new_of[o] < 0) new_of[o] = next_id++;
    std::vector<float> moved(out.nodes.size());
    for (size_t o = 0; o < nn; o++) {
      float* dst = moved.data() + (size_t)new_of[o] * kBlasNodeFloats;
      std::memcpy(dst, out.nodes.data() + o * kBlasNodeFloats, sizeof(float) * kBlasNodeFloats);
      for (int c = 0; c < 2; c++) { int32_t ch = (int32_t)urt::f_bits(dst[12 + c]); if (ch >= 0) dst[12 + c] = urt::bits_f((uint32_t)new_of[(size_t)ch]); }
    }
    out.nodes.swap(moved);
    for (int m = 0; m < n_meshes; m++) { int32_t& r = out.mesh_root[(size_t)m]; if (r >= 0 && r != kEmptyMeshRoot) r = new_of[(size_t)r]; }
  }
  // leaf-ordered triangle and normal records
  size_t nt = out.tri_slot.size();
  out.tri_verts.assign(nt * 12, 0.0f);
  out.tri_norms.assign(nt * 12, 0.0f);
  auto records = [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; k++) {
      int32_t i = out.tri_slot[k];
      int32_t m = out.tri_mesh[k];
      urt_MeshObject mo;
      std::memcpy(&mo, mesh_objects + (size_t)m * sizeof(urt_MeshObject), sizeof mo);
      urt::v3 w[3];
      for (int j = 0; j < 3; j++) {
        const float* v = vertices + 3 * (size_t)indices[i + j];
        w[j] = urt::mul_m4(mo.localToWorldMatrix, v[0], v[1], v[2], 1.0f);      // RS:244-246
      }
      urt::v3 e1 = w[1] - w[0], e2 = w[2] - w[0];                                // RS:201-202
      float* tv = out.tri_verts.data() + 12 * k;
      tv[0] = w[0].x; tv[1] = w[0].y; tv[2] = w[0].z; tv[3] = urt::bits_f((uint32_t)i);
      tv[4] = e1.x; tv[5] = e1.y; tv[6] = e1.z; tv[7] = urt::bits_f((uint32_t)m);
      tv[8] = e2.x; tv[9] = e2.y; tv[10] = e2.z; tv[11] = 0.0f;
      float* tn = out.tri_norms.data() + 12 * k;
      if (normals) {
        for (int j = 0; j < 3; j++) {
          const float* nn = normals + 3 * (size_t)indices[i + j];              // RS:259-261
          tn[4 * j] = nn[0]; tn[4 * j + 1] = nn[1]; tn[4 * j + 2] = nn[2];
        }
      }
    }
  };
  {
    int n_threads = nt >= 20000 ? (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u) : 1;
    if (const char* e = std::getenv("URT_BLAS_THREADS")) { int v = std::atoi(e); if (v >= 1) n_threads = std::min(v, 64); }
    size_t chunk = (nt + (size_t)n_threads - 1) / (size_t)n_threads;
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; t++) pool.emplace_back(records, std::min(nt, (size_t)t * chunk), std::min(nt, (size_t)(t + 1) * chunk));
    records(0, std::min(nt, chunk));
    for (auto& th : pool) th.join();
  }
  return true;
}

}  // namespace urtd
