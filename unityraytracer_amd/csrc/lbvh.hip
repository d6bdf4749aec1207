// lbvh.hip — triangle-BVH build ON THE GPU for dynamic scenes ("GPU BLAS build (LBVH)").
//
// The reference re-uploads every buffer whenever one object moves (RM:215-230 -> RM:262-336 -> SetData) and has no
// triangle-level structure at all (RS:243 tests every triangle).  The default builder of this library (blas_builder.cpp,
// binned SAH on host threads) gives the best trees but costs 14-65 ms for the BASELINE scenes; with
// urt_set_option("blas_builder", 1) the BVH is built here instead, from device copies of _MeshObjects / _Vertices /
// _Indices / _Normals exactly as SetData delivered them:
//   1. k_tri_bounds   one lane per triangle: world-space vertices by the normative mul(localToWorld, float4(v,1)) of RS:244-246
//                     (urt::mul_m4 — the same fma chain the host builder and the oracle use, so the triangle records are
//                     bit-identical), triangle box, per-mesh centroid bounds and coordinate extent (sharded atomics)
//   2. k_morton       30-bit Morton code of the centroid inside its MeshObject's centroid box; sort key = mesh id << 32 | code
//   3. rocprim::radix_sort_pairs on the 64-bit keys (a library sort; everything else here is hand-written)
//   4. k_karras       Karras 2012 binary radix tree, one independent tree per MeshObject segment (delta = -1 outside it)
//   5. k_fit          bottom-up boxes with one arrival counter per node (acquire/release at agent scope)
//   6. k_classify     subtrees of <= leaf_max triangles collapse into leaves (a node covers a contiguous range of the sorted
//                     triangles, so the leaf code is just (first, count)); depth of the deepest leaf
//   7. k_top_bfs      breadth-first numbering of the top kTopOrderNodes nodes of the forest (they live in LDS during
//                     traversal); the rest keep Karras order, compacted by a prefix sum
//   8. k_emit_nodes / k_emit_tris   64-byte two-child nodes with padded boxes, 48-byte triangle and normal records in
//                     sorted (= leaf) order — the formats of urt_device.h, so the trace kernels do not know which builder ran.
// Round 4 added two more builders on the same inputs and outputs ("blas_builder" 2 and 3; the default -1 = auto picks 3 for scenes of
// 200,000 triangles or more, context.cpp prepare_scene):
//   2  the radix tree built TOP-DOWN, one launch per level, with a depth budget (k_td_roots / k_td_level instead of k_karras): the
//      traversal stacks live in LDS, and a 30-level Karras tree costs workgroups per CU;
//   3  BINNED SAH, the host builder's algorithm level by level (k_sah_*): bins filled with LDS-privatised atomics, one thread per node
//      sweeps them, a flag + scan + scatter pass partitions every range in place — the host's trees in a fifth of the time.
// Any conservative BVH gives the same pixels: the closest-hit rule (strict t <, ties to the lower index slot, A.4) is in
// the traversal, not in the tree.  tests/test_gpu_lbvh.py checks structure, bit-identical frames and the build time of all three.
#include <hip/hip_runtime.h>
#include <string.h>                      // rocprim's texture_cache_iterator.hpp calls memset unqualified
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "../../include/urt.h"
#include "../../include/urt_math.h"
#include "blas_builder.h"
#include "lbvh.h"

namespace {

using namespace urt;
using namespace urtd;

constexpr int kStatShards = 32;          // per-mesh reduction targets are replicated: same-address atomics serialise (~88/us)

struct MeshStat { unsigned int cmin[3], cmax[3], ext, pad; };     // order-preserving uint images of floats (ext: plain bits, >= 0)

struct Dev {
  const uint8_t* mesh_objects; int n_meshes;
  const float* vertices; int n_vertices;
  const int32_t* indices; int n_indices;
  const float* normals; int n_normals;
  const int32_t* tri_first;              // [n_meshes + 1] prefix of triangles per MeshObject
  int T;                                 // triangles in total
  int leaf_max;
  // per triangle (unsorted id g)
  float4* tlo; float4* thi;              // box
  unsigned long long* keys; unsigned int* vals;             // sort input
  unsigned long long* keys_s; unsigned int* vals_s;         // sorted
  // per sorted position / internal node index
  int2* range;                           // [lo, hi] of internal node i
  int2* child;                           // child refs: >= 0 internal node, < 0 ~leaf position
  int* parent;                           // [2T]: parent of internal node i at [i], of leaf k at [T + k]; -1 = none
  unsigned int* visits;
  float4* nlo; float4* nhi;              // node boxes
  int* keep;                             // 1 = internal node that survives the leaf collapse
  int* is_top; int* top_id; int* flag; int* rank; int* queue;
  MeshStat* stats;                       // [kStatShards][n_meshes] then folded into shard 0
  int32_t* mesh_root;                    // out
  int* scalars;                          // [0] error (index slot + 1), [1] max depth, [2] n_top, [3] top-down builder: items queued for the next level
  int* root_id;                          // [n_meshes] internal-node index of the MeshObject's root (Karras: the first position of its segment)
  int4* tdq[2];                          // top-down builder (2): this level's and the next level's ranges {a, b, parent, side | depth << 1}
  int depth_cap;                         // top-down builder: deepest level a leaf may sit on
  // ---- builder 3: binned SAH on the GPU, level by level (k_sah_*) ----
  unsigned int* sidx[2];                 // [T] x 2 (ping-pong): triangle id at position p
  int* snode[2];                         // [T] x 2: index of the position's node in the level's list, or -1 (its leaf is final)
  int* sflag; int* sscan;                // [T] left-side flags of the level and their exclusive scan
  int4* sl_rng[2];                       // level node lists (ping-pong): {first, end, parent final id (-1 - mesh for a root), side | nb << 1}
  uint4* sl_cb0[2]; uint2* sl_cb1[2];    //   centroid bounds as order-preserving uints: {lo.x, lo.y, lo.z, hi.x}, {hi.y, hi.z}
  int* sl_binoff[2];                     //   first word of the node's bins
  int4* sl_dec;                          // this level's decisions: {axis (-1: halve by position), bin, n_left, child-node count}
  int* sl_cnt; int* sl_cscan; int* sl_words; int* sl_wscan;     // per node: child nodes / bin words of the next level, and their scans
  unsigned int* sbins;                   // bins: per (node, axis, bin) 7 words {count, lo.xyz, hi.xyz (ordered uints)}
  float4* nodes; float4* tri_verts; float4* tri_norms;      // out
};

__device__ __forceinline__ unsigned int f2ord(float f) { unsigned int b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned int k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }

__device__ __forceinline__ int find_mesh(const int32_t* first, int n, int g) {   // largest m with first[m] <= g (empty meshes skipped)
  int lo = 0, hi = n - 1;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (first[mid] <= g) lo = mid; else hi = mid - 1; }
  return lo;
}

struct TriWorld { v3 w[3]; int slot; int m; bool bad; };

__device__ __forceinline__ TriWorld load_tri(const Dev& D, int g) {
  TriWorld t;
  t.m = find_mesh(D.tri_first, D.n_meshes, g);
  const uint8_t* rec = D.mesh_objects + (size_t)t.m * sizeof(urt_MeshObject);
  const float* M = (const float*)rec;
  int off = *(const int*)(rec + 64);
  t.slot = off + 3 * (g - D.tri_first[t.m]);
  t.bad = false;
  for (int j = 0; j < 3; j++) {
    int vi = D.indices[t.slot + j];
    if (vi < 0 || vi >= D.n_vertices || (D.normals && vi >= D.n_normals)) { atomicMax(D.scalars, t.slot + j + 1); t.bad = true; vi = 0; }
    const float* v = D.vertices + 3 * (size_t)vi;
    t.w[j] = mul_m4(M, v[0], v[1], v[2], 1.0f);                                  // RS:244-246
  }
  return t;
}

__global__ __launch_bounds__(256) void k_init(Dev D) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kStatShards * D.n_meshes) {
    MeshStat s;
    for (int k = 0; k < 3; k++) { s.cmin[k] = 0xffffffffu; s.cmax[k] = 0u; }
    s.ext = 0u; s.pad = 0u;
    D.stats[i] = s;
  }
  if (i < 4) D.scalars[i] = 0;
  for (int k = i; k < 2 * D.T; k += gridDim.x * blockDim.x) D.parent[k] = -1;
  for (int k = i; k < D.T; k += gridDim.x * blockDim.x) { D.visits[k] = 0u; D.keep[k] = 0; D.is_top[k] = 0; D.top_id[k] = 0; }
}

__global__ __launch_bounds__(256) void k_tri_bounds(Dev D) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= D.T) return;
  TriWorld t = load_tri(D, g);
  float lo[3], hi[3];
  const float w[3][3] = {{t.w[0].x, t.w[0].y, t.w[0].z}, {t.w[1].x, t.w[1].y, t.w[1].z}, {t.w[2].x, t.w[2].y, t.w[2].z}};
  float ext = 0.0f;
  for (int k = 0; k < 3; k++) {
    lo[k] = f_min(f_min(w[0][k], w[1][k]), w[2][k]);
    hi[k] = f_max(f_max(w[0][k], w[1][k]), w[2][k]);
    for (int j = 0; j < 3; j++) { float a = f_abs(w[j][k]); if (a < URT_INF) ext = f_max(ext, a); }     // finite coordinates only
  }
  D.tlo[g] = make_float4(lo[0], lo[1], lo[2], 0.0f);
  D.thi[g] = make_float4(hi[0], hi[1], hi[2], 0.0f);
  MeshStat* s = D.stats + (size_t)(blockIdx.x % kStatShards) * D.n_meshes + t.m;
  unsigned int cmin[3], cmax[3], e = __float_as_uint(ext);
  for (int k = 0; k < 3; k++) {
    float c = 0.5f * lo[k] + 0.5f * hi[k];
    bool ok = c == c;
    cmin[k] = ok ? f2ord(c) : 0xffffffffu; cmax[k] = ok ? f2ord(c) : 0u;
  }
  // consecutive triangles nearly always belong to one MeshObject: reduce in the wave, one set of atomics per wave
  int m0 = __shfl(t.m, 0, 64);
  bool uniform = __ballot(t.m != m0) == 0 && __popcll(__ballot(1)) == 64;
  if (uniform) {
    for (int off = 32; off > 0; off >>= 1) {
      for (int k = 0; k < 3; k++) { cmin[k] = min(cmin[k], (unsigned int)__shfl_xor((int)cmin[k], off, 64)); cmax[k] = max(cmax[k], (unsigned int)__shfl_xor((int)cmax[k], off, 64)); }
      e = max(e, (unsigned int)__shfl_xor((int)e, off, 64));
    }
    if ((threadIdx.x & 63) != 0) return;
  }
  for (int k = 0; k < 3; k++) { atomicMin(&s->cmin[k], cmin[k]); atomicMax(&s->cmax[k], cmax[k]); }
  atomicMax(&s->ext, e);
}

__global__ __launch_bounds__(64) void k_fold_stats(Dev D) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= D.n_meshes) return;
  MeshStat a = D.stats[m];
  for (int sh = 1; sh < kStatShards; sh++) {
    MeshStat b = D.stats[(size_t)sh * D.n_meshes + m];
    for (int k = 0; k < 3; k++) { a.cmin[k] = min(a.cmin[k], b.cmin[k]); a.cmax[k] = max(a.cmax[k], b.cmax[k]); }
    a.ext = max(a.ext, b.ext);
  }
  D.stats[m] = a;
  D.root_id[m] = D.tri_first[m];                           // (the Karras tree's root; the top-down builder overwrites it)
  // MeshObjects without a tree of their own
  int n = D.tri_first[m + 1] - D.tri_first[m];
  if (n == 0) D.mesh_root[m] = kEmptyMeshRoot;
  else if (n == 1) { D.mesh_root[m] = (int32_t)~(((uint32_t)D.tri_first[m] << 3) | 0u); atomicMax(D.scalars + 1, 1); }
}

__device__ __forceinline__ unsigned int spread10(unsigned int v) {   // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ __launch_bounds__(256) void k_morton(Dev D) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= D.T) return;
  int m = find_mesh(D.tri_first, D.n_meshes, g);
  MeshStat s = D.stats[m];
  float4 lo = D.tlo[g], hi = D.thi[g];
  const float c[3] = {0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z};
  unsigned int q[3];
  for (int k = 0; k < 3; k++) {
    float a = ord2f(s.cmin[k]), b = ord2f(s.cmax[k]);
    float x = (c[k] - a) / (b - a) * 1024.0f;            // NaN (degenerate extent, non-finite centroid) lands in cell 0
    q[k] = x >= 0.0f ? (unsigned int)f_min(x, 1023.0f) : 0u;
  }
  unsigned int code = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
  D.keys[g] = ((unsigned long long)(unsigned int)m << 32) | code;
  D.vals[g] = (unsigned int)g;
}

__device__ __forceinline__ int delta(const unsigned long long* keys, int i, int j, int lo, int hi) {
  if (j < lo || j > hi) return -1;
  unsigned long long a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((unsigned int)(i ^ j));   // equal codes: the position breaks the tie (keys become unique)
  return __clzll((long long)(a ^ b));
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (HPG 2012), restricted to the segment
// [lo, hi] of the MeshObject the node belongs to.  Internal node i exists for lo <= i < hi; the segment's root is node lo.
__global__ __launch_bounds__(256) void k_karras(Dev D) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.T) return;
  int m = (int)(D.keys_s[i] >> 32);
  int lo = D.tri_first[m], hi = D.tri_first[m + 1] - 1;
  if (i >= hi) return;                                     // the last position of a segment has no internal node
  const unsigned long long* K = D.keys_s;
  int d = delta(K, i, i + 1, lo, hi) - delta(K, i, i - 1, lo, hi) >= 0 ? 1 : -1;
  int dmin = delta(K, i, i - d, lo, hi);
  int lmax = 2;
  while (delta(K, i, i + lmax * d, lo, hi) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(K, i, i + (l + t) * d, lo, hi) > dmin) l += t;
  int j = i + l * d;
  int dnode = delta(K, i, j, lo, hi);
  int s = 0, t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(K, i, i + (s + t) * d, lo, hi) > dnode) s += t;
  } while (t > 1);
  int gamma = i + s * d + min(d, 0);
  int a = min(i, j), b = max(i, j);
  int left = a == gamma ? ~gamma : gamma;
  int right = b == gamma + 1 ? ~(gamma + 1) : gamma + 1;
  D.range[i] = make_int2(a, b);
  D.child[i] = make_int2(left, right);
  D.parent[left >= 0 ? left : D.T + ~left] = i;
  D.parent[right >= 0 ? right : D.T + ~right] = i;
}


// ---- builder 2: the same radix tree built TOP-DOWN with a depth budget -------------------------------------------------------------
// Karras' tree splits every range at its highest differing Morton bit, however lopsided: 30 levels on a 70 k-triangle mesh, against the
// 21 of the host's SAH tree.  The traversal stacks live in LDS, one entry per level per lane, and their size decides how many workgroups
// a CU holds: the depth ALONE costs +6 % (C3) ... +17 % (C4) frame time at equal tree quality (profiles/r04_logs/r4_ab_ploc_upper_tree.log,
// stack_pad).  Here a range keeps its radix split only while the bigger side can still be finished by MEDIAN splits (in Morton order)
// within the budget `depth_cap` = levels a median tree of the biggest MeshObject needs + 6; otherwise it is halved.  One launch per
// level; a node's index is its split position (unique per internal node of a binary tree over the sorted positions), so the arrays the
// rest of the pipeline reads — range, child, parent — come out in Karras' conventions.
__device__ __forceinline__ int median_levels(int n, int leaf_max) {   // levels (interior + the leaf level) of a median-split tree over n triangles
  int h = 1;
  while (n > leaf_max) { n = (n + 1) >> 1; h++; }
  return h;
}

__global__ __launch_bounds__(64) void k_td_roots(Dev D) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= D.n_meshes) return;
  int lo = D.tri_first[m], n = D.tri_first[m + 1] - lo;
  if (n < 2) return;
  int at = atomicAdd(D.scalars + 3, 1);
  D.tdq[0][at] = make_int4(lo, lo + n - 1, -1 - m, 1 << 1);   // parent < 0: the root of MeshObject -1 - parent; depth 1
}

__global__ __launch_bounds__(256) void k_td_level(Dev D, int src, int n_items) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_items) return;
  int4 it = D.tdq[src][q];
  const int a = it.x, b = it.y, depth = it.w >> 1, side = it.w & 1;
  const unsigned long long* K = D.keys_s;
  // radix split: the last position whose key shares more than delta(a, b) leading bits with key a (Karras' search, from the left end)
  int gamma;
  {
    const int dnode = delta(K, a, b, a, b);
    int s = 0, t = b - a;
    do {
      t = (t + 1) >> 1;
      if (a + s + t <= b && delta(K, a, a + s + t, a, b) > dnode) s += t;
    } while (t > 1);
    gamma = a + s;
  }
  const int size = b - a + 1;
  int nl = gamma - a + 1, nr = size - nl;
  if (median_levels(max(nl, nr), D.leaf_max) > D.depth_cap - depth) { nl = (size + 1) >> 1; nr = size - nl; gamma = a + nl - 1; }   // over budget: halve (Morton order)
  const int i = gamma;                                       // the node's index
  D.range[i] = make_int2(a, b);
  D.parent[i] = it.z >= 0 ? it.z : -1;
  if (it.z >= 0) { if (side) D.child[it.z].y = i; else D.child[it.z].x = i; }
  else D.root_id[-1 - it.z] = i;
  int2 ch;
  ch.x = nl == 1 ? ~a : 0; ch.y = nr == 1 ? ~b : 0;          // (interior children write themselves in at the next level)
  D.child[i] = ch;
  if (nl == 1) D.parent[D.T + a] = i;
  if (nr == 1) D.parent[D.T + b] = i;
  int n_new = (nl > 1) + (nr > 1);
  if (n_new) {
    int at = atomicAdd(D.scalars + 3, n_new);                // (queue order varies from run to run; the tree does not)
    if (nl > 1) D.tdq[src ^ 1][at++] = make_int4(a, gamma, i, (0) | ((depth + 1) << 1));
    if (nr > 1) D.tdq[src ^ 1][at] = make_int4(gamma + 1, b, i, (1) | ((depth + 1) << 1));
  }
}

__device__ __forceinline__ void child_box(const Dev& D, int c, float4& lo, float4& hi) {
  if (c >= 0) { lo = D.nlo[c]; hi = D.nhi[c]; }
  else { unsigned int g = D.vals_s[~c]; lo = D.tlo[g]; hi = D.thi[g]; }
}

__global__ __launch_bounds__(256) void k_fit(Dev D) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= D.T) return;
  int node = D.parent[D.T + k];
  while (node >= 0) {
    // the first of the two children to arrive stops; the second one finds both boxes complete (release by the first's
    // fetch_add, acquire by the second's) and moves up
    unsigned int seen = __hip_atomic_fetch_add(&D.visits[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (seen == 0u) return;
    int2 ch = D.child[node];
    float4 l0, h0, l1, h1;
    child_box(D, ch.x, l0, h0);
    child_box(D, ch.y, l1, h1);
    D.nlo[node] = make_float4(f_min(l0.x, l1.x), f_min(l0.y, l1.y), f_min(l0.z, l1.z), 0.0f);
    D.nhi[node] = make_float4(f_max(h0.x, h1.x), f_max(h0.y, h1.y), f_max(h0.z, h1.z), 0.0f);
    node = D.parent[node];
  }
}

__device__ __forceinline__ int32_t leaf_code(int first, int count) { return (int32_t)~(((uint32_t)first << 3) | (uint32_t)(count - 1)); }

__global__ __launch_bounds__(256) void k_classify(Dev D) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool keep = false;
  if (i < D.T) {
    int m = (int)(D.keys_s[i] >> 32);
    int lo = D.tri_first[m], hi = D.tri_first[m + 1] - 1;
    if (i < hi) {
      int2 r = D.range[i];
      int size = r.y - r.x + 1;
      keep = size > D.leaf_max;
      D.keep[i] = keep ? 1 : 0;
      if (i == D.root_id[m] && !keep) { D.mesh_root[m] = leaf_code(lo, size); atomicMax(D.scalars + 1, 1); }   // the whole MeshObject is one leaf
    }
  }
  int depth = 0;
  if (keep) {
    depth = 2;                                              // the node itself + the leaf level below it
    for (int p = D.parent[i]; p >= 0; p = D.parent[p]) depth++;
  }
  for (int off = 32; off > 0; off >>= 1) depth = max(depth, __shfl_xor(depth, off, 64));     // one atomic per wave
  if ((threadIdx.x & 63) == 0 && depth > 0) atomicMax(D.scalars + 1, depth);
}

// Breadth-first numbering of the top of the forest (roots of all MeshObjects in MeshObject order, then their children, ...):
// node indices [0, n_top) — the part of the forest the trace kernel keeps in LDS.  One lane; <= kTopOrderNodes pops.
__global__ __launch_bounds__(64) void k_top_bfs(Dev D) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int head = 0, tail = 0, next = 0;
  for (int m = 0; m < D.n_meshes; m++) {
    int lo = D.tri_first[m], n = D.tri_first[m + 1] - lo;
    if (n >= 2 && D.keep[D.root_id[m]]) D.queue[tail++] = D.root_id[m];
  }
  while (head < tail && next < kTopOrderNodes) {
    int o = D.queue[head++];
    D.top_id[o] = next++;
    D.is_top[o] = 1;
    int2 ch = D.child[o];
    if (ch.x >= 0 && D.keep[ch.x]) D.queue[tail++] = ch.x;
    if (ch.y >= 0 && D.keep[ch.y]) D.queue[tail++] = ch.y;
  }
  D.scalars[2] = next;
}

__global__ __launch_bounds__(256) void k_flags(Dev D) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.T) D.flag[i] = (D.keep[i] && !D.is_top[i]) ? 1 : 0;
}

__device__ __forceinline__ int new_id(const Dev& D, int i) { return D.is_top[i] ? D.top_id[i] : D.scalars[2] + D.rank[i]; }

__global__ __launch_bounds__(256) void k_emit_nodes(Dev D) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.T || !D.keep[i]) return;
  int m = (int)(D.keys_s[i] >> 32);
  float pad = __uint_as_float(D.stats[m].ext) * 1.52587890625e-5f + 1e-30f;    // as blas_builder.cpp: 2^-16 of the mesh's extent
  int2 ch = D.child[i];
  int32_t code[2];
  float4 lo[2], hi[2];
  const int c2[2] = {ch.x, ch.y};
  for (int k = 0; k < 2; k++) {
    int c = c2[k];
    child_box(D, c, lo[k], hi[k]);
    if (c < 0) code[k] = leaf_code(~c, 1);
    else if (D.keep[c]) code[k] = new_id(D, c);
    else { int2 r = D.range[c]; code[k] = leaf_code(r.x, r.y - r.x + 1); }
  }
  int id = new_id(D, i);
  float4* nd = D.nodes + 4 * (size_t)id;
  nd[0] = make_float4(lo[0].x - pad, lo[0].y - pad, lo[0].z - pad, hi[0].x + pad);
  nd[1] = make_float4(hi[0].y + pad, hi[0].z + pad, lo[1].x - pad, lo[1].y - pad);
  nd[2] = make_float4(lo[1].z - pad, hi[1].x + pad, hi[1].y + pad, hi[1].z + pad);
  nd[3] = make_float4(__int_as_float(code[0]), __int_as_float(code[1]), 0.0f, 0.0f);
  if (i == D.root_id[m]) D.mesh_root[m] = id;
}

__global__ __launch_bounds__(256) void k_emit_tris(Dev D) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= D.T) return;
  int g = (int)D.vals_s[k];
  TriWorld t = load_tri(D, g);
  v3 e1 = t.w[1] - t.w[0], e2 = t.w[2] - t.w[0];                                  // RS:201-202
  float4* tv = D.tri_verts + 3 * (size_t)k;
  tv[0] = make_float4(t.w[0].x, t.w[0].y, t.w[0].z, __int_as_float(t.slot));
  tv[1] = make_float4(e1.x, e1.y, e1.z, __int_as_float(t.m));
  tv[2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
  float4* tn = D.tri_norms + 3 * (size_t)k;
  for (int j = 0; j < 3; j++) {
    float4 n = make_float4(0, 0, 0, 0);
    if (D.normals && !t.bad) { const float* p = D.normals + 3 * (size_t)D.indices[t.slot + j]; n = make_float4(p[0], p[1], p[2], 0.0f); }   // RS:259-261
    tn[j] = n;
  }
}


// =====================================================================================================================================
// builder 3: BINNED SAH ON THE GPU.  The host builder's algorithm (blas_builder.cpp Builder::build: per range the centroid bounds, 32
// bins per axis over them, the cheapest of the 3 x 31 planes by A(L) n_L + A(R) n_R, halving by position when no plane separates)
// run level by level over all MeshObjects at once: one pass bins every triangle of the level into its node's bins with atomics (ordered-
// uint min / max), one thread per node sweeps the bins and picks the plane, a flag + scan + scatter pass partitions the ranges in place
// (stable), and the scatter pass also gathers the children's centroid bounds for the next level.  Nodes are numbered level by level —
// which IS the breadth-first top-of-forest order the trace kernel keeps in LDS.  Nodes of fewer than 64 triangles use 8 bins per axis
// (they cannot fill more), which bounds the bin memory by 72 words per triangle.
// =====================================================================================================================================
constexpr int kSahBinsBig = 32, kSahBinsSmall = 8, kSahBigNode = 64;
__device__ __forceinline__ int sah_nb(int size) { return size >= kSahBigNode ? kSahBinsBig : kSahBinsSmall; }
__device__ __forceinline__ float3 tri_centroid(const Dev& D, unsigned int g) {
  float4 lo = D.tlo[g], hi = D.thi[g];
  return make_float3(0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z);
}
__device__ __forceinline__ int sah_bin(float c, float lo, float ext, int nb) {       // blas_builder.cpp: (int)((c - lo) * (bins / ext)), clamped
  int b = (int)((c - lo) * ((float)nb / ext));
  return min(max(b, 0), nb - 1);
}

__global__ __launch_bounds__(64) void k_sah_roots(Dev D) {        // one lane: level 0 = the MeshObjects that need a tree, in MeshObject order
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int n = 0, words = 0;
  for (int m = 0; m < D.n_meshes; m++) {
    int lo = D.tri_first[m], size = D.tri_first[m + 1] - lo;
    if (size <= 0) continue;
    if (size <= D.leaf_max) { D.mesh_root[m] = leaf_code(lo, size); atomicMax(D.scalars + 1, 1); continue; }
    int nb = sah_nb(size);
    D.sl_rng[0][n] = make_int4(lo, lo + size, -1 - m, nb << 1);
    MeshStat s = D.stats[m];
    D.sl_cb0[0][n] = make_uint4(s.cmin[0], s.cmin[1], s.cmin[2], s.cmax[0]);
    D.sl_cb1[0][n] = make_uint2(s.cmax[1], s.cmax[2]);
    D.sl_binoff[0][n] = words;
    words += 3 * nb * 7;
    D.mesh_root[m] = n;                                       // level 0 starts at node 0
    n++;
  }
  D.scalars[3] = n; D.scalars[2] = words;
}

__global__ __launch_bounds__(256) void k_sah_positions(Dev D) {   // level 0: identity order, every position belongs to its MeshObject's root
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= D.T) return;
  D.sidx[0][p] = (unsigned int)p;
  int m = find_mesh(D.tri_first, D.n_meshes, p);
  int size = D.tri_first[m + 1] - D.tri_first[m];
  D.snode[0][p] = size > D.leaf_max ? D.mesh_root[m] : -1;
}

__global__ __launch_bounds__(256) void k_sah_bins_init(Dev D, int words) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= words) return;
  int k = w % 7;
  D.sbins[w] = k == 0 ? 0u : (k <= 3 ? 0xffffffffu : 0u);     // count, lo (ordered +max), hi (ordered min)
}

// Binning.  The positions of a node are contiguous, so a workgroup whose first and last position belong to one node bins into a
// private copy in LDS and merges its non-empty bins into the node's with one atomic per word (near the root that is 8x fewer global
// atomics, and none of the same-address pile-ups: a level of C5 took 1.6 ms with per-lane atomics); workgroups that straddle
// nodes (deep levels: many small nodes, little contention) use the global atomics directly.
__global__ __launch_bounds__(256) void k_sah_bin(Dev D, int src) {
  __shared__ unsigned int sb[3 * kSahBinsBig * 7];
  const int p0 = blockIdx.x * blockDim.x, p1 = min(p0 + (int)blockDim.x, D.T) - 1;
  const int nd0 = D.snode[src][p0];
  const bool uniform = nd0 >= 0 && D.snode[src][p1] == nd0;              // (workgroup-uniform)
  int p = p0 + threadIdx.x;
  int nd = p < D.T ? D.snode[src][p] : -1;
  int nb_u = 0;
  if (uniform) {
    nb_u = D.sl_rng[src][nd0].w >> 1;
    for (int w = threadIdx.x; w < 3 * nb_u * 7; w += blockDim.x) { int k = w % 7; sb[w] = k == 0 ? 0u : (k <= 3 ? 0xffffffffu : 0u); }
    __syncthreads();
  }
  if (nd >= 0) {
    unsigned int g = D.sidx[src][p];
    float4 lo = D.tlo[g], hi = D.thi[g];
    const float c[3] = {0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z};
    uint4 b0 = D.sl_cb0[src][nd]; uint2 b1 = D.sl_cb1[src][nd];
    const float clo[3] = {ord2f(b0.x), ord2f(b0.y), ord2f(b0.z)}, chi[3] = {ord2f(b0.w), ord2f(b1.x), ord2f(b1.y)};
    const int nb = D.sl_rng[src][nd].w >> 1;
    unsigned int* B = uniform ? sb : D.sbins + D.sl_binoff[src][nd];
    const unsigned int olo[3] = {f2ord(lo.x), f2ord(lo.y), f2ord(lo.z)}, ohi[3] = {f2ord(hi.x), f2ord(hi.y), f2ord(hi.z)};
    for (int ax = 0; ax < 3; ax++) {
      float ext = chi[ax] - clo[ax];
      if (!(ext > 0.0f)) continue;
      unsigned int* q = B + (ax * nb + sah_bin(c[ax], clo[ax], ext, nb)) * 7;
      atomicAdd(q, 1u);
      atomicMin(q + 1, olo[0]); atomicMin(q + 2, olo[1]); atomicMin(q + 3, olo[2]);
      atomicMax(q + 4, ohi[0]); atomicMax(q + 5, ohi[1]); atomicMax(q + 6, ohi[2]);
    }
  }
  if (uniform) {
    __syncthreads();
    unsigned int* G = D.sbins + D.sl_binoff[src][nd0];
    for (int w = threadIdx.x; w < 3 * nb_u * 7; w += blockDim.x) {
      const int k = w % 7;
      const unsigned int v = sb[w];
      if (k == 0) { if (v) atomicAdd(G + w, v); }
      else if (k <= 3) { if (v != 0xffffffffu) atomicMin(G + w, v); }
      else if (v != 0u) atomicMax(G + w, v);
    }
  }
}

struct SBox { float lo[3], hi[3]; };
__device__ __forceinline__ void sbox_reset(SBox& b) { for (int k = 0; k < 3; k++) { b.lo[k] = URT_INF; b.hi[k] = -URT_INF; } }
__device__ __forceinline__ void sbox_grow(SBox& b, const unsigned int* q) {
  for (int k = 0; k < 3; k++) { b.lo[k] = f_min(b.lo[k], ord2f(q[1 + k])); b.hi[k] = f_max(b.hi[k], ord2f(q[4 + k])); }
}
__device__ __forceinline__ float sbox_half_area(const SBox& b) {
  float d0 = b.hi[0] - b.lo[0], d1 = b.hi[1] - b.lo[1], d2 = b.hi[2] - b.lo[2];
  return d0 * d1 + d1 * d2 + d2 * d0;
}

// one thread per node of the level: the sweep of blas_builder.cpp (right-to-left suffix areas, left-to-right costs, first strict minimum over
// axes 0, 1, 2), the sizes of the two sides, how many of them need a node of their own, and the bin words those will take
__global__ __launch_bounds__(64) void k_sah_eval(Dev D, int src, int n_nodes, int level) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  int4 rg = D.sl_rng[src][i];
  const int size = rg.y - rg.x, nb = rg.w >> 1;
  uint4 b0 = D.sl_cb0[src][i]; uint2 b1 = D.sl_cb1[src][i];
  const float clo[3] = {ord2f(b0.x), ord2f(b0.y), ord2f(b0.z)}, chi[3] = {ord2f(b0.w), ord2f(b1.x), ord2f(b1.y)};
  const unsigned int* B = D.sbins + D.sl_binoff[src][i];
  int best_axis = -1, best_bin = -1, best_nl = 0;
  float best_cost = URT_INF;
  for (int ax = 0; ax < 3; ax++) {
    if (!(chi[ax] - clo[ax] > 0.0f)) continue;
    const unsigned int* A = B + ax * nb * 7;
    float right_area[kSahBinsBig]; int right_cnt[kSahBinsBig];
    SBox acc; sbox_reset(acc); int c = 0;
    for (int b = nb - 1; b > 0; b--) { if (A[b * 7]) sbox_grow(acc, A + b * 7); c += (int)A[b * 7]; right_area[b] = sbox_half_area(acc); right_cnt[b] = c; }
    sbox_reset(acc); c = 0;
    for (int b = 0; b < nb - 1; b++) {
      if (A[b * 7]) sbox_grow(acc, A + b * 7);
      c += (int)A[b * 7];
      if (c == 0 || right_cnt[b + 1] == 0) continue;
      float cost = sbox_half_area(acc) * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
      if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; best_nl = c; }
    }
  }
  int nl = best_nl;
  if (best_axis < 0 || nl <= 0 || nl >= size || level > 56) { best_axis = -1; nl = size / 2; }      // no plane separates (or, as in blas_builder.cpp, the tree is 56 levels deep): halve by the current order — keeps the tree finite
  const int nr = size - nl;
  const int kids = (nl > D.leaf_max) + (nr > D.leaf_max);
  D.sl_dec[i] = make_int4(best_axis, best_bin, nl, kids);
  D.sl_cnt[i] = kids;
  D.sl_words[i] = (nl > D.leaf_max ? 3 * sah_nb(nl) * 7 : 0) + (nr > D.leaf_max ? 3 * sah_nb(nr) * 7 : 0);
}

// the node records of the level (final ids level_base + i) and the next level's list
__global__ __launch_bounds__(64) void k_sah_emit(Dev D, int src, int n_nodes, int level_base) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const int dst = src ^ 1;
  int4 rg = D.sl_rng[src][i], dec = D.sl_dec[i];
  const int a = rg.x, size = rg.y - rg.x, nb = rg.w >> 1, nl = dec.z, nr = size - nl;
  const int m = (int)(D.keys_s[a] >> 32);                     // (keys_s: MeshObject of the position; filled by k_sah_positions' caller)
  const float pad = __uint_as_float(D.stats[m].ext) * 1.52587890625e-5f + 1e-30f;
  SBox bl, br; sbox_reset(bl); sbox_reset(br);
  const unsigned int* B = D.sbins + D.sl_binoff[src][i];
  if (dec.x >= 0) {
    const unsigned int* A = B + dec.x * nb * 7;
    for (int b = 0; b < nb; b++) if (A[b * 7]) sbox_grow(b <= dec.y ? bl : br, A + b * 7);
  } else {
    // halved by position: the bins (if any axis had an extent) give the whole range's box, used for both sides — loose but sound
    uint4 b0 = D.sl_cb0[src][i]; uint2 b1 = D.sl_cb1[src][i];
    const float clo[3] = {ord2f(b0.x), ord2f(b0.y), ord2f(b0.z)}, chi[3] = {ord2f(b0.w), ord2f(b1.x), ord2f(b1.y)};
    bool any = false;
    for (int ax = 0; ax < 3 && !any; ax++) if (chi[ax] - clo[ax] > 0.0f) { const unsigned int* A = B + ax * nb * 7; for (int b = 0; b < nb; b++) if (A[b * 7]) sbox_grow(bl, A + b * 7); any = true; }
    if (!any) {
      // all centroids coincide, nothing was binned: the box the PARENT recorded for this child (its pad taken off again), or for a root the MeshObject's extent
      if (rg.z >= 0) {
        const float4* pn = D.nodes + 4 * (size_t)rg.z;
        float4 q0 = pn[0], q1 = pn[1], q2 = pn[2];
        if ((rg.w & 1) == 0) { bl.lo[0] = q0.x + pad; bl.lo[1] = q0.y + pad; bl.lo[2] = q0.z + pad; bl.hi[0] = q0.w - pad; bl.hi[1] = q1.x - pad; bl.hi[2] = q1.y - pad; }
        else { bl.lo[0] = q1.z + pad; bl.lo[1] = q1.w + pad; bl.lo[2] = q2.x + pad; bl.hi[0] = q2.y - pad; bl.hi[1] = q2.z - pad; bl.hi[2] = q2.w - pad; }
      } else { float e = __uint_as_float(D.stats[m].ext); for (int k = 0; k < 3; k++) { bl.lo[k] = -e; bl.hi[k] = e; } }
    }
    br = bl;
  }
  const int id = level_base + i;
  const int next_base = level_base + n_nodes;
  int kid = D.sl_cscan[i], wat = D.sl_wscan[i];
  int32_t code[2];
  const int ca[2] = {a, a + nl}, cs[2] = {nl, nr};
  for (int k = 0; k < 2; k++) {
    if (cs[k] <= D.leaf_max) { code[k] = leaf_code(ca[k], cs[k]); continue; }
    const int cnb = sah_nb(cs[k]);
    code[k] = next_base + kid;
    D.sl_rng[dst][kid] = make_int4(ca[k], ca[k] + cs[k], id, k | (cnb << 1));
    D.sl_cb0[dst][kid] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0u);      // gathered by the scatter pass
    D.sl_cb1[dst][kid] = make_uint2(0u, 0u);
    D.sl_binoff[dst][kid] = wat;
    wat += 3 * cnb * 7; kid++;
  }
  float4* nd = D.nodes + 4 * (size_t)id;
  nd[0] = make_float4(bl.lo[0] - pad, bl.lo[1] - pad, bl.lo[2] - pad, bl.hi[0] + pad);
  nd[1] = make_float4(bl.hi[1] + pad, bl.hi[2] + pad, br.lo[0] - pad, br.lo[1] - pad);
  nd[2] = make_float4(br.lo[2] - pad, br.hi[0] + pad, br.hi[1] + pad, br.hi[2] + pad);
  nd[3] = make_float4(__int_as_float(code[0]), __int_as_float(code[1]), 0.0f, 0.0f);
}

__global__ __launch_bounds__(256) void k_sah_flag(Dev D, int src) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= D.T) return;
  int nd = D.snode[src][p], f = 0;
  if (nd >= 0) {
    int4 rg = D.sl_rng[src][nd], dec = D.sl_dec[nd];
    if (dec.x < 0) f = (p - rg.x) < dec.z ? 1 : 0;
    else {
      uint4 b0 = D.sl_cb0[src][nd]; uint2 b1 = D.sl_cb1[src][nd];
      const float clo[3] = {ord2f(b0.x), ord2f(b0.y), ord2f(b0.z)}, chi[3] = {ord2f(b0.w), ord2f(b1.x), ord2f(b1.y)};
      float3 c = tri_centroid(D, D.sidx[src][p]);
      const float cc[3] = {c.x, c.y, c.z};
      f = sah_bin(cc[dec.x], clo[dec.x], chi[dec.x] - clo[dec.x], rg.w >> 1) <= dec.y ? 1 : 0;
    }
  }
  D.sflag[p] = f;
}

__global__ __launch_bounds__(256) void k_sah_scatter(Dev D, int src, int level_base, int n_nodes) {
  // the next level's centroid bounds are gathered here: per workgroup in LDS when all its positions belong to one node (two children, twelve
  // words — a million same-address atomics per level otherwise: 14 ms on C5's first level), per lane for workgroups that straddle nodes
  __shared__ unsigned int scb[12];
  const int p0 = blockIdx.x * blockDim.x, p1 = min(p0 + (int)blockDim.x, D.T) - 1;
  const int nd0 = D.snode[src][p0];
  const bool uniform = nd0 >= 0 && D.snode[src][p1] == nd0;
  if (uniform) { if (threadIdx.x < 12) scb[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xffffffffu : 0u; __syncthreads(); }
  const int p = p0 + threadIdx.x;
  const int dst = src ^ 1;
  int first_child = -1;
  if (p < D.T) {
    int nd = D.snode[src][p];
    unsigned int g = D.sidx[src][p];
    if (nd < 0) { D.sidx[dst][p] = g; D.snode[dst][p] = -1; }
    else {
      int4 rg = D.sl_rng[src][nd], dec = D.sl_dec[nd];
      const int a = rg.x, nl = dec.z, nr = (rg.y - rg.x) - nl;
      const int rank = D.sscan[p] - D.sscan[a];                   // left-side positions of this node before p
      const bool left = D.sflag[p] != 0;
      const int np = left ? a + rank : a + nl + ((p - a) - rank);
      // the side's index in the next level's list, as k_sah_emit numbers it
      first_child = D.sl_cscan[nd];
      int child = -1;
      if (left) { if (nl > D.leaf_max) child = first_child; }
      else if (nr > D.leaf_max) child = first_child + (nl > D.leaf_max ? 1 : 0);
      D.sidx[dst][np] = g;
      D.snode[dst][np] = child;
      if (child >= 0) {                                           // the next level's centroid bounds
        float3 c = tri_centroid(D, g);
        unsigned int ox = f2ord(c.x), oy = f2ord(c.y), oz = f2ord(c.z);
        unsigned int *qlo, *qhi0, *qhi1;
        if (uniform) { unsigned int* q = scb + 6 * (child - first_child); qlo = q; qhi0 = q + 3; qhi1 = q + 4; }
        else { unsigned int* q0 = (unsigned int*)&D.sl_cb0[dst][child]; qlo = q0; qhi0 = q0 + 3; qhi1 = (unsigned int*)&D.sl_cb1[dst][child]; }
        if (c.x == c.x) { atomicMin(qlo, ox); atomicMax(qhi0, ox); }
        if (c.y == c.y) { atomicMin(qlo + 1, oy); atomicMax(qhi1, oy); }
        if (c.z == c.z) { atomicMin(qlo + 2, oz); atomicMax(qhi1 + 1, oz); }
      }
    }
  }
  if (uniform) {
    __syncthreads();
    if (threadIdx.x < 12) {
      const int k = threadIdx.x / 6, w = threadIdx.x % 6;
      int4 dec = D.sl_dec[nd0];
      const int kids = dec.w, child = D.sl_cscan[nd0] + k;
      const unsigned int v = scb[threadIdx.x];
      if (k < kids) {
        unsigned int* q0 = (unsigned int*)&D.sl_cb0[dst][child]; unsigned int* q1 = (unsigned int*)&D.sl_cb1[dst][child];
        unsigned int* q = w < 4 ? q0 + w : q1 + (w - 4);
        if (w < 3) { if (v != 0xffffffffu) atomicMin(q, v); } else if (v != 0u) atomicMax(q, v);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_sah_finish(Dev D, int src) {     // final order -> vals_s (what k_emit_tris reads)
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.T) D.vals_s[p] = D.sidx[src][p];
}

__global__ __launch_bounds__(256) void k_sah_keys(Dev D) {                 // keys_s[p] = MeshObject of position p (positions never leave their MeshObject's segment)
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.T) D.keys_s[p] = (unsigned long long)(unsigned int)find_mesh(D.tri_first, D.n_meshes, p) << 32;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

namespace urtd {

#define LBVH_HIP(expr)                                                                              \
  do {                                                                                              \
    hipError_t e__ = (expr);                                                                        \
    if (e__ != hipSuccess) {                                                                        \
      err = std::string(#expr) + ": " + hipGetErrorString(e__);                                     \
      if (temp) (void)hipFree(temp);                                                                \
      for (void* p : out.allocs) (void)hipFree(p);                                                  \
      out = LbvhOutput();                                                                           \
      return e__ == hipErrorOutOfMemory ? URT_ERR_OUT_OF_MEMORY : URT_ERR_HIP;                      \
    }                                                                                               \
  } while (0)

int lbvh_build(const LbvhInput& in, hipStream_t st, LbvhOutput& out, std::string& err) {
  out = LbvhOutput();
  void* temp = nullptr;
  const int nm = in.n_meshes;
  out.h_mesh_root.assign((size_t)nm, kEmptyMeshRoot);
  std::vector<int32_t> first((size_t)nm + 1, 0);
  for (int m = 0; m < nm; m++) {
    long off = in.h_offsets[m], cnt = in.h_counts[m];
    if (off < 0 || cnt < 0 || off + cnt > in.n_indices) {
      err = "MeshObject " + std::to_string(m) + ": indices_offset/count outside _Indices";
      return URT_ERR_SCENE;
    }
    long t = (long)first[(size_t)m] + cnt / 3;
    if (t >= (1L << 28)) { err = "more than 2^28 triangles"; return URT_ERR_SCENE; }
    first[(size_t)m + 1] = (int32_t)t;
  }
  const int T = first[(size_t)nm];
  out.n_tris = T;
  // outputs
  auto alloc_out = [&](void** p, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e == hipSuccess) out.allocs.push_back(*p);
    return e;
  };
  LBVH_HIP(alloc_out((void**)&out.mesh_root, sizeof(int32_t) * (size_t)std::max(nm, 1)));
  if (T == 0 || nm == 0) {
    if (nm > 0) LBVH_HIP(hipMemcpyAsync(out.mesh_root, out.h_mesh_root.data(), sizeof(int32_t) * (size_t)nm, hipMemcpyHostToDevice, st));
    LBVH_HIP(hipStreamSynchronize(st));
    return URT_OK;
  }
  LBVH_HIP(alloc_out((void**)&out.nodes, sizeof(float4) * 4 * (size_t)T));          // upper bound: fewer than T interior nodes
  LBVH_HIP(alloc_out((void**)&out.tri_verts, sizeof(float4) * 3 * (size_t)T));
  LBVH_HIP(alloc_out((void**)&out.tri_norms, sizeof(float4) * 3 * (size_t)T));

  // one temporary slab, carved up
  size_t sort_bytes = 0, scan_bytes = 0;
  {
    unsigned long long* k = nullptr; unsigned int* v = nullptr; int* f = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, k, k, v, v, (size_t)T, 0, 64, st);
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, f, f, 0, (size_t)T, rocprim::plus<int>(), st);
  }
  size_t at = 0;
  auto carve = [&](size_t bytes) { size_t o = at; at += align256(bytes); return o; };
  const size_t o_first = carve(sizeof(int32_t) * ((size_t)nm + 1));
  const size_t o_tlo = carve(sizeof(float4) * (size_t)T), o_thi = carve(sizeof(float4) * (size_t)T);
  const size_t o_keys = carve(8 * (size_t)T), o_vals = carve(4 * (size_t)T), o_keys_s = carve(8 * (size_t)T), o_vals_s = carve(4 * (size_t)T);
  const size_t o_range = carve(8 * (size_t)T), o_child = carve(8 * (size_t)T), o_parent = carve(4 * 2 * (size_t)T), o_visits = carve(4 * (size_t)T);
  const size_t o_nlo = carve(sizeof(float4) * (size_t)T), o_nhi = carve(sizeof(float4) * (size_t)T);
  const size_t o_keep = carve(4 * (size_t)T), o_is_top = carve(4 * (size_t)T), o_top_id = carve(4 * (size_t)T), o_flag = carve(4 * (size_t)T), o_rank = carve(4 * (size_t)T);
  const size_t o_queue = carve(4 * ((size_t)nm + 2 * (size_t)kTopOrderNodes + 8));
  const size_t o_stats = carve(sizeof(MeshStat) * (size_t)kStatShards * (size_t)nm);
  const size_t o_scalars = carve(64);
  const size_t o_root_id = carve(4 * (size_t)nm);
  // builder 3 (GPU binned SAH)
  const size_t LN = in.sah ? (size_t)T / 2 + (size_t)nm + 2 : 0;          // most nodes a level can have (every node has > leaf_max >= 1 triangles)
  size_t o_sidx[2] = {0, 0}, o_snode[2] = {0, 0}, o_sflag = 0, o_sscan = 0, o_slrng[2] = {0, 0}, o_slcb0[2] = {0, 0}, o_slcb1[2] = {0, 0}, o_slbo[2] = {0, 0};
  size_t o_sldec = 0, o_slcnt = 0, o_slcscan = 0, o_slwords = 0, o_slwscan = 0, o_sbins = 0;
  // bin words a level can need: nodes of fewer than 64 triangles take 3 x 8 x 7 = 168 words and have > leaf_max triangles, bigger ones 672 per >= 64
  const size_t bin_words = in.sah ? (168 / ((size_t)std::min(std::max(in.leaf_max, 1), 8) + 1) + 12) * (size_t)T + 4096 : 0;
  if (in.sah) {
    for (int k = 0; k < 2; k++) {
      o_sidx[k] = carve(4 * (size_t)T); o_snode[k] = carve(4 * (size_t)T);
      o_slrng[k] = carve(16 * LN); o_slcb0[k] = carve(16 * LN); o_slcb1[k] = carve(8 * LN); o_slbo[k] = carve(4 * LN);
    }
    o_sflag = carve(4 * (size_t)T); o_sscan = carve(4 * (size_t)T);
    o_sldec = carve(16 * LN); o_slcnt = carve(4 * LN); o_slcscan = carve(4 * LN); o_slwords = carve(4 * LN); o_slwscan = carve(4 * LN);
    o_sbins = carve(4 * bin_words);
  }
  const size_t o_tdq0 = in.depth_budget ? carve(sizeof(int4) * (size_t)T) : 0, o_tdq1 = in.depth_budget ? carve(sizeof(int4) * (size_t)T) : 0;
  const size_t o_sort = carve(sort_bytes), o_scan = carve(scan_bytes);
  LBVH_HIP(hipMalloc(&temp, at));
  char* base = (char*)temp;

  Dev D{};
  D.mesh_objects = in.mesh_objects; D.n_meshes = nm;
  D.vertices = in.vertices; D.n_vertices = in.n_vertices;
  D.indices = in.indices; D.n_indices = in.n_indices;
  D.normals = in.normals; D.n_normals = in.n_normals;
  D.tri_first = (const int32_t*)(base + o_first); D.T = T; D.leaf_max = std::min(std::max(in.leaf_max, 1), 8);
  D.tlo = (float4*)(base + o_tlo); D.thi = (float4*)(base + o_thi);
  D.keys = (unsigned long long*)(base + o_keys); D.vals = (unsigned int*)(base + o_vals);
  D.keys_s = (unsigned long long*)(base + o_keys_s); D.vals_s = (unsigned int*)(base + o_vals_s);
  D.range = (int2*)(base + o_range); D.child = (int2*)(base + o_child); D.parent = (int*)(base + o_parent); D.visits = (unsigned int*)(base + o_visits);
  D.nlo = (float4*)(base + o_nlo); D.nhi = (float4*)(base + o_nhi);
  D.keep = (int*)(base + o_keep); D.is_top = (int*)(base + o_is_top); D.top_id = (int*)(base + o_top_id); D.flag = (int*)(base + o_flag); D.rank = (int*)(base + o_rank);
  D.queue = (int*)(base + o_queue);
  D.stats = (MeshStat*)(base + o_stats);
  D.mesh_root = out.mesh_root;
  D.scalars = (int*)(base + o_scalars);
  D.root_id = (int*)(base + o_root_id);
  if (in.sah) {
    for (int k = 0; k < 2; k++) {
      D.sidx[k] = (unsigned int*)(base + o_sidx[k]); D.snode[k] = (int*)(base + o_snode[k]);
      D.sl_rng[k] = (int4*)(base + o_slrng[k]); D.sl_cb0[k] = (uint4*)(base + o_slcb0[k]); D.sl_cb1[k] = (uint2*)(base + o_slcb1[k]); D.sl_binoff[k] = (int*)(base + o_slbo[k]);
    }
    D.sflag = (int*)(base + o_sflag); D.sscan = (int*)(base + o_sscan);
    D.sl_dec = (int4*)(base + o_sldec); D.sl_cnt = (int*)(base + o_slcnt); D.sl_cscan = (int*)(base + o_slcscan); D.sl_words = (int*)(base + o_slwords); D.sl_wscan = (int*)(base + o_slwscan);
    D.sbins = (unsigned int*)(base + o_sbins);
  }
  D.tdq[0] = (int4*)(base + o_tdq0); D.tdq[1] = (int4*)(base + o_tdq1);
  {   // depth budget of the top-down builder: what a median tree of the biggest MeshObject needs, + 6 levels of slack for lopsided radix splits
    int biggest = 1;
    for (int m = 0; m < nm; m++) biggest = std::max(biggest, first[(size_t)m + 1] - first[(size_t)m]);
    int h = 1, n = biggest, lm = std::min(std::max(in.leaf_max, 1), 8);
    while (n > lm) { n = (n + 1) >> 1; h++; }
    D.depth_cap = h + in.depth_slack;
  }
  D.nodes = out.nodes; D.tri_verts = out.tri_verts; D.tri_norms = out.tri_norms;

  LBVH_HIP(hipMemcpyAsync(base + o_first, first.data(), sizeof(int32_t) * ((size_t)nm + 1), hipMemcpyHostToDevice, st));
  const unsigned int gb = (unsigned int)((T + 255) / 256);
  const unsigned int gi = (unsigned int)std::max<size_t>(gb, ((size_t)kStatShards * (size_t)nm + 255) / 256);
  int mesh_bits = 1;
  while ((1 << mesh_bits) < nm) mesh_bits++;
  hipLaunchKernelGGL(k_init, dim3(gi), dim3(256), 0, st, D);
  hipLaunchKernelGGL(k_tri_bounds, dim3(gb), dim3(256), 0, st, D);
  hipLaunchKernelGGL(k_fold_stats, dim3((unsigned int)((nm + 63) / 64)), dim3(64), 0, st, D);
  int sah_nodes = -1, sah_levels = 0;
  if (in.sah) {
    // ---- builder 3: binned SAH, level by level ----
    auto scan = [&](int* in_, int* out_, size_t n) { return rocprim::exclusive_scan(base + o_scan, scan_bytes, in_, out_, 0, n, rocprim::plus<int>(), st); };
    auto blocks = [](size_t n, int per) { return dim3((unsigned int)((std::max<size_t>(n, 1) + (size_t)per - 1) / (size_t)per)); };
    hipLaunchKernelGGL(k_sah_keys, dim3(gb), dim3(256), 0, st, D);
    hipLaunchKernelGGL(k_sah_roots, dim3(1), dim3(64), 0, st, D);
    hipLaunchKernelGGL(k_sah_positions, dim3(gb), dim3(256), 0, st, D);
    int sc[4] = {0, 0, 0, 0};
    LBVH_HIP(hipMemcpyAsync(sc, D.scalars, sizeof sc, hipMemcpyDeviceToHost, st));
    LBVH_HIP(hipStreamSynchronize(st));
    size_t n = (size_t)sc[3], words = (size_t)sc[2];
    int src = 0, level_base = 0;
    while (n > 0) {
      if (n > LN || words > bin_words || sah_levels >= 120) { err = "GPU SAH builder: level " + std::to_string(sah_levels) + " does not fit its buffers"; if (temp) (void)hipFree(temp); for (void* q : out.allocs) (void)hipFree(q); out = LbvhOutput(); return URT_ERR_SCENE; }
      hipLaunchKernelGGL(k_sah_bins_init, blocks(words, 256), dim3(256), 0, st, D, (int)words);
      hipLaunchKernelGGL(k_sah_bin, dim3(gb), dim3(256), 0, st, D, src);
      hipLaunchKernelGGL(k_sah_eval, blocks(n, 64), dim3(64), 0, st, D, src, (int)n, sah_levels);
      LBVH_HIP(scan(D.sl_cnt, D.sl_cscan, n));
      LBVH_HIP(scan(D.sl_words, D.sl_wscan, n));
      hipLaunchKernelGGL(k_sah_emit, blocks(n, 64), dim3(64), 0, st, D, src, (int)n, level_base);
      hipLaunchKernelGGL(k_sah_flag, dim3(gb), dim3(256), 0, st, D, src);
      LBVH_HIP(scan(D.sflag, D.sscan, (size_t)T));
      hipLaunchKernelGGL(k_sah_scatter, dim3(gb), dim3(256), 0, st, D, src, level_base, (int)n);
      int t4[4] = {0, 0, 0, 0};
      LBVH_HIP(hipMemcpyAsync(&t4[0], D.sl_cscan + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
      LBVH_HIP(hipMemcpyAsync(&t4[1], D.sl_cnt + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
      LBVH_HIP(hipMemcpyAsync(&t4[2], D.sl_wscan + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
      LBVH_HIP(hipMemcpyAsync(&t4[3], D.sl_words + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
      LBVH_HIP(hipStreamSynchronize(st));
      level_base += (int)n;
      n = (size_t)(t4[0] + t4[1]); words = (size_t)(t4[2] + t4[3]);
      src ^= 1; sah_levels++;
    }
    hipLaunchKernelGGL(k_sah_finish, dim3(gb), dim3(256), 0, st, D, src);
    sah_nodes = level_base;
  } else {
  hipLaunchKernelGGL(k_morton, dim3(gb), dim3(256), 0, st, D);
  LBVH_HIP(rocprim::radix_sort_pairs(base + o_sort, sort_bytes, D.keys, D.keys_s, D.vals, D.vals_s, (size_t)T, 0, 32 + mesh_bits, st));
  if (!in.depth_budget) {
    hipLaunchKernelGGL(k_karras, dim3(gb), dim3(256), 0, st, D);
  } else {
    hipLaunchKernelGGL(k_td_roots, dim3((unsigned int)((nm + 63) / 64)), dim3(64), 0, st, D);
    int src = 0;
    for (int level = 0; level < 64; level++) {
      int n_items = 0;
      LBVH_HIP(hipMemcpyAsync(&n_items, D.scalars + 3, sizeof(int), hipMemcpyDeviceToHost, st));
      LBVH_HIP(hipStreamSynchronize(st));
      if (n_items <= 0) break;
      LBVH_HIP(hipMemsetAsync(D.scalars + 3, 0, sizeof(int), st));
      hipLaunchKernelGGL(k_td_level, dim3((unsigned int)((n_items + 255) / 256)), dim3(256), 0, st, D, src, n_items);
      src ^= 1;
    }
  }
  hipLaunchKernelGGL(k_fit, dim3(gb), dim3(256), 0, st, D);
  hipLaunchKernelGGL(k_classify, dim3(gb), dim3(256), 0, st, D);
  hipLaunchKernelGGL(k_top_bfs, dim3(1), dim3(64), 0, st, D);
  hipLaunchKernelGGL(k_flags, dim3(gb), dim3(256), 0, st, D);
  LBVH_HIP(rocprim::exclusive_scan(base + o_scan, scan_bytes, D.flag, D.rank, 0, (size_t)T, rocprim::plus<int>(), st));
  hipLaunchKernelGGL(k_emit_nodes, dim3(gb), dim3(256), 0, st, D);
  }
  hipLaunchKernelGGL(k_emit_tris, dim3(gb), dim3(256), 0, st, D);
  LBVH_HIP(hipGetLastError());
  int scalars[4] = {0, 0, 0, 0}, last_rank = 0, last_flag = 0;
  LBVH_HIP(hipMemcpyAsync(scalars, D.scalars, sizeof scalars, hipMemcpyDeviceToHost, st));
  LBVH_HIP(hipMemcpyAsync(&last_rank, D.rank + (T - 1), sizeof(int), hipMemcpyDeviceToHost, st));
  LBVH_HIP(hipMemcpyAsync(&last_flag, D.flag + (T - 1), sizeof(int), hipMemcpyDeviceToHost, st));
  LBVH_HIP(hipMemcpyAsync(out.h_mesh_root.data(), out.mesh_root, sizeof(int32_t) * (size_t)nm, hipMemcpyDeviceToHost, st));
  LBVH_HIP(hipStreamSynchronize(st));
  (void)hipFree(temp);
  temp = nullptr;
  if (scalars[0] != 0) {
    int at_slot = scalars[0] - 1;
    err = "_Indices[" + std::to_string(at_slot) + "] is outside _Vertices/_Normals";
    for (void* p : out.allocs) (void)hipFree(p);
    out = LbvhOutput();
    return URT_ERR_SCENE;
  }
  out.max_depth = scalars[1];
  out.n_nodes = scalars[2] + last_rank + last_flag;        // top of the forest + the compacted rest
  if (sah_nodes >= 0) { out.n_nodes = sah_nodes; out.max_depth = std::max(scalars[1], sah_nodes > 0 ? sah_levels + 1 : 0); }
  return URT_OK;
}

}  // namespace urtd
