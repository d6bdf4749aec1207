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
// Any conservative BVH gives the same pixels: the closest-hit rule (strict t <, ties to the lower index slot, A.4) is in
// the traversal, not in the tree.  tests/test_gpu_lbvh.py checks structure, bit-identical frames and the build time.
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
  return URT_OK;
}

}  // namespace urtd
