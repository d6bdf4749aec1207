// kernels.hip — hand-written HIP kernels for gfx950 (MI355X, CDNA4): the path-tracing hot path of
// RemyMuj/UnityRayTracer re-designed for 64-wide wavefronts.
//
// Reference being replaced: Assets/Shaders/RayTraceShader.compute ("RS", kernel CSMain RS:431-469 and
// everything it calls) and Assets/Shaders/AdditionShader.shader ("AS", AS:9,39-41).
//
// This is not a translation of the HLSL:
//  * the reference intersects EVERY triangle of a mesh per ray (RS:243); here each MeshObject has a
//    triangle BVH over pre-transformed world-space triangles (v0, e1, e2 as float4 records), traversed
//    with a per-lane stack that lives in LDS ([entry][lane] layout: conflict-free, no scratch);
//  * the reference carries a 68-byte RayHit with the material through traversal (RS:36-41); here the
//    traversal carries (t, kind, id, u, v) and normals/material are fetched once per closest hit;
//  * the reference is one thread per pixel for all bounces; the default kernel here (k_sched, kernel_mode 3) keeps a
//    fixed grid of waves resident for the whole frame: a lane whose path has ended takes a new pixel from the frame's
//    sharded work counter (wave64 ballot + prefix popcount + ONE atomic per refill), and the lanes of a wave are
//    scheduled by phase (object-level walk / triangle-BVH loop / shading) so that the long loops run for the lanes
//    that need them;
//  * the hot, small tables live in LDS next to the stacks: the breadth-first top of the triangle-BVH forest, the
//    object-level heaps, MeshObject roots and sphere centres/radii (k_sched prologue);
//  * the other kernel modes (0 one thread per pixel, 1 one launch per bounce over compacted queues, 2 persistent waves
//    without phase scheduling, 4 a path pool in LDS) share every device function with the default one and exist as
//    measured alternatives and cross-checks.
// Arithmetic is the normative float32 of include/urt_math.h, compiled with -ffp-contract=off; results
// are bit-identical to the scalar restatement in oracle/ (tests/test_gpu_parity.py).
#include "experiments.h"    // first: it looks at the -D switches before any default below is defined
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/urt_math.h"
#include "urt_device.h"
#include "kernels.h"

using namespace urt;
using namespace urtd;

namespace {

struct LocalCounters {
  unsigned int rays = 0, tlas_nodes = 0, blas_nodes = 0, tri_tests = 0, sphere_tests = 0;
  unsigned int hit_tri = 0, hit_sphere = 0, hit_ground = 0, hit_sky = 0, pixels = 0;
};

struct HitRec {
  float t;     // distance, +inf = miss
  int kid;     // id << 2 | kind;  kind: 0 none, 1 ground plane, 2 sphere, 3 triangle;  id: sphere index, or leaf-order triangle slot
  float u, v;  // barycentrics of a triangle hit
  __device__ __forceinline__ int kind() const { return kid & 3; }
  __device__ __forceinline__ int id() const { return (int)((unsigned)kid >> 2); }
  __device__ __forceinline__ void set(int kind, int id) { kid = (id << 2) | kind; }
};

__device__ __forceinline__ v3 xyz(float4 q) { return mk3(q.x, q.y, q.z); }
// wave64 vote straight from the lane predicate (HIP's wballot(int) first materialises the predicate as 0/1 in a VGPR and
// compares it again: two VALU instructions per vote, and the scheduler votes several times per trip)
__device__ __forceinline__ unsigned long long wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int as_int(float f) { return __builtin_bit_cast(int, f); }
__device__ __forceinline__ float as_float(int i) { return __builtin_bit_cast(float, i); }

// ---------------------------------------------------------------------------------------------------
// object-level BVH (the reference's implicit heap) — RS:271-291 slab test, RS:294-361 traversal
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool tlas_slab(float4 a, float4 b, v3 o, v3 rcp) {
  if (a.x == b.x && a.y == b.y && a.z == b.z) return false;      // RS:273 empty node
  float t_min = -kFLOAT_MAX, t_max = kFLOAT_MAX;
  float t1 = (a.x - o.x) * rcp.x, t2 = (b.x - o.x) * rcp.x;
  t_min = f_max(t_min, f_min(t1, t2)); t_max = f_min(t_max, f_max(t1, t2));
  t1 = (a.y - o.y) * rcp.y; t2 = (b.y - o.y) * rcp.y;
  t_min = f_max(t_min, f_min(t1, t2)); t_max = f_min(t_max, f_max(t1, t2));
  t1 = (a.z - o.z) * rcp.z; t2 = (b.z - o.z) * rcp.z;
  t_min = f_max(t_min, f_min(t1, t2)); t_max = f_min(t_max, f_max(t1, t2));
  return t_max >= t_min;
}

// the same, handing out the t_min / t_max it compared (0, 0 for an empty node): what the object-level cull looks at (urt_math.h tlas_cull)
__device__ __forceinline__ bool tlas_slab_t(float4 a, float4 b, v3 o, v3 rcp, float& t_min, float& t_max) {
  t_min = 0.0f; t_max = 0.0f;
  if (a.x == b.x && a.y == b.y && a.z == b.z) return false;      // RS:273 empty node
  t_min = -kFLOAT_MAX; t_max = kFLOAT_MAX;
  float t1 = (a.x - o.x) * rcp.x, t2 = (b.x - o.x) * rcp.x;
  t_min = f_max(t_min, f_min(t1, t2)); t_max = f_min(t_max, f_max(t1, t2));
  t1 = (a.y - o.y) * rcp.y; t2 = (b.y - o.y) * rcp.y;
  t_min = f_max(t_min, f_min(t1, t2)); t_max = f_min(t_max, f_max(t1, t2));
  t1 = (a.z - o.z) * rcp.z; t2 = (b.z - o.z) * rcp.z;
  t_min = f_max(t_min, f_min(t1, t2)); t_max = f_min(t_max, f_max(t1, t2));
  return t_max >= t_min;
}
// Object-level cull: the leaf's cull word (second float4 of the packed node, .w) is non-zero only when the library has verified that the
// leaf's box contains the object's triangles (csrc/cullflags.hip); then the object is skipped when the reference's own slab values say the
// ray passes the box, or the box lies behind the origin or beyond the ground-plane hit, by a margin (urt_math.h tlas_cull)
__device__ __forceinline__ bool leaf_culled(float4 b, float t_min, float t_max, float t_ground) {
  return as_int(b.w) != 0 && tlas_cull(t_min, t_max, t_ground);
}

// RS:175-196 without the material copy (fetched at shading time)
template <bool COUNT>
__device__ __forceinline__ void intersect_sphere(const DevScene& S, int idx, v3 o, v3 d, HitRec& best, LocalCounters& lc,
                                                 const float4* lds_pr = nullptr) {
  if (COUNT) lc.sphere_tests++;
  float4 pr;
  if (lds_pr) pr = lds_pr[idx]; else pr = S.sphere_pr[idx];
  v3 dd = o - xyz(pr);
  float p1 = -dot(d, dd);
  float p2sqr = p1 * p1 - dot(dd, dd) + pr.w * pr.w;
  if (p2sqr < 0) return;
  float p2 = f_sqrt(p2sqr);
  float t = p1 - p2 > 0 ? p1 - p2 : p1 + p2;
  if (t > 0 && t < best.t) { best.t = t; best.set(2, idx); }
}

// The triangles of one BVH leaf: Moller-Trumbore with back-face culling, RS:199-234 (edge1/edge2 pre-subtracted on
// upload), and the closer-hit rule RS:251 extended by "equal t inside one IntersectMeshObject call goes to the lower
// index slot" (A.4) — `best_i` is the index slot of a hit made in THIS call, or -1.
template <bool COUNT>
__device__ __forceinline__ void test_triangle(float4 r0, float4 r1, float4 r2, int slot_in_leaf_order, v3 o, v3 d, HitRec& best, int& best_i,
                                              LocalCounters& lc) {
  if (COUNT) lc.tri_tests++;
  v3 edge1 = xyz(r1), edge2 = xyz(r2);
  v3 pvec = cross(d, edge2);
  float det = dot(edge1, pvec);
  if (det < kEPSILON) return;
  float inv_det = 1.0f / det;
  v3 tvec = o - xyz(r0);
  float u = dot(tvec, pvec) * inv_det;
  if (u < 0.0f || u > 1.0f) return;
  v3 qvec = cross(tvec, edge1);
  float v = dot(d, qvec) * inv_det;
  if (v < 0.0f || u + v > 1.0f) return;
  float t = dot(edge2, qvec) * inv_det;
  int islot = as_int(r0.w);
  bool closer = (t > 0 && t < best.t) || (t > 0 && t == best.t && best_i >= 0 && islot < best_i);
  if (closer) { best.t = t; best.set(3, slot_in_leaf_order); best.u = u; best.v = v; best_i = islot; }
}

// `lds_first` >= 0: the leaf's records are read from `lds_tris` (an LDS copy) starting at triangle lds_first instead of
// from the global array; the leaf-order slot reported for a hit is the global one either way.
template <bool COUNT>
__device__ __forceinline__ void test_leaf(const DevScene& S, int32_t leaf, v3 o, v3 d, HitRec& best, int& best_i, LocalCounters& lc,
                                          const float4* lds_tris = nullptr, int lds_first = -1) {
  uint32_t code = ~(uint32_t)leaf;
  uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
  if (lds_first >= 0) {
    for (uint32_t k = 0; k < cnt; k += 2) {            // two records per round, both read before either is tested (a wall quad is one round)
      const bool two = k + 1 < cnt;
      const float4* ta = lds_tris + 3 * ((uint32_t)lds_first + k);
      const float4* tb = lds_tris + 3 * ((uint32_t)lds_first + k + (two ? 1u : 0u));
      float4 a0 = ta[0], a1 = ta[1], a2 = ta[2];
      float4 b0 = tb[0], b1 = tb[1], b2 = tb[2];
      test_triangle<COUNT>(a0, a1, a2, (int)(first + k), o, d, best, best_i, lc);
      if (two) test_triangle<COUNT>(b0, b1, b2, (int)(first + k + 1), o, d, best, best_i, lc);
    }
    return;
  }
  // two triangles per round: both records are requested before either is tested, so a leaf of 4 costs two memory
  // round trips on the dependent chain instead of four
  for (uint32_t k = 0; k < cnt; k += 2) {
    const float4* ta = (const float4*)((const char*)S.tri_verts + (first + k) * 48u);   // uniform base + 32-bit byte offset (< 4 GiB: checked on the host)
    bool two = k + 1 < cnt;
    const float4* tb = (const float4*)((const char*)S.tri_verts + (first + k + (two ? 1u : 0u)) * 48u);
    float4 a0 = ta[0], a1 = ta[1], a2 = ta[2];
    float4 b0 = tb[0], b1 = tb[1], b2 = tb[2];
    test_triangle<COUNT>(a0, a1, a2, (int)(first + k), o, d, best, best_i, lc);
    if (two) test_triangle<COUNT>(b0, b1, b2, (int)(first + k + 1), o, d, best, best_i, lc);
  }
}

// ---------------------------------------------------------------------------------------------------
// triangle BVH traversal for one MeshObject.  Replaces the brute-force loop RS:243-266 and returns
// the same winner: minimum t, ties inside one call going to the lowest index slot (A.4).
// Stack: LDS, entry e of this lane at stk[e * 64].  Cursor values: >= 0 interior node, < 0 leaf code,
// kBlasDone = traversal finished.
// ---------------------------------------------------------------------------------------------------
static constexpr int32_t kBlasDone = (int32_t)0x80000000;   // never a valid leaf code (it would be ~0x7fffffff)

// per-ray constants of the slab test on centre / half-extent boxes (include/urt_math.h "Slab test of the triangle BVH ..."): the
// traversal reads S.blas_cnodes, the (c, h) copy of the builders' [lo, hi] nodes —
//   q0 = c0.xyz, h0.x   q1 = h0.yz, c1.xy   q2 = c1.z, h1.xyz   q3 = child0, child1 (int bits), 0, 0
using BlasRay = CRay;
__device__ __forceinline__ BlasRay blas_ray(v3 o, v3 d) { return cray(o, d); }
// both children's [t near, t far]
__device__ __forceinline__ void cnode_slabs(float4 q0, float4 q1, float4 q2, const BlasRay& R, float tbest, float& tn0, float& tf0, float& tn1, float& tf1) {
  cslab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, R, tbest, tn0, tf0);
  cslab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, R, tbest, tn1, tf1);
}

__device__ __forceinline__ int32_t blas_pop(int* stk, int& sp) {
  if (sp == 0) return kBlasDone;
  sp--;
  return stk[sp * 64];
}

// One interior-node step: slab-test both children against [0, tbest], descend into the nearer hit child (ties: child 0),
// push the other; returns the next cursor.
__device__ __forceinline__ int32_t blas_node_eval(float4 q0, float4 q1, float4 q2, float4 q3, const BlasRay& R, float tbest, int* stk, int& sp) {
  float tn0, tf0, tn1, tf1;
  cnode_slabs(q0, q1, q2, R, tbest, tn0, tf0, tn1, tf1);
  bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
  int32_t c0 = as_int(q3.x), c1 = as_int(q3.y);
  if (h0 && h1) {
    bool swap = tn1 < tn0;
    stk[sp * 64] = swap ? c0 : c1;
    sp++;
    return swap ? c1 : c0;
  }
  if (h0) return c0;
  if (h1) return c1;
  return blas_pop(stk, sp);
}

// The same step without branches (the traversal loop of k_sched: per-wave instruction count is what bounds it, and the three-way
// branch of blas_node_eval costs a dozen scalar instructions per trip): the would-be pop value is read ahead (the LDS read
// overlaps the node fetch), the far child is written to the free slot above the stack top whether or not it is pushed (slot
// sp <= depth of the tree always exists: the stack has depth + 1 entries), and cursor / height are selected.
__device__ __forceinline__ int32_t blas_node_eval_flat(float4 q0, float4 q1, float4 q2, float4 q3, const BlasRay& R, float tbest, int* stk, int& sp) {
  int below = stk[max(sp - 1, 0) * 64];
  float tn0, tf0, tn1, tf1;
  cnode_slabs(q0, q1, q2, R, tbest, tn0, tf0, tn1, tf1);
  bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
  int32_t c0 = as_int(q3.x), c1 = as_int(q3.y);
  bool both = h0 && h1, none = !h0 && !h1;
  bool first1 = h1 && (!h0 || tn1 < tn0);          // child 1 is visited first: the only hit, or the nearer of two (ties: child 0)
  stk[sp * 64] = first1 ? c0 : c1;                  // the far child, where a push would put it
  int32_t popped = sp > 0 ? below : kBlasDone;
  int32_t nxt = none ? popped : (first1 ? c1 : c0);
  sp += both ? 1 : (none && sp > 0 ? -1 : 0);
  return nxt;
}

// The traversal loop of k_sched keeps the stack as a POINTER to its top entry (the one a pop returns) and a sentinel kBlasDone in
// entry 0 (written once per lane; a traversal starts at height 1): no address arithmetic per step (the far child goes to top[64], an
// immediate offset), no empty-stack test (popping the sentinel ends the traversal).  On this chip a compare, a select or a
// three-operand integer add each cost 1.8 fma (profiles/r03_logs/r3_valu_table_microbench.log): six of them per node step go.
__device__ __forceinline__ int32_t blas_node_select_ptr(bool h0, bool h1, float tn0, float tn1, int32_t c0, int32_t c1, int32_t below, int*& top_) {
  bool both = h0 && h1, none = !h0 && !h1;
  bool first1 = h1 && (!h0 || tn1 < tn0);          // child 1 is visited first: the only hit, or the nearer of two (ties: child 0)
  top_[64] = first1 ? c0 : c1;                      // the far child, where a push would put it
  int32_t nxt = none ? below : (first1 ? c1 : c0);
  top_ += both ? 64 : (none ? -64 : 0);
  return nxt;
}
__device__ __forceinline__ int32_t blas_node_eval_ptr(float4 q0, float4 q1, float4 q2, float4 q3, const BlasRay& R, float tbest, int*& top_) {
  int32_t below = *top_;
  float tn0, tf0, tn1, tf1;
  cnode_slabs(q0, q1, q2, R, tbest, tn0, tf0, tn1, tf1);
  return blas_node_select_ptr(tn0 <= tf0, tn1 <= tf1, tn0, tn1, as_int(q3.x), as_int(q3.y), below, top_);
}

// The same step on a 32-byte QUANTIZED node (csrc/qnodes.hip): two dwordx4 loads instead of four.  The twelve planes are 16-bit grid
// coordinates q; a plane's slab value is t = (origin + q cell - (o +- pad)) / d = fma(Q, S, B) with Q = 2^23 + q — built in ONE
// instruction per plane by putting q into the mantissa of 2^23 (0x4B000000 | q) —, S = cell / d and B = (origin - (o +- pad)) / d - 2^23 S
// per axis (QRay, derived from the ray at phase entry).  The 2^23 S terms cancel exactly but for the rounding of B: half a cell
// at worst, covered by the two cells the quantizer adds on every side.  Conservative culling only: the hits are the triangle tests'.
struct QRay { v3 S, Bp, Bm; };
__device__ __forceinline__ QRay make_qray(v3 o, v3 d, float4 forg, float4 fcell) {
  // the quantized planes are [lo, hi] planes: their per-ray constants are -(o +- pad) / d
  const float pad = f_max(f_max(f_abs(o.x), f_abs(o.y)), f_abs(o.z)) * 1.52587890625e-5f;
  const v3 idir = mk3(blas_rcp(d.x), blas_rcp(d.y), blas_rcp(d.z));
  const v3 nop = mk3(-((o.x + pad) * idir.x), -((o.y + pad) * idir.y), -((o.z + pad) * idir.z));
  const v3 nom = mk3(-((o.x - pad) * idir.x), -((o.y - pad) * idir.y), -((o.z - pad) * idir.z));
  QRay Q;
  Q.S = mk3(fcell.x * idir.x, fcell.y * idir.y, fcell.z * idir.z);
  Q.Bp = mk3(f_fma(-8388608.0f, Q.S.x, f_fma(forg.x, idir.x, nop.x)), f_fma(-8388608.0f, Q.S.y, f_fma(forg.y, idir.y, nop.y)),
             f_fma(-8388608.0f, Q.S.z, f_fma(forg.z, idir.z, nop.z)));
  Q.Bm = mk3(f_fma(-8388608.0f, Q.S.x, f_fma(forg.x, idir.x, nom.x)), f_fma(-8388608.0f, Q.S.y, f_fma(forg.y, idir.y, nom.y)),
             f_fma(-8388608.0f, Q.S.z, f_fma(forg.z, idir.z, nom.z)));
  return Q;
}
__device__ __forceinline__ float q_lo16(float w) { return as_float((int)(((unsigned int)as_int(w) & 0xffffu) | 0x4B000000u)); }
__device__ __forceinline__ float q_hi16(float w) { return as_float((int)__builtin_amdgcn_alignbit(0x4B00u, (unsigned int)as_int(w), 16u)); }
__device__ __forceinline__ int32_t qnode_eval_ptr(float4 u0, float4 u1, const QRay& Q, float tbest, int*& top_) {
  int32_t below = *top_;
  // child 0: lo (u0.x lo16, u0.x hi16, u0.y lo16) hi (u0.y hi16, u0.z lo16, u0.z hi16); child 1: the same from u0.w, u1.x, u1.y
  float a1x = f_fma(q_lo16(u0.x), Q.S.x, Q.Bp.x), a2x = f_fma(q_hi16(u0.y), Q.S.x, Q.Bm.x);
  float a1y = f_fma(q_hi16(u0.x), Q.S.y, Q.Bp.y), a2y = f_fma(q_lo16(u0.z), Q.S.y, Q.Bm.y);
  float a1z = f_fma(q_lo16(u0.y), Q.S.z, Q.Bp.z), a2z = f_fma(q_hi16(u0.z), Q.S.z, Q.Bm.z);
  float tn0 = f_max(f_max(f_min(a1x, a2x), f_min(a1y, a2y)), f_max(f_min(a1z, a2z), 0.0f));
  float tf0 = f_min(f_min(f_max(a1x, a2x), f_max(a1y, a2y)), f_min(f_max(a1z, a2z), tbest));
  float b1x = f_fma(q_lo16(u0.w), Q.S.x, Q.Bp.x), b2x = f_fma(q_hi16(u1.x), Q.S.x, Q.Bm.x);
  float b1y = f_fma(q_hi16(u0.w), Q.S.y, Q.Bp.y), b2y = f_fma(q_lo16(u1.y), Q.S.y, Q.Bm.y);
  float b1z = f_fma(q_lo16(u1.x), Q.S.z, Q.Bp.z), b2z = f_fma(q_hi16(u1.y), Q.S.z, Q.Bm.z);
  float tn1 = f_max(f_max(f_min(b1x, b2x), f_min(b1y, b2y)), f_max(f_min(b1z, b2z), 0.0f));
  float tf1 = f_min(f_min(f_max(b1x, b2x), f_max(b1y, b2y)), f_min(f_max(b1z, b2z), tbest));
  return blas_node_select_ptr(tn0 <= tf0, tn1 <= tf1, tn0, tn1, as_int(u1.z), as_int(u1.w), below, top_);
}

// One interior-node step: slab-test both children against [0, tbest], descend into the nearer hit child (ties: child 0),
// push the other; returns the next cursor.
template <bool COUNT>
__device__ __forceinline__ int32_t blas_node_step(const DevScene& S, int32_t cur, const BlasRay& R, float tbest, int* stk, int& sp,
                                                  LocalCounters& lc) {
  if (COUNT) lc.blas_nodes++;
  // uniform base + 32-bit byte offset (the node array is < 4 GiB: checked on the host), so the load needs no 64-bit address math
  const float4* n = (const float4*)((const char*)S.blas_cnodes + ((uint32_t)cur << 6));
  float4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
  return blas_node_eval(q0, q1, q2, q3, R, tbest, stk, sp);
}

// The same step on a node of the LDS-resident top of the forest (nodes [0, top_nodes), 4 x float4 each)
template <bool COUNT>
__device__ __forceinline__ int32_t blas_node_step_top(const float4* top, int32_t cur, const BlasRay& R, float tbest, int* stk, int& sp,
                                                      LocalCounters& lc) {
  if (COUNT) lc.blas_nodes++;
  const float4* n = top + 4 * cur;
  return blas_node_eval_flat(n[0], n[1], n[2], n[3], R, tbest, stk, sp);
}

// The walk of the LDS-resident top with the stack in pointer form (entry 0 of `bl` holds the sentinel, heights start at 1: k_sched and
// k_serve): `sp` is the height before and after.  Six half-rate instructions fewer per step than the index form above.
template <bool COUNT>
__device__ __forceinline__ int32_t blas_walk_top_ptr(const float4* top, int top_nodes, int32_t cur, const BlasRay& R, float tbest, int* bl, int& sp, LocalCounters& lc) {
  int* spp = bl + (sp - 1) * 64;
  do {
    if (COUNT) lc.blas_nodes++;
    const float4* n = top + 4 * cur;
    cur = blas_node_eval_ptr(n[0], n[1], n[2], n[3], R, tbest, spp);
  } while (cur >= 0 && cur < top_nodes);
  sp = ((int)(spp - bl) >> 6) + 1;
  return cur;
}

template <bool COUNT>
__device__ __forceinline__ void intersect_mesh(const DevScene& S, int32_t root, v3 o, v3 d, HitRec& best,
                                               int* stk, LocalCounters& lc) {
  if (root == kEmptyMeshRoot) return;
  BlasRay R = blas_ray(o, d);
  int best_i = -1;          // index slot of a hit made in THIS call (enables the equal-t tie rule)
  int sp = 0;
  int32_t cur = root;
  while (cur != kBlasDone) {
    if (cur >= 0) {
      cur = blas_node_step<COUNT>(S, cur, R, best.t, stk, sp, lc);
    } else {
      test_leaf<COUNT>(S, cur, o, d, best, best_i, lc);
      cur = blas_pop(stk, sp);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Trace — RS:364-383: ground plane, then the mesh object BVH, then the sphere BVH.
// tl / bl: this lane's LDS stacks for the object-level and the triangle-level traversals.
// ---------------------------------------------------------------------------------------------------
template <bool COUNT>
__device__ __forceinline__ HitRec trace(const DevScene& S, v3 o, v3 d, int* tl, int* bl, LocalCounters& lc) {
  lc.rays++;
  HitRec best; best.t = URT_INF; best.kid = 0; best.u = 0; best.v = 0;
  // IntersectGroundPlane RS:156-172
  {
    float t = -o.y / d.y;
    if (t > 0 && t < best.t) { best.t = t; best.kid = 1; }
  }
  // one reciprocal per axis for the object-level slab test (normative form of RS:282-283)
  v3 rcp = mk3(1.0f / (d.x + kEPSILON), 1.0f / (d.y + kEPSILON), 1.0f / (d.z + kEPSILON));
  // IntersectMeshBVH RS:294-326 (`tests` is never reset: once a leaf was reached, every later popped
  // node has its object intersected, A.5; object ids < 0 or out of range are skipped, not read)
  if (S.n_meshes > 0) {
    const float t_ground = best.t;                              // what the object-level cull compares with (urt_math.h tlas_cull)
    int check = 1; tl[0] = 0; bool seen = false;
    while (check > 0) {
      check--;
      int bi = tl[check * 64];
      bool hit = false, culled = false; int index = -1;
      if (bi < S.n_mesh_tlas) {
        if (COUNT) lc.tlas_nodes++;
        float4 a = S.mesh_tlas[2 * bi], b = S.mesh_tlas[2 * bi + 1];
        index = as_int(a.w);
        float t_min, t_max;
        hit = tlas_slab_t(a, b, o, rcp, t_min, t_max);
        culled = leaf_culled(b, t_min, t_max, t_ground);
      }
      if (hit) {
        if (index < 0) { tl[check * 64] = bi * 2 + 1; check++; tl[check * 64] = bi * 2 + 2; check++; }
        else seen = true;
      }
      if (seen && !culled && index >= 0 && index < S.n_meshes) intersect_mesh<COUNT>(S, S.mesh_root[index], o, d, best, bl, lc);
    }
  }
  // IntersectSphereBVH RS:329-361
  if (S.n_spheres > 0) {
    int check = 1; tl[0] = 0; bool seen = false;
    while (check > 0) {
      check--;
      int bi = tl[check * 64];
      bool hit = false; int index = -1;
      if (bi < S.n_sphere_tlas) {
        if (COUNT) lc.tlas_nodes++;
        float4 a = S.sphere_tlas[2 * bi], b = S.sphere_tlas[2 * bi + 1];
        index = as_int(a.w);
        hit = tlas_slab(a, b, o, rcp);
      }
      if (hit) {
        if (index < 0) { tl[check * 64] = bi * 2 + 1; check++; tl[check * 64] = bi * 2 + 2; check++; }
        else seen = true;
      }
      if (seen && index >= 0 && index < S.n_spheres) intersect_sphere<COUNT>(S, index, o, d, best, lc);
    }
  }
  return best;
}

// ---------------------------------------------------------------------------------------------------
// Shade — RS:386-428 (+ SampleHemisphere RS:103-111, GetTangentSpace RS:89-100, sky lookup A.11)
// ---------------------------------------------------------------------------------------------------
// inv_alpha1 = 1 / (alpha + 1) (RS:104), precomputed per material on the host (context.cpp pack_material); 0.5 for the diffuse lobe
__device__ __forceinline__ v3 sample_hemisphere(v3 normal, float inv_alpha1, float& seed, float px, float py) {
  float cosTheta = f_pow(rand_next(seed, px, py), inv_alpha1);
  float sinTheta = f_sqrt(1.0f - cosTheta * cosTheta);
  float phi = (2.0f * kPI) * rand_next(seed, px, py);
  float sp, cp; f_sincos(phi, sp, cp);
  v3 ts = mk3(cp * sinTheta, sp * sinTheta, cosTheta);
  v3 helper = mk3(1, 0, 0);
  if (f_abs(normal.x) > 0.99f) helper = mk3(0, 0, 1);
  v3 tangent = normalize(cross(normal, helper));
  v3 binormal = normalize(cross(normal, tangent));
  return mk3(f_fma(ts.z, normal.x, f_fma(ts.y, binormal.x, ts.x * tangent.x)),
             f_fma(ts.z, normal.y, f_fma(ts.y, binormal.y, ts.x * tangent.y)),
             f_fma(ts.z, normal.z, f_fma(ts.y, binormal.z, ts.x * tangent.z)));
}

// The result image is written once per pixel and not read by this kernel: stored non-temporally so that it does not push
// BVH lines out of the L2 (measured -1 %; the same hint on the sky's texel loads costs +3 % and is not used).
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_result(float4* p, float4 v) {
  f4v q = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(q, (f4v*)p);
}

__device__ __forceinline__ v3 sample_sky(const DevScene& S, float u, float v) {
  int W = S.sky_w, H = S.sky_h;
  float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
  float x0f = f_floor(x), y0f = f_floor(y);
  float fx = x - x0f, fy = y - y0f;
  int x0 = (int)x0f, y0 = (int)y0f;
  // repeat wrap.  The sky lookup's (u, v) lie in [-0.5, 0.5] x [-1, 0] (RS:424-425), so the texel index is within one period of the image:
  // one conditional add gives what the integer modulo (two dozen instructions each) gives; anything else takes the modulo
#ifndef URT_SKY_FASTWRAP
#define URT_SKY_FASTWRAP 1
#endif
  if (URT_SKY_FASTWRAP && (unsigned)x0 + (unsigned)W < 2u * (unsigned)W && (unsigned)y0 + (unsigned)H < 2u * (unsigned)H) {
    if (x0 < 0) x0 += W;
    if (y0 < 0) y0 += H;
  } else {
    x0 %= W; if (x0 < 0) x0 += W;
    y0 %= H; if (y0 < 0) y0 += H;
  }
  int x1 = x0 + 1; if (x1 == W) x1 = 0;
  int y1 = y0 + 1; if (y1 == H) y1 = 0;
  float4 c00 = S.sky[(size_t)y0 * W + x0], c10 = S.sky[(size_t)y0 * W + x1];
  float4 c01 = S.sky[(size_t)y1 * W + x0], c11 = S.sky[(size_t)y1 * W + x1];
  float ax = f_fma(fx, c10.x - c00.x, c00.x), bx = f_fma(fx, c11.x - c01.x, c01.x);
  float ay = f_fma(fx, c10.y - c00.y, c00.y), by = f_fma(fx, c11.y - c01.y, c01.y);
  float az = f_fma(fx, c10.z - c00.z, c00.z), bz = f_fma(fx, c11.z - c01.z, c01.z);
  return mk3(f_fma(fy, bx - ax, ax), f_fma(fy, by - ay, ay), f_fma(fy, bz - az, az));
}

// One bounce's shading: result += energy_before * Shade(ray, hit) (A.3); returns any(energy) (RS:457).
// The two halves of Shade (surface hit RS:388-419, sky miss RS:420-427) are separate functions: the phase-scheduled kernel
// runs them as separate phases (lanes of one wave that ended on the sky do not sit through the surface code and vice versa).
template <bool COUNT>
__device__ __forceinline__ bool shade_surface(const DevScene& S, const HitRec& h, v3& o, v3& d, v3& energy, v3& result,
                                              float& seed, float px, float py, LocalCounters& lc) {
  v3 e0 = energy;
  v3 s;
  {
    v3 pos = madd(h.t, d, o);
    v3 n;
    int mat;                                   // one material table: spheres, then mesh objects, then the ground plane
    if (h.kind() == 1) {                       // RS:164-170
      if (COUNT) lc.hit_ground++;
      n = mk3(0, 1, 0);
      mat = S.n_spheres + S.n_meshes;
    } else if (h.kind() == 2) {                // RS:192-194
      if (COUNT) lc.hit_sphere++;
      n = normalize(pos - xyz(S.sphere_pr[h.id()]));
      mat = h.id();
    } else {                                   // RS:259-264
      if (COUNT) lc.hit_tri++;
      const float4* tn = S.tri_norms + 3 * (size_t)h.id();
      v3 n0 = xyz(tn[0]), n1 = xyz(tn[1]), n2 = xyz(tn[2]);
      float w = 1.0f - h.u - h.v;
      n = normalize((n0 * w) + (n1 * h.u) + (n2 * h.v));
      mat = S.n_spheres + as_int(S.tri_verts[3 * (size_t)h.id() + 1].w);
    }
    // what RS:390-395, 401, 404-405, 411 derive from the material alone comes precomputed (context.cpp pack_material)
    const float4* m = S.materials + 4 * (size_t)mat;
    float4 m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3];
    float specChance = m0.w, bothChances = m1.w, diffChance = m2.w;
    float roulette = rand_next(seed, px, py);
    // RS:399-418.  The specular and the diffuse branch both end in SampleHemisphere: lanes of one wave take either, so the
    // branch-specific inputs (lobe axis, 1/(alpha+1)) are selected first and the long common part runs ONCE for both kinds
    // of lane.  Per lane the operations and their order are those of the two-branch form.
    bool is_spec = roulette < specChance;
    bool is_diff = !is_spec && diffChance > 0 && roulette < bothChances;
    if (is_spec || is_diff) {
      float inv_alpha1 = 0.5f;                 // diffuse: alpha = 1 (RS:410)
      v3 axis = n;
      if (is_spec) { inv_alpha1 = m3.y; axis = reflect(d, n); }
      o = madd(0.001f, n, pos);
      d = sample_hemisphere(axis, inv_alpha1, seed, px, py);
      if (is_spec) {
        float sd = f_saturate(dot(n, d) * m3.z);
        energy = energy * (xyz(m1) * sd);
      } else {
        energy = energy * xyz(m0);
      }
    } else {
      energy = mk3(0, 0, 0);
    }
    s = xyz(m2);
  }
  result = result + e0 * s;
  return any_nonzero(energy);
}

template <bool COUNT>
__device__ __forceinline__ bool shade_sky(const DevScene& S, v3 d, v3& energy, v3& result, LocalCounters& lc) {
  v3 e0 = energy;
  if (COUNT) lc.hit_sky++;
  energy = mk3(0, 0, 0);
  // RS:424-425 divide by the constant -PI: f_div_const (urt_math.h) = the IEEE quotient for every float (exhaustive test), ten
  // instructions fewer per division; the oracle keeps the divider
  float theta = f_div_const(f_acos(d.y), -kPI, 1.0f / -kPI);
  float phi = f_div_const(f_atan2(d.x, -d.z), -kPI, 1.0f / -kPI) * 0.5f;
  v3 s = sample_sky(S, phi, theta);
  result = result + e0 * s;
  return any_nonzero(energy);                  // false: the path ends here (RS:421,457)
}

template <bool COUNT>
__device__ __forceinline__ bool shade(const DevScene& S, const HitRec& h, v3& o, v3& d, v3& energy, v3& result,
                                      float& seed, float px, float py, LocalCounters& lc) {
  if (h.t < URT_INF) return shade_surface<COUNT>(S, h, o, d, energy, result, seed, px, py, lc);
  return shade_sky<COUNT>(S, d, energy, result, lc);
}

// Cold per-pixel uniforms (the two camera matrices, 128 B) are read from the kernel-argument segment AT USE through a
// laundered pointer instead of living in 32 SGPRs for the whole kernel: with them resident the register allocator spilled
// and re-loaded the hot BVH pointers inside the traversal loop (an s_load + s_waitcnt on every node step).
typedef const __attribute__((address_space(4))) float* kfloatp;
__device__ __forceinline__ kfloatp kernarg_floats(unsigned byte_offset) {
  const __attribute__((address_space(4))) char* p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  p += byte_offset;
  asm volatile("" : "+s"(p));                      // opaque to LICM: the loads below stay where they are written
  return (kfloatp)p;
}
__device__ __forceinline__ v3 mul_m4_k(kfloatp m, float x, float y, float z, float w) {   // urt::mul_m4 on a kernarg matrix
  v3 r;
  r.x = f_fma(m[12], w, f_fma(m[8], z, f_fma(m[4], y, m[0] * x)));
  r.y = f_fma(m[13], w, f_fma(m[9], z, f_fma(m[5], y, m[1] * x)));
  r.z = f_fma(m[14], w, f_fma(m[10], z, f_fma(m[6], y, m[2] * x)));
  return r;
}

// CreateCameraRay RS:142-153 with the uv of RS:448-449.  p_off = byte offset of the FrameParams argument in the kernarg segment.
template <unsigned P_OFF>
__device__ __forceinline__ void camera_ray(const FrameParams& P, int x, int y, float& seed, v3& o, v3& d) {
  float px = (float)x, py = (float)y;
  float r0 = rand_next(seed, px, py);
  float r1 = rand_next(seed, px, py);
  float u = (px + r0 + P.pixel_off_x) / (float)P.width * 2.0f - 1.0f;
  float v = (py + r1 + P.pixel_off_y) / (float)P.height * 2.0f - 1.0f;
  kfloatp c2w = kernarg_floats(P_OFF + (unsigned)__builtin_offsetof(FrameParams, c2w));
  kfloatp invp = kernarg_floats(P_OFF + (unsigned)__builtin_offsetof(FrameParams, invp));
  o = mul_m4_k(c2w, 0.0f, 0.0f, 0.0f, 1.0f);
  v3 dir = mul_m4_k(invp, u, v, 0.0f, 1.0f);
  dir = mul_m4_k(c2w, dir.x, dir.y, dir.z, 0.0f);
  d = normalize(dir);
}
// The same for a batched launch (modes 3, 5): the uniforms of the path's frame come from the launch's frame table in device
// memory, read with scalar loads (table pointer and frame index are wave-uniform).  `f` must be wave-uniform.
__device__ __forceinline__ void camera_ray_frame(const FrameUniforms* T, int f, const FrameParams& P, int x, int y, bool new_pixel, float& seed, v3& o, v3& d) {
  kfloatp q = (kfloatp)(unsigned long long)(T + __builtin_amdgcn_readfirstlane(f));
  if (new_pixel) seed = q[34];                     // RS:16: every pixel starts from the frame's _Seed; it carries over between a pixel's rays (RS:444)
  float px = (float)x, py = (float)y;
  float r0 = rand_next(seed, px, py);
  float r1 = rand_next(seed, px, py);
  float u = (px + r0 + q[32]) / (float)P.width * 2.0f - 1.0f;
  float v = (py + r1 + q[33]) / (float)P.height * 2.0f - 1.0f;
  kfloatp c2w = q, invp = q + 16;
  o = mul_m4_k(c2w, 0.0f, 0.0f, 0.0f, 1.0f);
  v3 dir = mul_m4_k(invp, u, v, 0.0f, 1.0f);
  dir = mul_m4_k(c2w, dir.x, dir.y, dir.z, 0.0f);
  d = normalize(dir);
}
static_assert(__builtin_offsetof(FrameUniforms, invp) == 64 && __builtin_offsetof(FrameUniforms, pixel_off_x) == 128 &&
              __builtin_offsetof(FrameUniforms, seed) == 136, "camera_ray_frame indexes the table as floats");

// Runs body(frame index as a wave-uniform value, lane predicate) once per distinct frame among the lanes of `pred` (almost
// always one: a wave's refill straddles two frames only at a frame boundary of the launch).
template <typename F>
__device__ __forceinline__ void for_each_frame(bool pred, int frame, F&& body) {
  unsigned long long todo = wballot(pred);
  while (todo) {
    int f = __builtin_amdgcn_readlane(frame, __builtin_ctzll(todo));
    bool mine = pred && frame == f;
    body(f, mine);
    todo &= ~wballot(mine);
  }
}

// kernels take (DevScene, FrameParams, ...) or (FrameParams, ...): by-value aggregates are laid out like C struct members
static constexpr unsigned kPOffAfterScene = (unsigned)((sizeof(DevScene) + alignof(FrameParams) - 1) / alignof(FrameParams) * alignof(FrameParams));

// tile -> pixel: one 8x8 tile per wave (the reference's [numthreads(8,8,1)] group, RS:431).
// Blocks are dealt round-robin to the 8 XCDs (b % 8 shares an XCD, each XCD has a private 4 MiB L2).
// xcd_run = G makes every XCD walk runs of G consecutive blocks (G * waves-per-block adjacent tiles):
// G = 1 is plain linear order, large G approaches one contiguous image band per XCD (best L2 locality,
// worst load balance: sky bands finish early).  Only speed depends on it, never results.
__device__ __forceinline__ bool tile_pixel(const FrameParams& P, int& x, int& y) {
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int b = blockIdx.x;
  int G = P.xcd_run;
  int sb = ((b / (8 * G)) * 8 + (b & 7)) * G + ((b >> 3) % G);
  int tile = sb * (blockDim.x >> 6) + wave;
  int ntiles = P.tiles_x * P.n_strips;
  if (tile >= ntiles) return false;
  int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
  x = tx * 8 + (lane & 7);
  y = (P.first_group_row + ty * P.row_stride) * 8 + (lane >> 3);
  return x < P.region_w && y < P.region_h;
}

__device__ __forceinline__ unsigned int wave_sum(unsigned int v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <bool COUNT>
__device__ __forceinline__ void flush_counters(const LocalCounters& lc, DevCounters* ctr) {
  // wave-reduce, then one atomic per wave and counter into one of kCounterShards slots: tens of
  // thousands of same-address atomics serialise at ~88/us on this chip, sharded ones do not.
  ctr += (blockIdx.x & (kCounterShards - 1));
  unsigned int r = wave_sum(lc.rays);
  unsigned int tn = 0, bn = 0, tt = 0, st = 0, ht = 0, hs = 0, hg = 0, hk = 0;
  if (COUNT) {
    tn = wave_sum(lc.tlas_nodes); bn = wave_sum(lc.blas_nodes); tt = wave_sum(lc.tri_tests); st = wave_sum(lc.sphere_tests);
    ht = wave_sum(lc.hit_tri); hs = wave_sum(lc.hit_sphere); hg = wave_sum(lc.hit_ground); hk = wave_sum(lc.hit_sky);
  }
  if ((threadIdx.x & 63) == 0) {
    if (r) atomicAdd(&ctr->rays, (unsigned long long)r);
    if (COUNT) {
      if (tn) atomicAdd(&ctr->tlas_nodes, (unsigned long long)tn);
      if (bn) atomicAdd(&ctr->blas_nodes, (unsigned long long)bn);
      if (tt) atomicAdd(&ctr->tri_tests, (unsigned long long)tt);
      if (st) atomicAdd(&ctr->sphere_tests, (unsigned long long)st);
      if (ht) atomicAdd(&ctr->hit_tri, (unsigned long long)ht);
      if (hs) atomicAdd(&ctr->hit_sphere, (unsigned long long)hs);
      if (hg) atomicAdd(&ctr->hit_ground, (unsigned long long)hg);
      if (hk) atomicAdd(&ctr->hit_sky, (unsigned long long)hk);
    }
  }
}

__device__ __forceinline__ void lane_stacks(const FrameParams& P, int*& tl, int*& bl) {
  extern __shared__ int lds[];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int per_wave = (P.tlas_stack + P.blas_stack) * 64;
  tl = lds + wave * per_wave + lane;
  bl = tl + P.tlas_stack * 64;
}

// ---------------------------------------------------------------------------------------------------
// mode 0: per-pixel megakernel — the whole of CSMain (RS:431-469) in one thread.
// ---------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(256) void k_mega(DevScene S, FrameParams P, float4* __restrict__ result, DevCounters* ctr) {
  int *tl, *bl;
  lane_stacks(P, tl, bl);
  LocalCounters lc;
  int x, y;
  if (tile_pixel(P, x, y)) {
    float px = (float)x, py = (float)y;
    float seed = P.seed;
    v3 avg = mk3(0, 0, 0);
    for (int i = 0; i < P.num_rays; i++) {
      v3 res = mk3(0, 0, 0);
      v3 o, d, energy = mk3(1, 1, 1);
      camera_ray<kPOffAfterScene>(P, x, y, seed, o, d);
      for (int k = 0; k < P.num_bounces; k++) {
        HitRec h = trace<COUNT>(S, o, d, tl, bl, lc);
        if (!shade<COUNT>(S, h, o, d, energy, res, seed, px, py, lc)) break;
      }
      avg = avg + res;
    }
    float n = (float)P.num_rays;
    st_result(result + (size_t)y * P.width + x, make_float4(avg.x / n, avg.y / n, avg.z / n, 1.0f));
  }
  flush_counters<COUNT>(lc, ctr);
}

// Work distribution of the persistent kernels.  The frame is a sequence of pixel slots in tile order (64 consecutive slots
// = one 8x8 tile).  One shared counter would be hit ~40,000 times per 1080p frame, and same-address atomics serialise at
// ~88/us on this chip — that alone cost 0.4 ms.  So the tiles are dealt round-robin to kWorkShards counters (tile t belongs
// to shard t % kWorkShards, each counter on its own 128-byte line); a wave draws from its home shard (its workgroup index)
// and moves on to the next shard when that one is dry.  All shards advance at a similar pace, so the frame is still swept
// roughly in natural order.
struct WorkCursor {
  unsigned int shard;       // shard this wave currently draws from
};
static_assert(kWorkShards == 64, "the dry-shard probe reads one counter per lane");

// Tiles are dealt to the shards in RUNS of G = P.xcd_run consecutive tiles (run r belongs to shard r % kWorkShards).  G = 1
// interleaves single tiles; a large G gives every shard contiguous image bands, and because workgroup b runs on XCD b % 8
// and starts on shard b % kWorkShards, each XCD's L2 then serves a few bands of the image instead of all of it.
__device__ __forceinline__ unsigned int shard_slots(unsigned int ntiles, unsigned int shard, unsigned int G, unsigned int NS) {   // slots owned by a shard
  unsigned int cycle = NS * G;
  unsigned int full = ntiles / cycle, rem = ntiles - full * cycle;
  unsigned int extra = rem > shard * G ? min(rem - shard * G, G) : 0u;
  return (full * G + extra) * 64u;
}
__device__ __forceinline__ unsigned int shard_tile(unsigned int shard, unsigned int q, unsigned int G, unsigned int NS) {   // q-th tile of a shard
  unsigned int run = q / G;
  return (run * NS + shard) * G + (q - run * G);
}

// slot -> pixel; false for slots that fall outside the dispatched region (ragged right/top edge)
__device__ __forceinline__ bool slot_pixel(const FrameParams& P, unsigned int tile, unsigned int l, int& x, int& y) {
  int ty = (int)tile / P.tiles_x, tx = (int)tile - ty * P.tiles_x;
  x = tx * 8 + (int)(l & 7u);
  y = (P.first_group_row + ty * P.row_stride) * 8 + (int)(l >> 3);
  return x < P.region_w && y < P.region_h;
}

// The wave takes popcount(want) slots with ONE atomic; each lane of `want` gets its own slot (prefix popcount).  Returns true
// and the pixel for lanes that received a valid one.  Sets `exhausted` when every shard is dry.
// Batched launches (mode 3): the work is the concatenation of the frames' tile sequences (ntiles = frames x tiles_per_frame,
// frame-major, so the launch sweeps frame 0 first); `frame` receives the frame a slot belongs to.
__device__ __forceinline__ bool wave_fetch_pixels(const FrameParams& P, unsigned long long want, bool mine, unsigned int* next,
                                                  unsigned int ntiles, WorkCursor& wc, bool& exhausted, int& x, int& y,
                                                  unsigned int tiles_per_frame = 0, int* frame = nullptr) {
  const int lane = threadIdx.x & 63;
  unsigned int n = (unsigned int)__popcll(want);
  const unsigned int G = (unsigned int)P.xcd_run, NS = (unsigned int)P.n_shards;
  unsigned int own = shard_slots(ntiles, wc.shard, G, NS);
  unsigned int base = 0;
  if (lane == 0) base = atomicAdd(next + wc.shard * 32u, n);
  base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);   // called by the whole wave: lane 0's value, and wave-uniform for the compiler (what hangs off it — shard moves, `exhausted` — stays in scalar registers)
  unsigned int shard = wc.shard;
  if (base + n >= own) {   // this shard is (now) dry: every lane looks at one counter, the wave moves to the next shard with work
    unsigned int seen = __hip_atomic_load(next + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long avail = wballot((unsigned int)lane < NS && seen < shard_slots(ntiles, (unsigned int)lane, G, NS)) & ~(1ull << shard);
    if (!avail) {
      exhausted = true;    // counters only grow, so this is final
    } else {
      unsigned long long after = shard == 63u ? 0ull : avail & ~((2ull << shard) - 1ull);
      wc.shard = (unsigned int)__builtin_ctzll(after ? after : avail);
    }
  }
  unsigned int local = base + (unsigned int)__popcll(want & ((1ull << lane) - 1ull));
  if (!mine || local >= own) return false;
  unsigned int tile = shard_tile(shard, local >> 6, G, NS);
  if (frame) {
    const unsigned int FG = (unsigned int)P.frame_group;
    if (FG <= 1u) { unsigned int f = tile / tiles_per_frame; tile -= f * tiles_per_frame; *frame = (int)f; }
    else {
      // frames interleaved in groups of FG: the global sequence is run 0 of frames 0..FG-1, run 1 of frames 0..FG-1, ... — the same
      // tiles of consecutive frames (same pixels, other jitter and seeds) are traced back to back, while their BVH subtrees are hot
      unsigned int rg = tile / G, w = tile - rg * G;
      unsigned int runs_pf = (tiles_per_frame + G - 1u) / G, group_runs = runs_pf * FG;
      unsigned int grp = rg / group_runs, r = rg - grp * group_runs;
      unsigned int f = grp * FG + r % FG;
      tile = (r / FG) * G + w;
      *frame = (int)f;
      if (tile >= tiles_per_frame || f >= (unsigned int)P.n_frames) return false;
    }
    ntiles = tiles_per_frame;
  }
  if (P.tile_order == 1) tile = ntiles - 1u - tile;            // top strip first
  return slot_pixel(P, tile, local & 63u, x, y);
}

// ---------------------------------------------------------------------------------------------------
// mode 2 (default): persistent waves with path regeneration.
// A fixed grid of waves stays resident for the whole frame.  Every lane owns one path at a time; when
// enough lanes of a wave have finished their pixel (sky hit, energy gone, bounce limit) the wave
// ballots the dead lanes, takes that many new pixels from the frame's work counter with ONE atomic
// (prefix popcount gives each dead lane its slot) and starts their camera rays — so the 64 lanes stay
// busy through all bounces without per-bounce launches or path state round-trips through HBM.
// Pixels are handed out in tile order (64 consecutive slots = one 8x8 tile), so refills stay coherent.
// Per-pixel arithmetic is exactly CSMain's (RS:431-469); only the lane a pixel runs on changes.
// ---------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(256) void k_persist(DevScene S, FrameParams P, float4* __restrict__ result, DevCounters* ctr,
                                                 unsigned int* __restrict__ next) {
  int *tl, *bl;
  lane_stacks(P, tl, bl);
  LocalCounters lc;
  const unsigned int ntiles = (unsigned int)(P.tiles_x * P.n_strips);
  WorkCursor wc; wc.shard = blockIdx.x & ((unsigned int)P.n_shards - 1u);
  bool alive = false, exhausted = false;
#ifdef URT_STAMPS
  unsigned long long t_start = wall_clock64(), t_exh = 0; unsigned int n_iter = 0, n_fetch = 0;
#endif
  int x = 0, y = 0, ray_i = 0, k = 0;
  float px = 0, py = 0, seed = 0;
  v3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), energy = mk3(0, 0, 0), res = mk3(0, 0, 0), avg = mk3(0, 0, 0);
  for (;;) {
    unsigned long long dead = wballot(!alive);
    int ndead = __popcll(dead);
    if (!exhausted && ndead >= P.refill_min) {
      bool got = wave_fetch_pixels(P, dead, !alive, next, ntiles, wc, exhausted, x, y);
#ifdef URT_STAMPS
      n_fetch++; if (exhausted && !t_exh) t_exh = wall_clock64();
#endif
      if (got) {
        alive = true;
        px = (float)x; py = (float)y;
        seed = P.seed; ray_i = 0; k = 0;
        avg = mk3(0, 0, 0); res = mk3(0, 0, 0); energy = mk3(1, 1, 1);
        camera_ray<kPOffAfterScene>(P, x, y, seed, o, d);
      }
    }
    if (wballot(alive) == 0) {
      if (exhausted) break;
      continue;                       // every fetched slot fell outside the region: fetch again
    }
#ifdef URT_STAMPS
    n_iter++;
#endif
    if (alive) {
      HitRec h = trace<COUNT>(S, o, d, tl, bl, lc);
      bool cont = shade<COUNT>(S, h, o, d, energy, res, seed, px, py, lc);
      k++;
      if (!cont || k >= P.num_bounces) {            // RS:453,457-460
        avg = avg + res;                             // RS:464
        ray_i++;
        if (ray_i < P.num_rays) {                    // RS:444: next ray of this pixel, _Seed carries over
          res = mk3(0, 0, 0); energy = mk3(1, 1, 1); k = 0;
          camera_ray<kPOffAfterScene>(P, x, y, seed, o, d);
        } else {
          float n = (float)P.num_rays;
          st_result(result + (size_t)y * P.width + x, make_float4(avg.x / n, avg.y / n, avg.z / n, 1.0f));   // RS:468
          alive = false;
        }
      }
    }
  }
#ifdef URT_STAMPS
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* st = (unsigned long long*)(next + kWorkShards * 32);
    size_t w = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
    st[w] = t_start; st[w + 1] = t_exh; st[w + 2] = wall_clock64(); st[w + 3] = ((unsigned long long)n_iter << 32) | n_fetch;
  }
#endif
  flush_counters<COUNT>(lc, ctr);
}

// Trace() (RS:364-383) cut at its triangle-BVH visits, for the phase-scheduled kernels: runs from the start of Trace
// (`fresh`) or from the return of a triangle-BVH visit up to the NEXT MeshObject whose triangle BVH must be walked
// (returns true, `cur` = its root) or to the end of Trace (returns false; `best` is final).  `check`/`seen` are the
// object-level walk's stack height and its never-reset `tests` flag (RS:296-297, A.5); the object-level stack entry e of
// this path is tl[e * stride].
// TOPF (multi-mesh scenes): a ray entering a MeshObject walks the LDS-resident top of the forest (`top`, nodes
// [0, top_nodes)) right here, far children going on its traversal stack `bl` (height *sp_out): when nothing of the mesh is
// near the ray the heap walk simply continues — no round trip through the traversal phase for a mesh that is only grazed.
// FrontLds: LDS copies of the small object-level tables (null = read the global buffer).  An object-level walk is a chain
// of dependent fetches (C2: 19 heap nodes per ray); from LDS each costs tens of cycles instead of an L1/L2 round trip.
struct FrontLds {
  const float4* mesh_tlas = nullptr;     // [2 * n_mesh_tlas]
  const int32_t* mesh_root = nullptr;    // [n_meshes]
  const float4* sphere_tlas = nullptr;   // [2 * n_sphere_tlas]
  const float4* sphere_pr = nullptr;     // [n_spheres]
  const float4* small_tris = nullptr;    // [3 * n_small] triangle records of the single-leaf MeshObjects
  const int32_t* small_first = nullptr;  // [n_meshes] first triangle of MeshObject m in small_tris, or -1
};

// SP0: the height an empty triangle-BVH stack has for the caller (1 = a sentinel sits in entry 0: k_sched)
template <bool COUNT, bool TOPF = false, bool RAYS = true, int SP0 = 0>
__device__ __forceinline__ bool trace_front(const DevScene& S, bool fresh, v3 o, v3 d, HitRec& best, int& check, bool& seen,
                                            int* tl, int stride, int32_t& cur, LocalCounters& lc, const FrontLds& L = FrontLds(),
                                            const float4* top = nullptr, int top_nodes = 0, int* bl = nullptr, int* sp_out = nullptr) {
  if (fresh) {
    if (RAYS) lc.rays++;                                    // (k_sched counts its rays per wave instead: one register less per lane)
    best.t = URT_INF; best.kid = 0; best.u = 0; best.v = 0;
    float t = -o.y / d.y;                                   // IntersectGroundPlane RS:156-172
    if (t > 0 && t < best.t) { best.t = t; best.kid = 1; }
    check = 0; seen = false;
    if (S.n_meshes > 0) { check = 1; tl[0] = 0; }
  }
  v3 rcp = mk3(1.0f / (d.x + kEPSILON), 1.0f / (d.y + kEPSILON), 1.0f / (d.z + kEPSILON));
  // the object-level cull (urt_math.h tlas_cull) compares with the ground-plane hit distance; the walk resumes after triangle-BVH visits that
  // have changed best.t, so that distance is re-derived where a leaf with a cull word is met (the same operations as RS:156-172 above — and
  // only there: kept live across the loop it cost 32 B/lane of scratch in the single-mesh instantiation)
  const bool may_cull = S.cull_any != 0;
  while (check > 0) {                                        // IntersectMeshBVH RS:294-326
    check--;
    int bi = tl[check * stride];
    bool hit = false; int index = -1;
    float t_min = 0.0f, t_max = 0.0f; int cull_word = 0;
    if (bi < S.n_mesh_tlas) {
      if (COUNT) lc.tlas_nodes++;
      float4 a, b;
      if (L.mesh_tlas) { a = L.mesh_tlas[2 * bi]; b = L.mesh_tlas[2 * bi + 1]; } else { a = S.mesh_tlas[2 * bi]; b = S.mesh_tlas[2 * bi + 1]; }
      index = as_int(a.w);
      if (may_cull) { hit = tlas_slab_t(a, b, o, rcp, t_min, t_max); cull_word = as_int(b.w); }
      else hit = tlas_slab(a, b, o, rcp);
    }
    if (hit) {
      if (index < 0) { tl[check * stride] = bi * 2 + 1; check++; tl[check * stride] = bi * 2 + 2; check++; }
      else seen = true;
    }
    bool culled = false;
    if (cull_word != 0) { float t = -o.y / d.y; culled = tlas_cull(t_min, t_max, t > 0 ? t : URT_INF); }      // (the ground-plane hit distance, re-derived: RS:156-172)
    if (seen && !culled && index >= 0 && index < S.n_meshes) {
      int32_t root;
      if (L.mesh_root) root = L.mesh_root[index]; else root = S.mesh_root[index];
      if (root < 0 && root != kBlasDone) {               // a mesh of <= 8 triangles is one leaf: test it here, no phase switch
        int bi_local = -1;
        if (L.small_tris) test_leaf<COUNT>(S, root, o, d, best, bi_local, lc, L.small_tris, L.small_first[index]);
        else test_leaf<COUNT>(S, root, o, d, best, bi_local, lc);
      } else if (root != kEmptyMeshRoot) {
        if (TOPF) {
          int sp = SP0;
          if (root < top_nodes) {
            BlasRay R = blas_ray(o, d);    // recomputed per MeshObject entered: keeping it live across the heap walk costs more (spills)
            if (SP0 == 1) root = blas_walk_top_ptr<COUNT>(top, top_nodes, root, R, best.t, bl, sp, lc);
            else do root = blas_node_step_top<COUNT>(top, root, R, best.t, bl, sp, lc); while (root >= 0 && root < top_nodes);
          }
          *sp_out = sp;
          if (root == kBlasDone) continue;                   // nothing of this mesh is near the ray: on with the heap walk
        }
        cur = root;
        return true;
      }
    }
  }
  if (S.n_spheres > 0) {                                   // IntersectSphereBVH RS:329-361
    int c2 = 1; tl[0] = 0; bool seen2 = false;
    while (c2 > 0) {
      c2--;
      int bi = tl[c2 * stride];
      bool hit = false; int index = -1;
      if (bi < S.n_sphere_tlas) {
        if (COUNT) lc.tlas_nodes++;
        float4 a, b;
        if (L.sphere_tlas) { a = L.sphere_tlas[2 * bi]; b = L.sphere_tlas[2 * bi + 1]; } else { a = S.sphere_tlas[2 * bi]; b = S.sphere_tlas[2 * bi + 1]; }
        index = as_int(a.w);
        hit = tlas_slab(a, b, o, rcp);
      }
      if (hit) {
        if (index < 0) { tl[c2 * stride] = bi * 2 + 1; c2++; tl[c2 * stride] = bi * 2 + 2; c2++; }
        else seen2 = true;
      }
      if (seen2 && index >= 0 && index < S.n_spheres) intersect_sphere<COUNT>(S, index, o, d, best, lc, L.sphere_pr);
    }
  }
  return false;
}

// FRONT for multi-mesh scenes, "listed" form (front mode 2).  In trace_front<TOPF> the expensive bodies — the inline triangle tests
// of single-leaf MeshObjects (wall quads: ~150 VALU) and the walk of the LDS-resident top of a big MeshObject's BVH (~60 + 50 per
// node) — sit INSIDE the per-lane heap-walk loop: every iteration of that loop pays for both whenever any lane of the wave
// happens to be at such a leaf.  The object-level slab test (RS:271-291) never looks at the best hit so far, so WHICH objects a ray
// tests, and in which order, is a function of the ray and the heap alone.  Here a fresh ray first walks the whole heap (cheap:
// ~30 VALU per node) and writes the object ids it has to test, in the reference's order (pop order, `tests` never reset: A.5),
// as bytes into its LDS column; then the wave works the lists off in two alternating bodies: the inline triangle tests for every
// lane whose next entry is a single-leaf MeshObject, until all lanes stand at a big one, then ONE BVH-top walk for all of
// them.  Every lane still tests its objects in list order, so ties in t resolve exactly as before.  A lane whose ray has to enter a MeshObject's BVH below the LDS top leaves for the
// BLAS phase and resumes with its next entry.  cs = entries left | next entry << 8.  Called by the whole wave (`mine` = lanes in
// FRONT / RESUME); needs the object-level mesh tables in LDS and n_meshes <= 12.
// The list: up to 12 object ids of 5 bits, six per dword, kept in two registers during the walk and then in the first two
// entries of the lane's object-level stack column — the stack is dead once the walk is over, so the list costs no LDS at all
// (LDS is what limits the size of the BVH top a workgroup can keep: a first version with a byte list of its own shrank that top
// and tripled the time spent in the BLAS phase).
__device__ __forceinline__ int list_get(const int* tl, int j) {
  unsigned int w = (unsigned int)tl[j >= 6 ? 64 : 0];
  return (int)((w >> (5 * (j >= 6 ? j - 6 : j))) & 31u);
}

// Returns per lane: 0 = Trace() is complete (shade next), 1 = the ray must enter a triangle BVH (BLAS phase next), 2 = not served in this
// trip (a fresh ray whose heap walk was put off: fresh rays walk the heap together, when at least 16 of them wait or when no
// resumed ray needs the trip — resumed rays are the majority in scenes where a ray meets several big meshes, and a walk for a
// few fresh lanes would hold all of them up).
#ifdef URT_STAMPS
#define URT_FS_DECL , unsigned long long* fs
#define URT_FS_ARG , fs_arr
#define URT_FS(stmt) stmt
#else
#define URT_FS_DECL
#define URT_FS_ARG
#define URT_FS(stmt)
#endif
template <bool COUNT, int SP0 = 0>
__device__ __forceinline__ int front_listed(const DevScene& S, const FrameParams& P, bool mine, bool fresh, v3 o, v3 d, HitRec& best, int& cs,
                                            int* tl, int32_t& cur, LocalCounters& lc, const FrontLds& L, const float4* top, int* bl, int& sp,
                                            unsigned int& wave_rays URT_FS_DECL) {
  URT_FS(unsigned long long fs_t0 = wall_clock64();)
  int remaining = cs & 0xff, cursor = cs >> 8;
  const int n_fresh = __popcll(wballot(mine && fresh)), n_resumed = __popcll(wballot(mine && !fresh));
  const bool walk_now = n_fresh >= 16 || n_resumed == 0;
  if (!walk_now) mine = mine && !fresh;
  else wave_rays += (unsigned int)n_fresh;                  // Trace() invocations (RS:454), counted per wave
  const bool put_off = !walk_now && fresh;
  if (mine && fresh) {
    best.t = URT_INF; best.kid = 0; best.u = 0; best.v = 0;
    float t = -o.y / d.y;                                   // IntersectGroundPlane RS:156-172
    if (t > 0 && t < best.t) { best.t = t; best.kid = 1; }
    v3 rcp = mk3(1.0f / (d.x + kEPSILON), 1.0f / (d.y + kEPSILON), 1.0f / (d.z + kEPSILON));
    int count = 0, check = 0;
    unsigned int l0 = 0, l1 = 0;
    bool seen = false;
    if (S.n_meshes > 0) { check = 1; tl[0] = 0; }
    const float t_ground = best.t;                           // what the object-level cull compares with (urt_math.h tlas_cull)
    while (check > 0) {                                      // IntersectMeshBVH RS:294-326, the walk alone
      check--;
      int bi = tl[check * 64];
      bool hit = false, culled = false; int index = -1;
      if (bi < S.n_mesh_tlas) {
        if (COUNT) lc.tlas_nodes++;
        float4 a = L.mesh_tlas[2 * bi], b = L.mesh_tlas[2 * bi + 1];
        index = as_int(a.w);
        float t_min, t_max;
        hit = tlas_slab_t(a, b, o, rcp, t_min, t_max);
        culled = leaf_culled(b, t_min, t_max, t_ground);
      }
      if (hit) {
        if (index < 0) { tl[check * 64] = bi * 2 + 1; check++; tl[check * 64] = bi * 2 + 2; check++; }
        else seen = true;
      }
      if (seen && !culled && index >= 0 && index < S.n_meshes && L.mesh_root[index] != kEmptyMeshRoot) {
        if (count < 6) l0 |= (unsigned int)index << (5 * count); else l1 |= (unsigned int)index << (5 * (count - 6));
        count++;
      }
    }
    tl[0] = (int)l0; tl[64] = (int)l1;                       // the walk's stack is dead: its first two entries keep the list
    remaining = count; cursor = 0;
  }
  URT_FS(if (walk_now && n_fresh > 0) { fs[0] += wall_clock64() - fs_t0; fs[3]++; fs[6] += (unsigned long long)n_fresh; })
  bool need = false;
  bool has = mine && remaining > 0;
  for (;;) {
    // (1) every lane works off the single-leaf MeshObjects (<= 8 triangles: wall quads, planes) at the head of its list:
    //     one cheap body for all of them, until every lane's next entry is a big MeshObject (or its list is done)
    int obj = 0; int32_t root = kEmptyMeshRoot;
    URT_FS(unsigned long long fs_t1 = wall_clock64();)
    for (;;) {
      if (has) { obj = list_get(tl, cursor); root = L.mesh_root[obj]; }
      bool small = has && root < 0;
      if (wballot(small) == 0) break;
      URT_FS(fs[4]++;)
      if (small) {
        int bi_local = -1;
        if (L.small_tris) test_leaf<COUNT>(S, root, o, d, best, bi_local, lc, L.small_tris, L.small_first[obj]);
        else test_leaf<COUNT>(S, root, o, d, best, bi_local, lc);
        cursor++; remaining--; has = remaining > 0;
      }
    }
    // (2) ONE walk of the LDS-resident BVH top for all the lanes that now stand at a big MeshObject: the expensive body runs with
    //     as many lanes as the wave can muster, as often as the longest list has big entries
    URT_FS(fs[1] += wall_clock64() - fs_t1; fs_t1 = wall_clock64();)
    if (wballot(has) == 0) break;
    URT_FS(fs[5]++;)
    if (has) {
      sp = SP0;
      if (root < P.top_nodes) {
        BlasRay R = blas_ray(o, d);
        if (SP0 == 1) root = blas_walk_top_ptr<COUNT>(top, P.top_nodes, root, R, best.t, bl, sp, lc);
        else do root = blas_node_step_top<COUNT>(top, root, R, best.t, bl, sp, lc); while (root >= 0 && root < P.top_nodes);
      }
      cursor++; remaining--;
      if (root == kBlasDone) has = remaining > 0;           // nothing of this mesh is near the ray
      else { cur = root; need = true; has = false; }         // on to the BLAS phase; the list continues at RESUME
    }
    URT_FS(fs[2] += wall_clock64() - fs_t1;)
  }
  cs = remaining | (cursor << 8);
  if (mine && !need && S.n_spheres > 0) {                    // IntersectSphereBVH RS:329-361
    v3 rcp = mk3(1.0f / (d.x + kEPSILON), 1.0f / (d.y + kEPSILON), 1.0f / (d.z + kEPSILON));
    int c2 = 1; tl[0] = 0; bool seen2 = false;
    while (c2 > 0) {
      c2--;
      int bi = tl[c2 * 64];
      bool hit = false; int index = -1;
      if (bi < S.n_sphere_tlas) {
        if (COUNT) lc.tlas_nodes++;
        float4 a, b;
        if (L.sphere_tlas) { a = L.sphere_tlas[2 * bi]; b = L.sphere_tlas[2 * bi + 1]; } else { a = S.sphere_tlas[2 * bi]; b = S.sphere_tlas[2 * bi + 1]; }
        index = as_int(a.w);
        hit = tlas_slab(a, b, o, rcp);
      }
      if (hit) {
        if (index < 0) { tl[c2 * 64] = bi * 2 + 1; c2++; tl[c2 * 64] = bi * 2 + 2; c2++; }
        else seen2 = true;
      }
      if (seen2 && index >= 0 && index < S.n_spheres) intersect_sphere<COUNT>(S, index, o, d, best, lc, L.sphere_pr);
    }
  }
  return put_off ? 2 : need ? 1 : 0;
}

// FRONT for multi-mesh scenes, "masked" form (front mode 3): front_listed without the divergent heap walk and without the list.
// For a mesh heap of <= 31 nodes the object-level walk (RS:294-326) is a function of one bit per node — did the ray pass the node's
// slab test (RS:271-291; it never looks at the best hit so far) — and of the heap's static shape.  The nodes are kept in POP
// order (right-first pre-order: children are pushed 2i+1 then 2i+2, so the right child is popped first): the right child of the
// node at position p sits at p + 1, the left child at p + 2^(levels below p).  A fresh ray evaluates the slab test of every node
// whose outcome can matter (wave-uniform loop, bounds broadcast from LDS, no stack, no divergence), then derives with a few
// mask operations
//     P = popped nodes: the root, and level by level the children of popped, hit, interior nodes (two shifts per level),
//     T = the MeshObjects to test: popped leaves from the first popped-AND-hit leaf on in pop order (`tests` is never reset: A.5),
// and keeps T in one register: bit order = the reference's test order.  The wave then works the masks off exactly as front_listed
// works its lists off (inline triangle tests for lanes at a single-leaf MeshObject, one BVH-top walk for lanes at a big one).
// W = the walk table in LDS (context.cpp build_walk_table).  cs = T.  Returns 0 / 1 / 2 like front_listed.
struct WalkLds {
  const int* hdr = nullptr;            // [0] n_eval, levels, interior mask, exist mask  [4] leaf_any, leaf_valid  [8..11] depth masks  [12..15] left-child shifts
  const int* pos_tab = nullptr;        // [2p] triangle-BVH root of the object at position p, [2p+1] its first triangle in small_tris or -1
  const float4* eval = nullptr;        // [2e] vmin.xyz, position bit of the parent (0: the root)  [2e+1] vmax.xyz, position bit
};
template <bool COUNT, int SP0 = 0>
__device__ __forceinline__ int front_masked(const DevScene& S, const FrameParams& P, bool mine, bool fresh, v3 o, v3 d, HitRec& best, int& cs,
                                            int* tl, int32_t& cur, LocalCounters& lc, const FrontLds& L, const WalkLds& W, const float4* top, int* bl, int& sp,
                                            unsigned int& wave_rays URT_FS_DECL) {
  URT_FS(unsigned long long fs_t0 = wall_clock64();)
  unsigned int T = (unsigned int)cs;
  const int n_fresh = __popcll(wballot(mine && fresh)), n_resumed = __popcll(wballot(mine && !fresh));
  const bool walk_now = n_fresh >= 16 || n_resumed == 0;
  if (!walk_now) mine = mine && !fresh;
  else wave_rays += (unsigned int)n_fresh;                  // Trace() invocations (RS:454), counted per wave
  const bool put_off = !walk_now && fresh;
  if (walk_now && n_fresh > 0) {                             // (wave-uniform: the loop below runs on scalar control flow)
    const bool walker = mine && fresh;
    if (walker) {
      best.t = URT_INF; best.kid = 0; best.u = 0; best.v = 0;
      float t = -o.y / d.y;                                 // IntersectGroundPlane RS:156-172
      if (t > 0 && t < best.t) { best.t = t; best.kid = 1; }
    }
    v3 rcp = mk3(1.0f / (d.x + kEPSILON), 1.0f / (d.y + kEPSILON), 1.0f / (d.z + kEPSILON));
    unsigned int H = 0, Cm = 0;                              // slab test passed; object culled (urt_math.h tlas_cull)
    const float t_ground = best.t;
    const unsigned int cull_ok = (unsigned int)__builtin_amdgcn_readfirstlane(W.hdr[6]);   // leaves whose box was verified to contain their object (csrc/cullflags.hip)
    const int n_eval = __builtin_amdgcn_readfirstlane(W.hdr[0]);
    if (walker) for (int e = 0; e < n_eval; e++) {           // the slab tests that can matter, bounds broadcast from LDS
      float4 a = W.eval[2 * e], b = W.eval[2 * e + 1];
      // a node whose parent no ray of this wave passed is popped by none of them: skipped for the whole wave (pre-order: the parent's
      // bit is final by now).  Sparse scenes (C5: 4.6 of 31 nodes popped per ray) keep the cost of the stack walk, dense ones lose nothing.
      const unsigned int pbit = (unsigned int)as_int(a.w);
      if (pbit != 0u && wballot((H & pbit) != 0u) == 0) continue;
      float t_min = -kFLOAT_MAX, t_max = kFLOAT_MAX;         // tlas_slab without the empty-node test (empty nodes are not in the table)
      float t1 = (a.x - o.x) * rcp.x, t2 = (b.x - o.x) * rcp.x;
      t_min = f_max(t_min, f_min(t1, t2)); t_max = f_min(t_max, f_max(t1, t2));
      t1 = (a.y - o.y) * rcp.y; t2 = (b.y - o.y) * rcp.y;
      t_min = f_max(t_min, f_min(t1, t2)); t_max = f_min(t_max, f_max(t1, t2));
      t1 = (a.z - o.z) * rcp.z; t2 = (b.z - o.z) * rcp.z;
      t_min = f_max(t_min, f_min(t1, t2)); t_max = f_min(t_max, f_max(t1, t2));
      H |= t_max >= t_min ? (unsigned int)as_int(b.w) : 0u;
      if (cull_ok & (unsigned int)__builtin_amdgcn_readfirstlane(as_int(b.w)))      // (wave-uniform: the entry is broadcast from LDS)
        Cm |= tlas_cull(t_min, t_max, t_ground) ? (unsigned int)as_int(b.w) : 0u;
    }
    const unsigned int imask = (unsigned int)W.hdr[2];
    const int levels = __builtin_amdgcn_readfirstlane(W.hdr[1]);
    unsigned int Pm = 1u;                                    // popped: the root ...
    for (int dpt = 0; dpt + 1 < levels && dpt < 4; dpt++) {  // ... and the children of popped, hit, interior nodes, level by level
      unsigned int X = Pm & H & imask & (unsigned int)W.hdr[8 + dpt];
      Pm |= (X << 1) | (X << W.hdr[12 + dpt]);
    }
    if (COUNT && walker) lc.tlas_nodes += (unsigned int)__popc(Pm & (unsigned int)W.hdr[3]);     // BVHNode fetches of the reference's walk (bi < count)
    unsigned int src = Pm & H & (unsigned int)W.hdr[4];      // popped and hit leaves: the first one sets `tests` (RS:315), for good
    unsigned int Tn = 0;
    if (src) Tn = Pm & (unsigned int)W.hdr[5] & ~((1u << __builtin_ctz(src)) - 1u);
    if (walker) T = Tn & ~Cm;
  }
  URT_FS(if (walk_now && n_fresh > 0) { fs[0] += wall_clock64() - fs_t0; fs[3]++; fs[6] += (unsigned long long)n_fresh; })
  bool need = false;
  bool has = mine && T != 0;
  for (;;) {
    // (1) every lane works off the single-leaf MeshObjects at the head of its mask, until every lane stands at a big one (or is done)
    int32_t root = kEmptyMeshRoot; int sfirst = -1;
    URT_FS(unsigned long long fs_t1 = wall_clock64();)
    for (;;) {
      if (has) { int p = __builtin_ctz(T); root = W.pos_tab[2 * p]; sfirst = W.pos_tab[2 * p + 1]; }
      bool small = has && root < 0;
      if (wballot(small) == 0) break;
      URT_FS(fs[4]++;)
      if (small) {
        int bi_local = -1;
        if (L.small_tris) test_leaf<COUNT>(S, root, o, d, best, bi_local, lc, L.small_tris, sfirst);
        else test_leaf<COUNT>(S, root, o, d, best, bi_local, lc);
        T &= T - 1u; has = T != 0;
      }
    }
    URT_FS(fs[1] += wall_clock64() - fs_t1; fs_t1 = wall_clock64();)
    // (2) ONE walk of the LDS-resident BVH top for all the lanes that now stand at a big MeshObject
    if (wballot(has) == 0) break;
    URT_FS(fs[5]++;)
    if (has) {
      sp = SP0;
      if (root < P.top_nodes) {
        BlasRay R = blas_ray(o, d);
        if (SP0 == 1) root = blas_walk_top_ptr<COUNT>(top, P.top_nodes, root, R, best.t, bl, sp, lc);
        else do root = blas_node_step_top<COUNT>(top, root, R, best.t, bl, sp, lc); while (root >= 0 && root < P.top_nodes);
      }
      T &= T - 1u;
      if (root == kBlasDone) has = T != 0;                   // nothing of this mesh is near the ray
      else { cur = root; need = true; has = false; }         // on to the BLAS phase; the mask continues at RESUME
    }
    URT_FS(fs[2] += wall_clock64() - fs_t1;)
  }
  cs = (int)T;
  if (mine && !need && S.n_spheres > 0) {                    // IntersectSphereBVH RS:329-361
    v3 rcp = mk3(1.0f / (d.x + kEPSILON), 1.0f / (d.y + kEPSILON), 1.0f / (d.z + kEPSILON));
    int c2 = 1; tl[0] = 0; bool seen2 = false;
    while (c2 > 0) {
      c2--;
      int bi = tl[c2 * 64];
      bool hit = false; int index = -1;
      if (bi < S.n_sphere_tlas) {
        if (COUNT) lc.tlas_nodes++;
        float4 a, b;
        if (L.sphere_tlas) { a = L.sphere_tlas[2 * bi]; b = L.sphere_tlas[2 * bi + 1]; } else { a = S.sphere_tlas[2 * bi]; b = S.sphere_tlas[2 * bi + 1]; }
        index = as_int(a.w);
        hit = tlas_slab(a, b, o, rcp);
      }
      if (hit) {
        if (index < 0) { tl[c2 * 64] = bi * 2 + 1; c2++; tl[c2 * 64] = bi * 2 + 2; c2++; }
        else seen2 = true;
      }
      if (seen2 && index >= 0 && index < S.n_spheres) intersect_sphere<COUNT>(S, index, o, d, best, lc, L.sphere_pr);
    }
  }
  return put_off ? 2 : need ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------
// mode 3: persistent waves, lanes SCHEDULED BY PHASE inside the wave.
// Measured on mode 2 (profiles/README.md): after the first bounce only a minority of a wave's lanes needs
// the triangle-BVH loop and the rest idle through it (18 % VALU lane utilisation).  Here every lane carries
// a small state machine
//     DEAD -> FRONT (ground plane + object-level heap walk) -> BLAS (triangle BVH of one MeshObject)
//          -> RESUME (rest of the heap walk, spheres) -> SHADE -> FRONT (next bounce / ray) | DEAD
// and each trip round the wave loop the 64 lanes vote (ballot) on ONE phase to run.  Cheap phases (SHADE from
// `shade_min` lanes, FRONT, refill from `refill_min` dead lanes) run until at least `blas_min` lanes are parked in
// BLAS, then the traversal loop runs with that many lanes; it hands control back when fewer than `blas_exit` lanes are
// still traversing (their stack lives in LDS and the node cursor in registers, so they resume later).
// A workgroup is 4 such waves that share nothing but read-only LDS copies made in the prologue: the breadth-first top
// of the BVH forest (a ray entering a mesh walks it on its own, at LDS latency, before it joins the wave-wide loop —
// inside FRONT when TOPF, i.e. for multi-mesh scenes, else at the start of the BLAS phase) and the small object-level
// tables.  Per-pixel arithmetic and the order of its operations are exactly those of modes 0-2 (same device
// functions) — only WHEN and WHERE a lane executes them changes, so pixels are bit-identical.
// ---------------------------------------------------------------------------------------------------
enum : int { ST_DEAD = 0, ST_FRONT = 1, ST_RESUME = 2, ST_BLAS = 3, ST_SHADE = 4, ST_SKY = 5 };
// Every persistent kernel leaves its scheduler loop after P.sched_trips trips per wave, whatever the data (a frame needs ~1e3-1e5;
// the host scales the cap with the launch: frames x rays x bounces, context.cpp).  A wave that leaves that way — or through the
// per-phase traversal cap — counts itself in DevCounters::watchdog and raises the host-visible flag: its pixels are missing.
__device__ __forceinline__ void report_watchdog(const FrameParams& P, DevCounters* shard) {
  atomicAdd(&shard->watchdog, 1ull);
  if (P.trip_flag) __hip_atomic_fetch_add(P.trip_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#ifndef URT_SCHED_OCC
#define URT_SCHED_OCC 5
#endif
// MULTI: _numRays > 1 (the running resultAverage and the ray counter are live path state only then).
// One launch traces P.n_frames consecutive frames (frame table T): a lane whose path has ended takes its next pixel from the
// NEXT frame once the current one is handed out, so only the last frame of a launch pays the drain of the long paths.
// FMODE: how FRONT treats MeshObjects — 0: a ray that must enter a triangle BVH goes to the BLAS phase at once (one mesh);
// 1: it first walks the LDS-resident top of that BVH inside FRONT (several meshes); 2: listed form of 1 (front_listed above).
// QN: the traversal loop reads the 32-byte quantized nodes (S.blas_qnodes; never together with COUNT: the counting instantiation walks
// the float nodes, like the oracle).
template <bool COUNT, int BLOCK, int FMODE, bool MULTI, bool QN = false>
__global__ __launch_bounds__(BLOCK, URT_SCHED_OCC) void k_sched(DevScene S, FrameParams P, const FrameUniforms* __restrict__ T, float4* __restrict__ result, DevCounters* ctr,
                                               unsigned int* __restrict__ next) {
  // LDS of the workgroup: [top of the triangle-BVH forest: top_nodes x 64 B, shared by its waves][stacks of wave 0][wave 1]...
  // The waves of a workgroup share nothing else and never synchronise after this copy.
  extern __shared__ int lds[];
  float4* lds4 = (float4*)lds;
  const float4* top = lds4;
  for (int i = threadIdx.x; i < P.top_nodes * 4; i += blockDim.x) lds4[i] = S.blas_cnodes[i];
  int at = P.top_nodes * 4;                                     // running offset in float4 units
  FrontLds L;
  if (P.lds_mesh) {                                             // object-level mesh heap + MeshObject roots
    for (int i = threadIdx.x; i < 2 * S.n_mesh_tlas; i += blockDim.x) lds4[at + i] = S.mesh_tlas[i];
    L.mesh_tlas = lds4 + at; at += 2 * S.n_mesh_tlas;
    for (int i = threadIdx.x; i < S.n_meshes; i += blockDim.x) ((int32_t*)(lds4 + at))[i] = S.mesh_root[i];
    L.mesh_root = (const int32_t*)(lds4 + at); at += (S.n_meshes + 3) / 4;
    if (P.lds_small) {                                          // triangle records of the single-leaf MeshObjects (quads, planes)
      for (int i = threadIdx.x; i < S.n_meshes; i += blockDim.x) ((int32_t*)(lds4 + at))[i] = S.mesh_small_first[i];
      L.small_first = (const int32_t*)(lds4 + at); at += (S.n_meshes + 3) / 4;
      for (int m = threadIdx.x; m < S.n_meshes; m += blockDim.x) {
        int sf = S.mesh_small_first[m];
        if (sf >= 0) {
          uint32_t code = ~(uint32_t)S.mesh_root[m];
          uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
          for (uint32_t q = 0; q < 3 * cnt; q++) lds4[at + 3 * sf + q] = S.tri_verts[3 * (size_t)first + q];
        }
      }
      L.small_tris = lds4 + at; at += 3 * S.n_small;
    }
  }
  WalkLds W;
  if (FMODE == 3) {                                             // masked FRONT: the walk table instead of the heap (front_masked)
    const float4* wsrc = S.mesh_tlas + 2 * S.n_mesh_tlas;
    for (int i = threadIdx.x; i < P.walk_f4; i += blockDim.x) lds4[at + i] = wsrc[i];
    W.hdr = (const int*)(lds4 + at); W.pos_tab = W.hdr + 16; W.eval = lds4 + at + 20; at += P.walk_f4;
    if (P.lds_small) {                                          // triangle records of the single-leaf MeshObjects
      for (int m = threadIdx.x; m < S.n_meshes; m += blockDim.x) {
        int sf = S.mesh_small_first[m];
        if (sf >= 0) {
          uint32_t code = ~(uint32_t)S.mesh_root[m];
          uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
          for (uint32_t q = 0; q < 3 * cnt; q++) lds4[at + 3 * sf + q] = S.tri_verts[3 * (size_t)first + q];
        }
      }
      L.small_tris = lds4 + at; at += 3 * S.n_small;
    }
  }
  if (P.lds_sphere) {                                           // object-level sphere heap + sphere positions/radii
    for (int i = threadIdx.x; i < 2 * S.n_sphere_tlas; i += blockDim.x) lds4[at + i] = S.sphere_tlas[i];
    L.sphere_tlas = lds4 + at; at += 2 * S.n_sphere_tlas;
    for (int i = threadIdx.x; i < S.n_spheres; i += blockDim.x) lds4[at + i] = S.sphere_pr[i];
    L.sphere_pr = lds4 + at; at += S.n_spheres;
  }
  __syncthreads();
  int* tl = lds + at * 4 + (threadIdx.x >> 6) * ((P.tlas_stack + P.blas_stack) * 64) + (threadIdx.x & 63);
  int* bl = tl + P.tlas_stack * 64;
  LocalCounters lc;
  const unsigned int tiles_per_frame = (unsigned int)(P.tiles_x * P.n_strips);
  const unsigned int ntiles = P.frame_group <= 1 ? tiles_per_frame * (unsigned int)P.n_frames
                                                 : (((unsigned int)P.n_frames + (unsigned int)P.frame_group - 1u) / (unsigned int)P.frame_group) * (unsigned int)P.frame_group *
                                                   ((tiles_per_frame + (unsigned int)P.xcd_run - 1u) / (unsigned int)P.xcd_run) * (unsigned int)P.xcd_run;
  // the waves of a workgroup draw from ONE shard (and, workgroups b, b + 256, ... landing on the same CU, so does the whole CU):
  // neighbours on the chip work on neighbouring tiles (a shard per wave: C2 +6 %, C3 +4 %, C4 +3 %, C5 +3 % time)
  WorkCursor wc; wc.shard = blockIdx.x & ((unsigned int)P.n_shards - 1u);
  bool exhausted = false;
  int st = ST_DEAD;
  // path state
  int xy = 0;                                    // pixel: x | y << 16 (both < 65536)
  int ray_i = 0, kf = 0;                         // kf: bounce index k | frame of the launch << 24
  float seed = 0;
  v3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), energy = mk3(0, 0, 0), res = mk3(0, 0, 0), avg = mk3(0, 0, 0);
  // trace state (one Trace() in flight per lane)
  HitRec best; best.t = URT_INF; best.kid = 0; best.u = 0; best.v = 0;
  int cs = 0;                                    // object-level heap walk (RS:294-326): stack height `check` | never-reset `tests` flag << 8
  bl[0] = kBlasDone;                                  // the sentinel below every traversal stack of this lane (blas_node_eval_ptr): heights start at 1
  int32_t cur = kBlasDone; int sp = 1, best_i = -1;   // triangle-BVH cursor of the current MeshObject
  unsigned int wave_iters = 0, wave_rays = 0;
  bool watchdog = false;
#ifdef URT_STAMPS
  unsigned long long ph_t[4] = {0, 0, 0, 0}, ph_lanes[4] = {0, 0, 0, 0}, ph_trips[4] = {0, 0, 0, 0};   // FRONT, BLAS, SHADE, blas inner trips
  unsigned long long t_begin = wall_clock64(), t_dry = 0;
  unsigned long long c_begin = __builtin_amdgcn_s_memtime();
  unsigned long long dr_trips[4] = {0, 0, 0, 0}, dr_t[3] = {0, 0, 0}, dr_live = 0, dr_lanes3 = 0;   // after the work ran dry
  unsigned long long fs_arr[7] = {0, 0, 0, 0, 0, 0, 0};   // listed FRONT: time in the heap walk / single-leaf tests / BVH-top walks, walks, single-leaf rounds, top walks, fresh lanes walked
  unsigned long long rf_t[3] = {0, 0, 0};                  // refill: time in the work-counter hand-out (the atomic's round trip), in the camera rays, refills
  unsigned long long bl_part[3] = {0, 0, 0};               // BLAS loop: lanes that took part in their trip (sum), node trips, lanes at a leaf during node trips (sum)
#endif

  for (;;) {
    if (watchdog) break;
#ifdef URT_STAMPS
    if (exhausted && !t_dry) t_dry = wall_clock64();
#endif
    unsigned long long mD = wballot(st == ST_DEAD);
    int nD = __popcll(mD);
    int nB = __popcll(wballot(st == ST_BLAS));
    int nS = __popcll(wballot(st == ST_SHADE));
    int nK = __popcll(wballot(st == ST_SKY));
    int nF = __popcll(wballot(st == ST_FRONT || st == ST_RESUME));
    // ---- refill dead lanes from the frame's work counter (one atomic per refill): when enough lanes are dead, or when
    // nothing else is left to run.  The atomic takes 2.3 us to return (7.6 % of a wave's time on C3, profiles/r03_logs/r3_stamps_refill.log),
    // but hiding it buys nothing — the SIMDs are busy with the other waves' vector work meanwhile: issuing it a trip early cost +23 % (r3_ab_split_refill.log: lanes stay dead a trip longer), reserving several
    // tiles per atomic +4 % (r3_ab_refill_chunk.log: more live state in the loop). ----
    if (!exhausted && nD > 0 && (nD >= P.refill_min || nB + nS + nK + nF == 0)) {
      int x = 0, y = 0, frame = 0;
#ifdef URT_STAMPS
      unsigned long long t_rf = wall_clock64();
#endif
      bool got = wave_fetch_pixels(P, mD, st == ST_DEAD, next, ntiles, wc, exhausted, x, y, tiles_per_frame, &frame);
#ifdef URT_STAMPS
      rf_t[0] += wall_clock64() - t_rf; t_rf = wall_clock64();
#endif
      for_each_frame(got, frame, [&](int f, bool mine) {
        if (mine) {
          st = ST_FRONT;
          ray_i = 0; kf = frame << 24; xy = x | (y << 16);
          avg = mk3(0, 0, 0); res = mk3(0, 0, 0); energy = mk3(1, 1, 1);
          camera_ray_frame(T, f, P, x, y, true, seed, o, d);
        }
      });
#ifdef URT_STAMPS
      rf_t[1] += wall_clock64() - t_rf; rf_t[2]++;
#endif
      nF = __popcll(wballot(st == ST_FRONT || st == ST_RESUME));
      nD = __popcll(wballot(st == ST_DEAD));
    }
    bool can_refill = !exhausted && nD >= P.refill_min;
    if (++wave_iters > P.sched_trips) { watchdog = true; break; }   // an exit every wave reaches, whatever the data
    int phase;
    int exit_below = 1;              // traversal runs to completion unless other lanes can make progress meanwhile
    // Surface shading is the longest straight-line code (~750 VALU whatever the lane count, the sky lookup ~270): a thin batch
    // waits while the FRONT phase can still feed it.  P.shade_split: surface hits and misses are separate phases with their
    // own thresholds (fuller lanes per trip; pays when FRONT is cheap, i.e. one mesh); otherwise they count together and
    // run back to back in one trip, so that FRONT — the expensive phase of multi-mesh scenes — gets all of them at once.
    bool sky_too = false;
    if (nB >= P.blas_min) { phase = ST_BLAS; if (nS + nK + nF > 0 || can_refill) exit_below = min(P.blas_exit, nB); }   // <= nB: the phase always advances a lane
    else if (P.shade_split) {
      if (nS >= P.shade_min) phase = ST_SHADE;
      else if (nK >= P.sky_min) phase = ST_SKY;
      else if (nF > 0) phase = ST_FRONT;
      else if (nS > 0 && nS >= nK) phase = ST_SHADE;
      else if (nK > 0) phase = ST_SKY;
      else if (nB > 0) phase = ST_BLAS;
      else if (exhausted) break;       // every lane dead and no work left
      else continue;                   // every fetched slot fell outside the region: fetch again
    }
    else if (nS + nK >= P.shade_min) { phase = nS > 0 ? ST_SHADE : ST_SKY; sky_too = true; }
    else if (nF > 0) phase = ST_FRONT;
    else if (nS + nK > 0) { phase = nS > 0 ? ST_SHADE : ST_SKY; sky_too = true; }
    else if (nB > 0) phase = ST_BLAS;
    else if (exhausted) break;       // every lane dead and no work left
    else continue;                   // every fetched slot fell outside the region: fetch again

#ifdef URT_STAMPS
    unsigned long long t_ph = wall_clock64();
    int ph_id = phase == ST_FRONT ? 0 : phase == ST_BLAS ? 1 : 2;
    ph_lanes[ph_id] += (unsigned long long)(phase == ST_FRONT ? nF : phase == ST_BLAS ? nB : phase == ST_SHADE ? nS : nK);
    ph_trips[ph_id]++;
    if (exhausted) { if (!dr_live) dr_live = 64 - nD; dr_trips[ph_id]++; }
#endif
    if (phase == ST_FRONT) {
      // ---------------- FRONT / RESUME: Trace() up to the next triangle-BVH visit (RS:364-383) ----------------
      if (FMODE < 2) wave_rays += (unsigned int)__popcll(wballot(st == ST_FRONT));     // Trace() invocations (RS:454), counted per wave
      if (FMODE >= 2) {
        bool mine = st == ST_FRONT || st == ST_RESUME;
        int r = FMODE == 3 ? front_masked<COUNT, 1>(S, P, mine, st == ST_FRONT, o, d, best, cs, tl, cur, lc, L, W, top, bl, sp, wave_rays URT_FS_ARG)
                           : front_listed<COUNT, 1>(S, P, mine, st == ST_FRONT, o, d, best, cs, tl, cur, lc, L, top, bl, sp, wave_rays URT_FS_ARG);
        if (mine && r != 2) {
          if (r == 1) { best_i = -1; st = ST_BLAS; }
          else st = best.t < URT_INF ? ST_SHADE : ST_SKY;
        }
      } else if (st == ST_FRONT || st == ST_RESUME) {
        sp = 1;
        int check = cs & 0xff; bool seen = (cs >> 8) != 0;
        bool need = FMODE == 1 ? trace_front<COUNT, true, false, 1>(S, st == ST_FRONT, o, d, best, check, seen, tl, 64, cur, lc, L, top, P.top_nodes, bl, &sp)
                               : trace_front<COUNT, false, false, 1>(S, st == ST_FRONT, o, d, best, check, seen, tl, 64, cur, lc, L);
        cs = check | (seen ? 256 : 0);
        if (need) { best_i = -1; st = ST_BLAS; }
        else st = best.t < URT_INF ? ST_SHADE : ST_SKY;
      }
    } else if (phase == ST_BLAS) {
      // ---------------- BLAS: triangle BVH of one MeshObject, resumable ----------------
      bool mine = st == ST_BLAS;
      BlasRay R = blas_ray(o, d);
      // A ray that has just entered a MeshObject first walks the LDS-resident top of the forest (nodes [0, top_nodes)) on its
      // own, at LDS latency: the first ~6 of its ~12 node visits then never wait for another lane's cache miss.  Same
      // visits in the same order as the wave-wide loop below would make; far children go on the lane's stack as usual.
      int* spp = bl + (sp - 1) * 64;                         // the top entry of the lane's stack: what a pop returns (height 1 = the sentinel)
      if (mine && cur >= 0 && cur < P.top_nodes) {           // (pointer-form stack here too: no address arithmetic per step)
        do {
          if (COUNT) lc.blas_nodes++;
          const float4* n = top + 4 * cur;
          cur = blas_node_eval_ptr(n[0], n[1], n[2], n[3], R, best.t, spp);
        } while (cur >= 0 && cur < P.top_nodes);
      }
      // The loop works on ONE integer per lane: c = the cursor of the lanes that take part, kBlasDone for every other lane.  Both
      // votes are then single compares whose result IS the ballot (kBlasDone is negative, so c >= 0 <=> an interior node of a
      // participating lane), and the loop control sits in scalar registers (the limits are pinned there).
      int32_t c = mine ? cur : kBlasDone;
      int budget = __builtin_amdgcn_readfirstlane((int)min(P.watchdog_steps, 0x7fffffffu));   // trips left before the watchdog ends this phase (counted down: no kernel argument in the loop)
      const int exit_s = __builtin_amdgcn_readfirstlane(exit_below);
      QRay Q;
      if (QN) Q = make_qray(o, d, S.blas_qnodes[0], S.blas_qnodes[1]);     // the grid frame: two wave-uniform loads per phase entry
      for (;;) {
        int nA = __popcll(wballot(c != kBlasDone));
        if (nA < exit_s) break;
        if (--budget < 0) { watchdog = true; break; }
#ifdef URT_STAMPS
        ph_trips[3]++; ph_lanes[3] += (unsigned long long)nA;
        if (exhausted) { dr_trips[3]++; dr_lanes3 += (unsigned long long)nA; }
#endif
        // majority vote: this trip runs EITHER the interior-node step OR the leaf step, for the lanes that hold that kind
        // of cursor (the others wait one trip) — so a trip costs one of the two bodies, not their sum.
        int nI = __popcll(wballot(c >= 0));
        bool node_trip = 2 * nI >= nA;             // (weighting the vote 1/3 or 2/3: +0.4 % / +2 %; a minority step: +4 ... +7 % — profiles/r03_logs/r3_ab_vote_xload.log, r3_ab_minority.log)
#ifdef URT_STAMPS
        bl_part[0] += (unsigned long long)(node_trip ? nI : nA - nI); bl_part[1] += node_trip ? 1 : 0; bl_part[2] += (unsigned long long)(node_trip ? nA - nI : 0);
#endif
        if (node_trip) {
          if (QN) {
            if (c >= 0) {
              const float4* n = (const float4*)((const char*)(S.blas_qnodes + 2) + ((uint32_t)c << 5));
              float4 u0 = n[0], u1 = n[1];
              c = qnode_eval_ptr(u0, u1, Q, best.t, spp);
            }
          } else if (c >= 0) {
            if (COUNT) lc.blas_nodes++;
            const float4* n = (const float4*)((const char*)S.blas_cnodes + ((uint32_t)c << 6));
            float4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
            c = blas_node_eval_ptr(q0, q1, q2, q3, R, best.t, spp);
          }
        } else if (c < 0 && c != kBlasDone) {
          test_leaf<COUNT>(S, c, o, d, best, best_i, lc);
          c = *spp; spp -= 64;                               // pop (the sentinel ends the traversal)
        }
      }
      if (mine) { cur = c; sp = ((int)(spp - bl) >> 6) + 1; }
      // back to the heap walk (RS:323-325 continues) — or, when nothing of Trace() is left to do (empty object-level stack and
      // no spheres), straight to shading: saves the path one scheduling round trip per bounce
      if (mine && cur == kBlasDone) st = ((FMODE == 3 ? cs : (cs & 0xff)) == 0 && S.n_spheres == 0) ? (best.t < URT_INF ? ST_SHADE : ST_SKY) : ST_RESUME;
    } else {
      // ---------------- SHADE + bookkeeping of CSMain's loops (RS:444-468) ----------------
      bool next_ray = false;
      bool cont = false, shaded = false;
      if (phase == ST_SHADE) {                              // surface hits
        if (st == ST_SHADE) { shaded = true; cont = shade_surface<COUNT>(S, best, o, d, energy, res, seed, (float)(xy & 0xffff), (float)((unsigned)xy >> 16), lc); }
      }
      if (phase == ST_SKY || (sky_too && nK > 0)) {         // misses
        if (st == ST_SKY) { shaded = true; cont = shade_sky<COUNT>(S, d, energy, res, lc); }
      }
      if (shaded) {
        kf++;
        st = ST_FRONT;
        if (!cont || (kf & 0xffffff) >= P.num_bounces) {    // RS:453,457-460
          v3 sum = (MULTI ? avg : mk3(0, 0, 0)) + res;       // RS:464
          if (MULTI) { avg = sum; ray_i++; next_ray = ray_i < P.num_rays; }
          if (!next_ray) {
            float n = (float)P.num_rays;                      // (!MULTI: n = 1 and x / 1 = x — no divisions)
            st_result(result + (size_t)((unsigned)kf >> 24) * P.frame_stride + (size_t)((unsigned)xy >> 16) * P.width + (xy & 0xffff),
                      MULTI ? make_float4(sum.x / n, sum.y / n, sum.z / n, 1.0f) : make_float4(sum.x, sum.y, sum.z, 1.0f));   // RS:468
            st = ST_DEAD;
          }
        }
      }
      if (MULTI) {                                          // RS:444: next ray of the pixel, _Seed carries over
        for_each_frame(next_ray, (int)((unsigned)kf >> 24), [&](int f, bool mine) {
          if (mine) {
            res = mk3(0, 0, 0); energy = mk3(1, 1, 1); kf &= (int)0xff000000;
            camera_ray_frame(T, f, P, xy & 0xffff, (int)((unsigned)xy >> 16), false, seed, o, d);
          }
        });
      }
    }
#ifdef URT_STAMPS
    ph_t[ph_id] += wall_clock64() - t_ph;
    if (exhausted) dr_t[ph_id] += wall_clock64() - t_ph;
#endif
  }
#ifdef URT_STAMPS
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* sp_ = (unsigned long long*)(next + kWorkShards * 32);
    size_t w = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32;
    for (int q = 0; q < 4; q++) sp_[w + 16 + q] = dr_trips[q];
    for (int q = 0; q < 3; q++) sp_[w + 20 + q] = dr_t[q];
    sp_[w + 23] = dr_live; sp_[w + 24] = dr_lanes3;
    for (int q = 0; q < 4; q++) { sp_[w + q] = ph_t[q]; sp_[w + 4 + q] = ph_lanes[q]; sp_[w + 8 + q] = ph_trips[q]; }
    sp_[w + 12] = t_begin; sp_[w + 13] = wall_clock64(); sp_[w + 14] = t_dry; sp_[w + 15] = __builtin_amdgcn_s_memtime() - c_begin;
    for (int q = 0; q < 7; q++) sp_[w + 25 + q] = fs_arr[q];
    if (FMODE < 2) { sp_[w + 25] = rf_t[0]; sp_[w + 26] = rf_t[1]; sp_[w + 27] = rf_t[2]; sp_[w + 28] = 0; sp_[w + 29] = bl_part[0]; sp_[w + 30] = bl_part[1]; sp_[w + 31] = bl_part[2]; }     // (single-mesh instantiations: no FRONT split, the refill's instead)
  }
#endif
  if (watchdog && (threadIdx.x & 63) == 0) report_watchdog(P, ctr);
  lc.rays = (threadIdx.x & 63) == 0 ? wave_rays : 0u;
  flush_counters<COUNT>(lc, ctr);
}

// ---------------------------------------------------------------------------------------------------
// mode 5: mode 3 with the triangle-BVH phase turned into a SERVICE shared by the waves of a workgroup.
// Measured on mode 3 (profiles/README.md, round 2): the kernel is VALU-issue-bound, two thirds of its vector instructions are
// the triangle-BVH loop, and that loop runs with 24 of 64 lanes on average — a wave owns 64 paths, only those that stand at a
// mesh can take part, and their number falls while the loop runs.  A VALU instruction costs the same 4 cycles whatever the
// number of active lanes, so the idle lanes are the cost.
// Here a path that must enter a triangle BVH does not traverse it on its own lane.  It POSTS the ray: origin/direction to its
// slot of a per-workgroup mailbox in global memory (L2-resident, written once), the traversal state — closest hit so far, node
// cursor, stack height — to its slot of a small LDS table, and waits (ST_WAIT).  Any wave of the workgroup that enters the
// traversal phase CLAIMS waiting rays (of its own paths or of its neighbours') onto its idle lanes — compare-and-swap on the
// slot's flag word — and keeps claiming while it runs, so the loop stays full for as long as the workgroup has rays waiting:
// 256 paths feed it instead of 64.  The traversal stack stays where it was: entry e of slot s lives in the LDS column of the
// path's own lane, whoever walks the ray uses that column.  A finished traversal writes (t, hit, u, v) back to the slot and
// flags it DONE; the owner picks it up at its next scheduling trip and carries on (RESUME / SHADE / SKY) exactly as in mode 3.
// A wave that leaves the phase with traversals in flight (other work is waiting) SUSPENDS them: cursor, stack height and the
// closest hit go back to the slot, which is flagged REQ again — any wave resumes it later.  Foreign state therefore never
// lives in registers outside the phase.
// `avail` counts the posted-and-unreserved rays of the workgroup (a semaphore: a wave reserves before it scans, returns what it
// could not claim), so waves do not all rush for the same few rays.
// Per-ray arithmetic and operation order are those of modes 0-4 (same device functions): pixels are bit-identical.
// ---------------------------------------------------------------------------------------------------
enum : int { ST_WAIT = 6 };
enum : int { MB_IDLE = 0, MB_REQ = 1, MB_BUSY = 2, MB_DONE = 3 };

__device__ __forceinline__ int lanes_below(unsigned long long m) {      // set bits of m below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}
__device__ __forceinline__ int lds_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

#ifndef URT_SERVE_OCC
#define URT_SERVE_OCC 4
#endif
#ifndef URT_SERVE_SLEEP
#define URT_SERVE_SLEEP 8
#endif
// A wave whose live paths are all being walked by its neighbours has nothing to run: it sleeps until one of its rays is
// answered or a ray is posted that it could walk itself — a short poll loop (one LDS word per lane + the counter), bounded,
// instead of full scheduling trips.
__device__ __forceinline__ void serve_wait(bool waiting, const int* my_flag, const int* avail) {
  for (int spin = 0; spin < 64; spin++) {
    __builtin_amdgcn_s_sleep(URT_SERVE_SLEEP);
    bool done = waiting && (lds_load(my_flag) & 3) == MB_DONE;
    if (wballot(done) != 0 || __builtin_amdgcn_readfirstlane(lds_load(avail)) > 0) break;
  }
}
template <bool COUNT, int BLOCK, int FMODE, bool MULTI>
__global__ __launch_bounds__(BLOCK, URT_SERVE_OCC) void k_serve(DevScene S, FrameParams P, const FrameUniforms* __restrict__ T, float4* __restrict__ result, DevCounters* ctr,
                                               unsigned int* __restrict__ next, float4* __restrict__ mail) {
  constexpr int NW = BLOCK / 64;
  static_assert(NW >= 1 && (NW & (NW - 1)) == 0, "waves per workgroup: a power of two");
  // LDS of the workgroup: [top of the BVH forest][object-level tables][mailbox: hit x BLOCK float4, best_i, cursor, flag, candidate
  // list x BLOCK ints, avail][stacks of wave 0][wave 1]...
  extern __shared__ int lds[];
  float4* lds4 = (float4*)lds;
  const float4* top = lds4;
  for (int i = threadIdx.x; i < P.top_nodes * 4; i += blockDim.x) lds4[i] = S.blas_cnodes[i];
  int at = P.top_nodes * 4;                                     // running offset in float4 units
  FrontLds L;
  if (P.lds_mesh) {
    for (int i = threadIdx.x; i < 2 * S.n_mesh_tlas; i += blockDim.x) lds4[at + i] = S.mesh_tlas[i];
    L.mesh_tlas = lds4 + at; at += 2 * S.n_mesh_tlas;
    for (int i = threadIdx.x; i < S.n_meshes; i += blockDim.x) ((int32_t*)(lds4 + at))[i] = S.mesh_root[i];
    L.mesh_root = (const int32_t*)(lds4 + at); at += (S.n_meshes + 3) / 4;
    if (P.lds_small) {
      for (int i = threadIdx.x; i < S.n_meshes; i += blockDim.x) ((int32_t*)(lds4 + at))[i] = S.mesh_small_first[i];
      L.small_first = (const int32_t*)(lds4 + at); at += (S.n_meshes + 3) / 4;
      for (int m = threadIdx.x; m < S.n_meshes; m += blockDim.x) {
        int sf = S.mesh_small_first[m];
        if (sf >= 0) {
          uint32_t code = ~(uint32_t)S.mesh_root[m];
          uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
          for (uint32_t q = 0; q < 3 * cnt; q++) lds4[at + 3 * sf + q] = S.tri_verts[3 * (size_t)first + q];
        }
      }
      L.small_tris = lds4 + at; at += 3 * S.n_small;
    }
  }
  if (P.lds_sphere) {
    for (int i = threadIdx.x; i < 2 * S.n_sphere_tlas; i += blockDim.x) lds4[at + i] = S.sphere_tlas[i];
    L.sphere_tlas = lds4 + at; at += 2 * S.n_sphere_tlas;
    for (int i = threadIdx.x; i < S.n_spheres; i += blockDim.x) lds4[at + i] = S.sphere_pr[i];
    L.sphere_pr = lds4 + at; at += S.n_spheres;
  }
  float4* m_hit = lds4 + at; at += BLOCK;                       // t, kind|id (int bits; 0 = no hit made in this call yet), u, v
  int* m_besti = lds + at * 4;                                  // index slot of that hit (the equal-t tie rule), -1 = none
  int* m_cur = m_besti + BLOCK;                                 // node cursor
  int* m_flag = m_cur + BLOCK;                                  // MB_* | stack height << 8
  int* m_cand = m_flag + BLOCK;                                 // per wave: 64 candidate slots of a refill
  int* m_avail = m_cand + BLOCK;                                // posted rays nobody has reserved yet
  at += BLOCK + 1;
  m_flag[threadIdx.x] = MB_IDLE;
  if (threadIdx.x == 0) *m_avail = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per_wave = (P.tlas_stack + P.blas_stack) * 64;
  int* const stacks = lds + at * 4;
  int* tl = stacks + wave * per_wave + lane;
  int* bl = tl + P.tlas_stack * 64;
  const int myslot = (int)threadIdx.x;
  float4* const wgmail = mail + (size_t)blockIdx.x * (size_t)(2 * BLOCK);
  LocalCounters lc;
  const unsigned int tiles_per_frame = (unsigned int)(P.tiles_x * P.n_strips);
  const unsigned int ntiles = P.frame_group <= 1 ? tiles_per_frame * (unsigned int)P.n_frames
                                                 : (((unsigned int)P.n_frames + (unsigned int)P.frame_group - 1u) / (unsigned int)P.frame_group) * (unsigned int)P.frame_group *
                                                   ((tiles_per_frame + (unsigned int)P.xcd_run - 1u) / (unsigned int)P.xcd_run) * (unsigned int)P.xcd_run;
  WorkCursor wc; wc.shard = blockIdx.x & ((unsigned int)P.n_shards - 1u);
  bool exhausted = false;
  int st = ST_DEAD;
  // path state
  int xy = 0;                                    // pixel: x | y << 16
  int ray_i = 0, kf = 0;                         // kf: bounce index k | frame of the launch << 24
  float seed = 0;
  v3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), energy = mk3(0, 0, 0), res = mk3(0, 0, 0), avg = mk3(0, 0, 0);
  HitRec best; best.t = URT_INF; best.kid = 0; best.u = 0; best.v = 0;
  int cs = 0;                                    // object-level heap walk: stack height | `tests` flag << 8 (listed FRONT: entries left | next << 8)
  unsigned int wave_iters = 0, wave_rays = 0;
  bool watchdog = false;
  unsigned long long sv[6] = {0, 0, 0, 0, 0, 0};   // COUNT: service visits, trips, lane-trips, claim rounds, rays claimed, rays suspended (per wave)
#ifdef URT_STAMPS
  unsigned long long fs_arr[7] = {0, 0, 0, 0, 0, 0, 0};
#endif

  for (;;) {
    if (watchdog) break;
    // ---- answers to the rays this wave's paths have posted ----
    if (st == ST_WAIT) {
      int f = lds_load(m_flag + myslot);
      if ((f & 3) == MB_DONE) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float4 h = m_hit[myslot];
        if (as_int(h.y) != 0) { best.t = h.x; best.kid = as_int(h.y); best.u = h.z; best.v = h.w; }   // a hit made in that call is closer (RS:251)
        st = ((cs & 0xff) == 0 && S.n_spheres == 0) ? (best.t < URT_INF ? ST_SHADE : ST_SKY) : ST_RESUME;
      }
    }
    unsigned long long mD = wballot(st == ST_DEAD);
    int nD = __popcll(mD);
    int nW = __popcll(wballot(st == ST_WAIT));
    int nS = __popcll(wballot(st == ST_SHADE));
    int nK = __popcll(wballot(st == ST_SKY));
    int nF = __popcll(wballot(st == ST_FRONT || st == ST_RESUME));
    int av = __builtin_amdgcn_readfirstlane(lds_load(m_avail));
    if (!exhausted && nD > 0 && (nD >= P.refill_min || (nS + nK + nF == 0 && (nW == 0 || av <= 0)))) {
      int x = 0, y = 0, frame = 0;
      bool got = wave_fetch_pixels(P, mD, st == ST_DEAD, next, ntiles, wc, exhausted, x, y, tiles_per_frame, &frame);
      for_each_frame(got, frame, [&](int f, bool mine) {
        if (mine) {
          st = ST_FRONT;
          ray_i = 0; kf = frame << 24; xy = x | (y << 16);
          avg = mk3(0, 0, 0); res = mk3(0, 0, 0); energy = mk3(1, 1, 1);
          camera_ray_frame(T, f, P, x, y, true, seed, o, d);
        }
      });
      nF = __popcll(wballot(st == ST_FRONT || st == ST_RESUME));
      nD = __popcll(wballot(st == ST_DEAD));
    }
    bool can_refill = !exhausted && nD >= P.refill_min;
    if (++wave_iters > P.sched_trips) { watchdog = true; break; }   // an exit every wave reaches, whatever the data
    int phase;
    bool sky_too = false;
    if (av >= P.blas_min) phase = ST_BLAS;
    else if (P.shade_split) {
      if (nS >= P.shade_min) phase = ST_SHADE;
      else if (nK >= P.sky_min) phase = ST_SKY;
      else if (nF > 0) phase = ST_FRONT;
      else if (nS > 0 && nS >= nK) phase = ST_SHADE;
      else if (nK > 0) phase = ST_SKY;
      else if (av > 0) phase = ST_BLAS;
      else if (nW > 0) { serve_wait(st == ST_WAIT, m_flag + myslot, m_avail); continue; }   // every live path of the wave is being walked by a neighbour
      else if (exhausted) break;
      else continue;
    }
    else if (nS + nK >= P.shade_min) { phase = nS > 0 ? ST_SHADE : ST_SKY; sky_too = true; }
    else if (nF > 0) phase = ST_FRONT;
    else if (nS + nK > 0) { phase = nS > 0 ? ST_SHADE : ST_SKY; sky_too = true; }
    else if (av > 0) phase = ST_BLAS;
    else if (nW > 0) { serve_wait(st == ST_WAIT, m_flag + myslot, m_avail); continue; }
    else if (exhausted) break;
    else continue;

    if (phase == ST_FRONT) {
      // ---------------- FRONT / RESUME: Trace() up to the next triangle-BVH visit (RS:364-383), as in mode 3 ----------------
      bool need = false;
      int32_t cur = kBlasDone; int sp = 0;
      if (FMODE != 2) wave_rays += (unsigned int)__popcll(wballot(st == ST_FRONT));
      if (FMODE == 2) {
        bool mine = st == ST_FRONT || st == ST_RESUME;
        int r = front_listed<COUNT>(S, P, mine, st == ST_FRONT, o, d, best, cs, tl, cur, lc, L, top, bl, sp, wave_rays URT_FS_ARG);
        if (mine && r != 2) {
          if (r == 1) need = true;
          else st = best.t < URT_INF ? ST_SHADE : ST_SKY;
        }
      } else if (st == ST_FRONT || st == ST_RESUME) {
        int check = cs & 0xff; bool seen = (cs >> 8) != 0;
        need = FMODE == 1 ? trace_front<COUNT, true, false>(S, st == ST_FRONT, o, d, best, check, seen, tl, 64, cur, lc, L, top, P.top_nodes, bl, &sp)
                          : trace_front<COUNT, false, false>(S, st == ST_FRONT, o, d, best, check, seen, tl, 64, cur, lc, L);
        cs = check | (seen ? 256 : 0);
        if (!need) st = best.t < URT_INF ? ST_SHADE : ST_SKY;
      }
      // post the rays that must enter a triangle BVH
      if (need) {
        wgmail[2 * myslot] = make_float4(o.x, o.y, o.z, 0.0f);
        wgmail[2 * myslot + 1] = make_float4(d.x, d.y, d.z, 0.0f);
        m_hit[myslot] = make_float4(best.t, 0.0f, 0.0f, 0.0f);
        m_besti[myslot] = -1;
        m_cur[myslot] = cur;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        lds_store(m_flag + myslot, MB_REQ | (sp << 8));
        st = ST_WAIT;
      }
      int n_post = __popcll(wballot(need));
      if (n_post > 0 && lane == 0) __hip_atomic_fetch_add(m_avail, n_post, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (phase == ST_BLAS) {
      // ---------------- the traversal service ----------------
      const bool others = nS + nK + nF > 0 || can_refill;     // own work waits: yield once the loop runs thin
      bool factive = false;
      v3 fo = mk3(0, 0, 0), fd = mk3(0, 0, 1);
      BlasRay R; R.idir = mk3(0, 0, 0); R.b = mk3(0, 0, 0); R.pa = mk3(0, 0, 0);
      HitRec fb; fb.t = URT_INF; fb.kid = 0; fb.u = 0; fb.v = 0;
      int fbest_i = -1, fsp = 0, fhome = 0;
      int32_t fcur = kBlasDone;
      int* fstk = bl;
      int budget = (int)min(P.watchdog_steps, 0x7fffffffu);
      bool stepped = false;
      if (COUNT) sv[0]++;
      for (;;) {
        unsigned long long mA = wballot(factive);
        int nA = __popcll(mA);
        // ---- claim waiting rays onto the idle lanes ----
        if (64 - nA >= P.pool_inloop) {
          int a2 = __builtin_amdgcn_readfirstlane(lds_load(m_avail));
          if (a2 > 0) {
            int want = 64 - nA, g = 0;
            if (lane == 0) {
              int old = __hip_atomic_fetch_add(m_avail, -want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              g = max(0, min(old, want));
              if (g < want) __hip_atomic_fetch_add(m_avail, want - g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            g = __builtin_amdgcn_readfirstlane(g);
            if (g > 0) {
              int total = 0;
#pragma unroll
              for (int j = 0; j < NW; j++) {                    // the wave's own paths first, then its neighbours'
                int s = (((wave + j) & (NW - 1)) << 6) | lane;
                int f = lds_load(m_flag + s);
                bool pend = (f & 3) == MB_REQ;
                unsigned long long m = wballot(pend);
                int r = total + lanes_below(m);
                if (pend && r < 64) m_cand[wave * 64 + r] = s | ((f >> 8) << 16);
                total += __popcll(m);
              }
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the list was written by other lanes of this wave
              int n_take = min(total, g);
              int r = lanes_below(~mA);
              bool ok = false; int c = 0;
              if (!factive && r < n_take) {
                c = m_cand[wave * 64 + r];
                int expect = MB_REQ | ((c >> 16) << 8);
                ok = __hip_atomic_compare_exchange_strong(m_flag + (c & 0xffff), &expect, MB_BUSY, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
              int nc = __popcll(wballot(ok));
              if (COUNT) { sv[3]++; sv[4] += (unsigned long long)nc; }
              if (nc < g && lane == 0) __hip_atomic_fetch_add(m_avail, g - nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (ok) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                fhome = c & 0xffff; fsp = c >> 16;
                float4 qo = wgmail[2 * fhome], qd = wgmail[2 * fhome + 1];
                float4 h = m_hit[fhome];
                fo = xyz(qo); fd = xyz(qd);
                fb.t = h.x; fb.kid = as_int(h.y); fb.u = h.z; fb.v = h.w;
                fbest_i = m_besti[fhome]; fcur = m_cur[fhome];
                fstk = stacks + (fhome >> 6) * per_wave + P.tlas_stack * 64 + (fhome & 63);
                R = blas_ray(fo, fd);
                factive = true;
              }
              mA = wballot(factive);
              nA = __popcll(mA);
            }
          }
        }
        // ---- yield? (never before the rays of this visit have advanced one trip: a visit always makes progress) ----
        if (nA == 0) break;
        if (stepped && nA < P.blas_exit) {
          bool mine_done = st == ST_WAIT && (lds_load(m_flag + myslot) & 3) == MB_DONE;
          if (others || wballot(mine_done) != 0) break;
        }
        if (--budget < 0) { watchdog = true; break; }
        stepped = true;
        if (COUNT) { sv[1]++; sv[2] += (unsigned long long)nA; }
        // ---- one trip: EITHER the interior-node step OR the leaf step (majority vote, as in mode 3) ----
        bool interior = factive && fcur >= 0;
        int nI = __popcll(wballot(interior));
        if (nI >= nA - nI) {
          if (interior) {
            if (COUNT) lc.blas_nodes++;
            float4 q0, q1, q2, q3;
            if (FMODE == 0 && fcur < P.top_nodes) { const float4* n = top + 4 * fcur; q0 = n[0]; q1 = n[1]; q2 = n[2]; q3 = n[3]; }
            else { const float4* n = (const float4*)((const char*)S.blas_cnodes + ((uint32_t)fcur << 6)); q0 = n[0]; q1 = n[1]; q2 = n[2]; q3 = n[3]; }
            fcur = blas_node_eval(q0, q1, q2, q3, R, fb.t, fstk, fsp);
          }
        } else if (factive && !interior) {
          test_leaf<COUNT>(S, fcur, fo, fd, fb, fbest_i, lc);
          fcur = blas_pop(fstk, fsp);
        }
        // ---- finished traversals: answer and free the lane ----
        if (factive && fcur == kBlasDone) {
          m_hit[fhome] = make_float4(fb.t, as_float(fb.kid), fb.u, fb.v);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          lds_store(m_flag + fhome, MB_DONE);
          factive = false;
        }
      }
      // ---- suspend what is still in flight: any wave resumes it ----
      if (factive) {
        m_hit[fhome] = make_float4(fb.t, as_float(fb.kid), fb.u, fb.v);
        m_besti[fhome] = fbest_i;
        m_cur[fhome] = fcur;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        lds_store(m_flag + fhome, MB_REQ | (fsp << 8));
      }
      int n_back = __popcll(wballot(factive));
      if (COUNT) sv[5] += (unsigned long long)n_back;
      if (n_back > 0 && lane == 0) __hip_atomic_fetch_add(m_avail, n_back, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      // ---------------- SHADE + bookkeeping of CSMain's loops (RS:444-468), as in mode 3 ----------------
      bool next_ray = false;
      bool cont = false, shaded = false;
      if (phase == ST_SHADE) {
        if (st == ST_SHADE) { shaded = true; cont = shade_surface<COUNT>(S, best, o, d, energy, res, seed, (float)(xy & 0xffff), (float)((unsigned)xy >> 16), lc); }
      }
      if (phase == ST_SKY || (sky_too && nK > 0)) {
        if (st == ST_SKY) { shaded = true; cont = shade_sky<COUNT>(S, d, energy, res, lc); }
      }
      if (shaded) {
        kf++;
        st = ST_FRONT;
        if (!cont || (kf & 0xffffff) >= P.num_bounces) {    // RS:453,457-460
          v3 sum = (MULTI ? avg : mk3(0, 0, 0)) + res;       // RS:464
          if (MULTI) { avg = sum; ray_i++; next_ray = ray_i < P.num_rays; }
          if (!next_ray) {
            float n = (float)P.num_rays;                      // (!MULTI: n = 1 and x / 1 = x — no divisions)
            st_result(result + (size_t)((unsigned)kf >> 24) * P.frame_stride + (size_t)((unsigned)xy >> 16) * P.width + (xy & 0xffff),
                      MULTI ? make_float4(sum.x / n, sum.y / n, sum.z / n, 1.0f) : make_float4(sum.x, sum.y, sum.z, 1.0f));   // RS:468
            st = ST_DEAD;
          }
        }
      }
      if (MULTI) {                                          // RS:444: next ray of the pixel, _Seed carries over
        for_each_frame(next_ray, (int)((unsigned)kf >> 24), [&](int f, bool mine) {
          if (mine) {
            res = mk3(0, 0, 0); energy = mk3(1, 1, 1); kf &= (int)0xff000000;
            camera_ray_frame(T, f, P, xy & 0xffff, (int)((unsigned)xy >> 16), false, seed, o, d);
          }
        });
      }
    }
  }
  if (watchdog && (threadIdx.x & 63) == 0) report_watchdog(P, ctr);
  if (COUNT && (threadIdx.x & 63) == 0) {
    DevCounters* c = ctr + (blockIdx.x & (kCounterShards - 1));
    for (int q = 0; q < 6; q++) if (sv[q]) atomicAdd(&c->serve[q], sv[q]);
  }
  lc.rays = (threadIdx.x & 63) == 0 ? wave_rays : 0u;
  flush_counters<COUNT>(lc, ctr);
}

// ---------------------------------------------------------------------------------------------------
// mode 4: persistent waves over a POOL of paths (K x 64 path slots per wave, state in LDS).
// Measured on mode 3 (profiles/README.md): a wave that owns exactly 64 paths runs its triangle-BVH phase with 16-20 active
// lanes and its SHADE phase with ~30 — the paths of one wave are simply spread over the phases.  Every VALU instruction
// costs 4 cycles whatever the number of active lanes, and the kernel is ~45 % VALU-issue bound, so idle lanes are the cost.
// Here a wave owns NP = 64*K paths whose state (24 words, SoA [field][slot]) lives in LDS.  Each trip the wave takes a census
// of the slot states, elects ONE phase, compacts up to 64 slots that are in that phase onto its lanes (ballot + prefix
// popcount), loads what that phase needs, runs it, and stores the state back:
//     FREE -> FRONT -> BLAS -> RESUME -> ... -> SHADE -> FRONT | FREE
// The triangle-BVH phase keeps its 64 lanes fed from the list of waiting BLAS slots while it runs (a lane whose ray has
// finished retires it and takes the next one), and yields when few lanes are left; a suspended traversal stays PINNED to
// its lane, because its stack is the lane's ([entry][lane] in LDS), and resumes there.
// Per-pixel arithmetic and operation order are those of modes 0-3 (same device functions): pixels are bit-identical.
// ---------------------------------------------------------------------------------------------------
enum : int { PS_FREE = 0, PS_FRONT = 1, PS_RESUME = 2, PS_BLAS = 3, PS_PINNED = 4, PS_SHADE = 5 };
enum : int { F_PIX = 0, F_K, F_RAYI, F_SEED, F_OX, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_EX, F_EY, F_EZ, F_RX, F_RY, F_RZ,
             F_T, F_KINDID, F_U, F_V, F_CHECK, F_CUR, F_SP, F_BESTI, F_COUNT1,      // _numRays == 1: 24 words per path
             F_AX = F_COUNT1, F_AY, F_AZ, F_COUNTN };                                // + resultAverage when _numRays > 1

// Slots whose state is in [lo, hi], in slot order: list[] receives all of them (`total`), lane L gets the L-th or -1.
template <int K>
__device__ __forceinline__ int pool_select(const int* stt, int* list, int lo, int hi, int& total) {
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  int base = 0;
#pragma unroll
  for (int j = 0; j < K; j++) {
    int slot = j * 64 + lane;
    int v = stt[slot];
    bool m = v >= lo && v <= hi;
    unsigned long long b = wballot(m);
    if (m) list[base + __popcll(b & below)] = slot;
    base += __popcll(b);
  }
  total = base;
  __syncthreads();                 // one wave per workgroup: orders the LDS writes above before the reads below
  return lane < total ? list[lane] : -1;
}

template <bool COUNT, int K>
__global__ __launch_bounds__(64) void k_pool(DevScene S, FrameParams P, float4* __restrict__ result, DevCounters* ctr,
                                             unsigned int* __restrict__ next) {
  constexpr int NP = 64 * K;
  extern __shared__ int lds[];
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  int* bl = lds + lane;                                // triangle-BVH stack of this LANE, entry e at bl[e * 64]
  int* pin = lds + P.blas_stack * 64;                  // [64] slot whose suspended traversal owns the lane's stack, or -1
  int* list = pin + 64;                                // [NP] compaction scratch
  int* stt = list + NP;                                // [NP] slot state
  int* pf = stt + NP;                                  // [fields][NP] path state
  const bool multi = P.num_rays > 1;
  int* tls = pf + (multi ? F_COUNTN : F_COUNT1) * NP;  // [tlas_stack][NP] object-level stack of each SLOT
#define PF(field, slot) pf[(field) * NP + (slot)]
#define PFf(field, slot) as_float(pf[(field) * NP + (slot)])
#define PFset(field, slot, val) pf[(field) * NP + (slot)] = as_int(val)
  pin[lane] = -1;
#pragma unroll
  for (int j = 0; j < K; j++) stt[j * 64 + lane] = PS_FREE;
  LocalCounters lc;
  const unsigned int ntiles = (unsigned int)(P.tiles_x * P.n_strips);
  WorkCursor wc; wc.shard = blockIdx.x & ((unsigned int)P.n_shards - 1u);
  bool exhausted = false, watchdog = false;
  unsigned int wave_iters = 0;
#ifdef URT_STAMPS
  unsigned long long ph_t[4] = {0, 0, 0, 0}, ph_lanes[5] = {0, 0, 0, 0, 0}, ph_trips[5] = {0, 0, 0, 0, 0};   // FRONT, BLAS, SHADE, blas inner, refill
  unsigned long long t_begin = wall_clock64(), t_dry = 0;
#endif

  for (;;) {
    if (watchdog) break;
#ifdef URT_STAMPS
    if (exhausted && !t_dry) t_dry = wall_clock64();
#endif
    __syncthreads();                                   // slot states written by other lanes during the last trip
    int nFree = 0, nFront = 0, nNew = 0, nPin = 0, nShade = 0;
#pragma unroll
    for (int j = 0; j < K; j++) {
      int v = stt[j * 64 + lane];
      nFree += __popcll(wballot(v == PS_FREE));
      nFront += __popcll(wballot(v == PS_FRONT || v == PS_RESUME));
      nNew += __popcll(wballot(v == PS_BLAS));
      nPin += __popcll(wballot(v == PS_PINNED));
      nShade += __popcll(wballot(v == PS_SHADE));
    }
    if (++wave_iters > P.sched_trips) { watchdog = true; break; }   // an exit every wave reaches, whatever the data
    const int busy = nFront + nNew + nPin + nShade;
    // ---- phase election ----
    // The triangle-BVH phase is the expensive one (hundreds of dependent steps per quantum, each costing the same whether
    // 8 or 64 lanes take part), so it waits until `blas_min` rays are queued for it; meanwhile the cheap phases run whenever
    // they have `pool_other_min` lanes of work, and free slots are refilled with new pixels.  Only when nothing reaches its
    // threshold does the fullest phase run.
    const int nB = nNew + nPin;
    const bool can_fetch = !exhausted && nFree > 0;
    int phase;
    if (nB >= P.blas_min) phase = PS_BLAS;
    else if (can_fetch && nFree >= P.refill_min) phase = PS_FREE;
    else if (nShade >= P.pool_other_min && nShade >= nFront) phase = PS_SHADE;
    else if (nFront >= P.pool_other_min) phase = PS_FRONT;
    else if (nShade >= P.pool_other_min) phase = PS_SHADE;
    else if (can_fetch) phase = PS_FREE;
    else if (busy == 0) break;                           // nothing in the pool and no work left to fetch
    else if (nB >= nShade && nB >= nFront) phase = PS_BLAS;
    else if (nShade >= nFront) phase = PS_SHADE;
    else phase = PS_FRONT;

    if (phase == PS_FREE) {
      // ---- new pixels into free slots (one atomic per refill) ----
      int total;
      int mine = pool_select<K>(stt, list, PS_FREE, PS_FREE, total);
#ifdef URT_STAMPS
      ph_trips[4]++; ph_lanes[4] += (unsigned long long)min(total, 64);
#endif
      int x = 0, y = 0;
      if (wave_fetch_pixels(P, wballot(mine >= 0), mine >= 0, next, ntiles, wc, exhausted, x, y)) {
        float seed = P.seed;
        v3 o, d;
        camera_ray<kPOffAfterScene>(P, x, y, seed, o, d);
        PF(F_PIX, mine) = x | (y << 16); PF(F_K, mine) = 0; PF(F_RAYI, mine) = 0; PFset(F_SEED, mine, seed);
        PFset(F_OX, mine, o.x); PFset(F_OY, mine, o.y); PFset(F_OZ, mine, o.z);
        PFset(F_DX, mine, d.x); PFset(F_DY, mine, d.y); PFset(F_DZ, mine, d.z);
        PFset(F_EX, mine, 1.0f); PFset(F_EY, mine, 1.0f); PFset(F_EZ, mine, 1.0f);
        PFset(F_RX, mine, 0.0f); PFset(F_RY, mine, 0.0f); PFset(F_RZ, mine, 0.0f);
        if (multi) { PFset(F_AX, mine, 0.0f); PFset(F_AY, mine, 0.0f); PFset(F_AZ, mine, 0.0f); }
        stt[mine] = PS_FRONT;
      }
      continue;
    }
#ifdef URT_STAMPS
    unsigned long long t_ph = wall_clock64();
    int ph_id = phase == PS_FRONT ? 0 : phase == PS_BLAS ? 1 : 2;
    ph_lanes[ph_id] += (unsigned long long)min(64, phase == PS_FRONT ? nFront : phase == PS_BLAS ? nB : nShade);
    ph_trips[ph_id]++;
#endif

    if (phase == PS_FRONT) {
      // ---------------- FRONT / RESUME: Trace() up to the next triangle-BVH visit (RS:364-383) ----------------
      int total;
      int mine = pool_select<K>(stt, list, PS_FRONT, PS_RESUME, total);
      if (mine >= 0) {
        bool fresh = stt[mine] == PS_FRONT;
        v3 o = mk3(PFf(F_OX, mine), PFf(F_OY, mine), PFf(F_OZ, mine)), d = mk3(PFf(F_DX, mine), PFf(F_DY, mine), PFf(F_DZ, mine));
        HitRec best; best.t = URT_INF; best.kid = 0; best.u = 0; best.v = 0;
        int check = 0; bool seen = false;
        if (!fresh) {
          int ki = PF(F_KINDID, mine), cs = PF(F_CHECK, mine);
          best.t = PFf(F_T, mine); best.kid = ki; best.u = PFf(F_U, mine); best.v = PFf(F_V, mine);
          check = cs >> 1; seen = (cs & 1) != 0;
        }
        int32_t cur = kBlasDone;
        bool need = trace_front<COUNT>(S, fresh, o, d, best, check, seen, tls + mine, NP, cur, lc);
        PFset(F_T, mine, best.t); PF(F_KINDID, mine) = best.kid; PFset(F_U, mine, best.u); PFset(F_V, mine, best.v);
        PF(F_CHECK, mine) = (check << 1) | (seen ? 1 : 0);
        if (need) { PF(F_CUR, mine) = cur; PF(F_SP, mine) = 0; PF(F_BESTI, mine) = -1; stt[mine] = PS_BLAS; }
        else stt[mine] = PS_SHADE;
      }
    } else if (phase == PS_BLAS) {
      // ---------------- BLAS: triangle BVH of one MeshObject per ray; lanes are re-fed from the waiting list ----------------
      int total;
      (void)pool_select<K>(stt, list, PS_BLAS, PS_BLAS, total);     // list[0, total) = the waiting rays, in slot order
      int taken = 0;
      int mys = pin[lane];                                          // a suspended traversal resumes on the lane that holds its stack
      const int n0 = min(64, nB);
      const int exit_below = (nShade + nFront > 0 || can_fetch) ? min(P.blas_exit, n0) : 1;
      v3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
      HitRec best; best.t = URT_INF; best.kid = 0; best.u = 0; best.v = 0;
      int32_t cur = kBlasDone; int sp = 0, best_i = -1;
      bool load = mys >= 0, first = true;
      BlasRay R = blas_ray(o, d);
      unsigned long long steps = 0;
      const unsigned long long step_cap = (unsigned long long)P.watchdog_steps * 64ull;   // between two re-feeds
      for (;;) {
        unsigned long long mA = wballot(mys >= 0);
        int nA = __popcll(mA);
        if (taken < total && (first || 64 - nA >= P.pool_inloop || nA < exit_below)) {     // feed the idle lanes
          int r = __popcll(~mA & below);
          if (mys < 0 && taken + r < total) { mys = list[taken + r]; load = true; }
          taken = min(total, taken + 64 - nA);
          steps = 0;
        }
        first = false;
        if (load) {
          int ki = PF(F_KINDID, mys);
          o = mk3(PFf(F_OX, mys), PFf(F_OY, mys), PFf(F_OZ, mys)); d = mk3(PFf(F_DX, mys), PFf(F_DY, mys), PFf(F_DZ, mys));
          best.t = PFf(F_T, mys); best.kid = ki; best.u = PFf(F_U, mys); best.v = PFf(F_V, mys);
          cur = PF(F_CUR, mys); sp = PF(F_SP, mys); best_i = PF(F_BESTI, mys);
          R = blas_ray(o, d);
          load = false;
        }
        mA = wballot(mys >= 0);
        nA = __popcll(mA);
        if (nA < exit_below) break;
        if (++steps > step_cap) { watchdog = true; break; }
#ifdef URT_STAMPS
        ph_trips[3]++; ph_lanes[3] += (unsigned long long)nA;
#endif
        // majority vote: this trip runs EITHER the interior-node step OR the leaf step (see mode 3)
        bool active = mys >= 0;
        bool interior = active && cur >= 0;
        int nI = __popcll(wballot(interior));
        if (nI >= nA - nI) {
          if (interior) cur = blas_node_step<COUNT>(S, cur, R, best.t, bl, sp, lc);
        } else if (active && !interior) {
          test_leaf<COUNT>(S, cur, o, d, best, best_i, lc);
          cur = blas_pop(bl, sp);
        }
        if (active && cur == kBlasDone) {                              // ray finished: back to the object-level walk (RS:323-325)
          PFset(F_T, mys, best.t); PF(F_KINDID, mys) = best.kid; PFset(F_U, mys, best.u); PFset(F_V, mys, best.v);
          stt[mys] = ((PF(F_CHECK, mys) >> 1) == 0 && S.n_spheres == 0) ? PS_SHADE : PS_RESUME;   // nothing of Trace() left: shade next
          pin[lane] = -1;
          mys = -1;
        }
      }
      if (mys >= 0) {                                                  // yield: the traversal stays pinned to this lane
        PFset(F_T, mys, best.t); PF(F_KINDID, mys) = best.kid; PFset(F_U, mys, best.u); PFset(F_V, mys, best.v);
        PF(F_CUR, mys) = cur; PF(F_SP, mys) = sp; PF(F_BESTI, mys) = best_i;
        stt[mys] = PS_PINNED;
        pin[lane] = mys;
      }
    } else {
      // ---------------- SHADE + bookkeeping of CSMain's loops (RS:444-468) ----------------
      int total;
      int mine = pool_select<K>(stt, list, PS_SHADE, PS_SHADE, total);
      if (mine >= 0) {
        int pix = PF(F_PIX, mine), k = PF(F_K, mine), ray_i = PF(F_RAYI, mine), ki = PF(F_KINDID, mine);
        int x = pix & 0xffff, y = (int)((unsigned)pix >> 16);
        float px = (float)x, py = (float)y, seed = PFf(F_SEED, mine);
        v3 o = mk3(PFf(F_OX, mine), PFf(F_OY, mine), PFf(F_OZ, mine)), d = mk3(PFf(F_DX, mine), PFf(F_DY, mine), PFf(F_DZ, mine));
        v3 energy = mk3(PFf(F_EX, mine), PFf(F_EY, mine), PFf(F_EZ, mine)), res = mk3(PFf(F_RX, mine), PFf(F_RY, mine), PFf(F_RZ, mine));
        HitRec best; best.t = PFf(F_T, mine); best.kid = ki; best.u = PFf(F_U, mine); best.v = PFf(F_V, mine);
        bool cont = shade<COUNT>(S, best, o, d, energy, res, seed, px, py, lc);
        k++;
        int nst = PS_FRONT;
        if (!cont || k >= P.num_bounces) {
          v3 avg = multi ? mk3(PFf(F_AX, mine), PFf(F_AY, mine), PFf(F_AZ, mine)) : mk3(0, 0, 0);
          avg = avg + res;
          ray_i++;
          if (ray_i < P.num_rays) {
            res = mk3(0, 0, 0); energy = mk3(1, 1, 1); k = 0;
            camera_ray<kPOffAfterScene>(P, x, y, seed, o, d);
            if (multi) { PFset(F_AX, mine, avg.x); PFset(F_AY, mine, avg.y); PFset(F_AZ, mine, avg.z); }
          } else {
            float n = (float)P.num_rays;
            st_result(result + (size_t)y * P.width + x, make_float4(avg.x / n, avg.y / n, avg.z / n, 1.0f));
            nst = PS_FREE;
          }
        }
        if (nst != PS_FREE) {
          PF(F_K, mine) = k; PF(F_RAYI, mine) = ray_i; PFset(F_SEED, mine, seed);
          PFset(F_OX, mine, o.x); PFset(F_OY, mine, o.y); PFset(F_OZ, mine, o.z);
          PFset(F_DX, mine, d.x); PFset(F_DY, mine, d.y); PFset(F_DZ, mine, d.z);
          PFset(F_EX, mine, energy.x); PFset(F_EY, mine, energy.y); PFset(F_EZ, mine, energy.z);
          PFset(F_RX, mine, res.x); PFset(F_RY, mine, res.y); PFset(F_RZ, mine, res.z);
        }
        stt[mine] = nst;
      }
    }
#ifdef URT_STAMPS
    ph_t[ph_id] += wall_clock64() - t_ph;
#endif
  }
#ifdef URT_STAMPS
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* sp_ = (unsigned long long*)(next + kWorkShards * 32);
    size_t w = (size_t)blockIdx.x * 32;
    for (int q = 0; q < 4; q++) { sp_[w + q] = ph_t[q]; sp_[w + 4 + q] = ph_lanes[q]; sp_[w + 8 + q] = ph_trips[q]; }
    sp_[w + 12] = t_begin; sp_[w + 13] = wall_clock64(); sp_[w + 14] = t_dry; sp_[w + 15] = 0;
    sp_[w + 16] = ph_trips[4]; sp_[w + 17] = ph_lanes[4]; sp_[w + 18] = wave_iters;
  }
#endif
#undef PF
#undef PFf
#undef PFset
  if (watchdog && (threadIdx.x & 63) == 0) report_watchdog(P, ctr + (blockIdx.x & (kCounterShards - 1)));
  flush_counters<COUNT>(lc, ctr);
}

// ---------------------------------------------------------------------------------------------------
// mode 1: wavefront pipeline.  generate -> (bounce x num_bounces) per ray index, over compacted queues.
// ---------------------------------------------------------------------------------------------------
// Append the alive lanes of this wave to a queue: ballot, prefix popcount, one atomic per wave.
__device__ __forceinline__ int wave_append(bool alive, unsigned int* counter) {
  unsigned long long m = wballot(alive);
  if (m == 0) return -1;
  int lane = threadIdx.x & 63;
  int leader = __ffsll((long long)m) - 1;
  unsigned int base = 0;
  if (lane == leader) base = atomicAdd(counter, (unsigned int)__popcll(m));
  base = __shfl(base, leader, 64);
  int rank = __popcll(m & ((1ull << lane) - 1ull));
  return alive ? (int)(base + rank) : -1;
}

// Path finished: fold its radiance into the pixel.  Result.xyz holds the running resultAverage
// (RS:441,464) and .w the running _Seed between the rays of one pixel; the last ray writes RS:468.
__device__ __forceinline__ void finish_path(const FrameParams& P, float4* result, int pixel, int ray_index, v3 res, float seed) {
  int x = pixel & 0xffff, y = (unsigned)pixel >> 16;
  size_t at = (size_t)y * P.width + x;
  v3 avg = res;
  if (ray_index > 0) { float4 prev = result[at]; avg = xyz(prev) + res; }
  if (ray_index == P.num_rays - 1) {
    float n = (float)P.num_rays;
    result[at] = make_float4(avg.x / n, avg.y / n, avg.z / n, 1.0f);
  } else {
    result[at] = make_float4(avg.x, avg.y, avg.z, seed);
  }
}

__global__ __launch_bounds__(256) void k_generate(FrameParams P, PathQueues Q, const float4* __restrict__ result,
                                                  int ray_index, DevCounters* ctr) {
  int x = 0, y = 0;
  bool ok = tile_pixel(P, x, y);
  float seed = P.seed;
  v3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
  if (ok) {
    if (ray_index > 0) seed = result[(size_t)y * P.width + x].w;
    camera_ray<0>(P, x, y, seed, o, d);
  }
  unsigned int* cnt = Q.counts + (size_t)ray_index * (P.num_bounces + 1);
  int slot = wave_append(ok, cnt);
  if (ok) {
    Q.s[0][0][slot] = make_float4(o.x, o.y, o.z, seed);
    Q.s[0][1][slot] = make_float4(d.x, d.y, d.z, as_float((y << 16) | x));
    Q.s[0][2][slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    Q.s[0][3][slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
}

template <bool COUNT>
__global__ __launch_bounds__(256) void k_bounce(DevScene S, FrameParams P, PathQueues Q, float4* __restrict__ result,
                                                int ray_index, int bounce, DevCounters* ctr) {
  int *tl, *bl;
  lane_stacks(P, tl, bl);
  LocalCounters lc;
  unsigned int* cnt = Q.counts + (size_t)ray_index * (P.num_bounces + 1) + bounce;
  unsigned int n_in = cnt[0];
  unsigned int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int in = bounce & 1, out = in ^ 1;
  bool alive = false;
  v3 o, d, energy, res; float seed = 0; int pixel = 0;
  if (gid < n_in) {
    float4 s0 = Q.s[in][0][gid], s1 = Q.s[in][1][gid], s2 = Q.s[in][2][gid], s3 = Q.s[in][3][gid];
    o = xyz(s0); seed = s0.w; d = xyz(s1); pixel = as_int(s1.w); energy = xyz(s2); res = xyz(s3);
    float px = (float)(pixel & 0xffff), py = (float)((unsigned)pixel >> 16);
    HitRec h = trace<COUNT>(S, o, d, tl, bl, lc);
    alive = shade<COUNT>(S, h, o, d, energy, res, seed, px, py, lc);
    if (bounce == P.num_bounces - 1) alive = false;      // loop bound RS:453
    if (!alive) finish_path(P, result, pixel, ray_index, res, seed);
  }
  int slot = wave_append(alive, cnt + 1);
  if (alive) {
    Q.s[out][0][slot] = make_float4(o.x, o.y, o.z, seed);
    Q.s[out][1][slot] = make_float4(d.x, d.y, d.z, as_float(pixel));
    Q.s[out][2][slot] = make_float4(energy.x, energy.y, energy.z, 0.0f);
    Q.s[out][3][slot] = make_float4(res.x, res.y, res.z, 0.0f);
  }
  flush_counters<COUNT>(lc, ctr);
}

// ---------------------------------------------------------------------------------------------------
// AdditionShader — AS:9,39-41 as driven by RM:817-818.  dst = src*a + dst*(1-a), a = 1/(sample+1);
// the fragment's alpha is a itself.  16 B per lane, grid-stride.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blit_add(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, float sample) {
  float a = 1.0f / (sample + 1.0f);
  float ia = 1.0f - a;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 t = src[i], c = dst[i];
    c.x = t.x * a + c.x * ia;
    c.y = t.y * a + c.y * ia;
    c.z = t.z * a + c.z * ia;
    c.w = a * a + c.w * ia;
    dst[i] = c;
  }
}

// n consecutive blends in one pass (frames of a batched launch): per pixel the SAME operations in the same order as n
// k_blit_add launches, with 16 (n + 2) bytes of traffic per pixel instead of 48 n.
// `present` (may be null): the image the host presents the accumulated frame to after every blend (Graphics.Blit(_converged,
// destination), RM:819).  Of the n presents of a fused run only the last is observable (every observer of `present` submits the
// deferred work first, context.cpp), so the last blended value is stored to both images: the bytes a copy of dst would carry.
struct BlendSamples { float s[kMaxFramesPerLaunch]; };
__global__ __launch_bounds__(256) void k_blit_add_multi(const float4* __restrict__ src, size_t frame_stride, int n, BlendSamples smp,
                                                        float4* __restrict__ dst, float4* __restrict__ present, size_t npix) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    float4 c = dst[i];
    for (int f = 0; f < n; f++) {
      float a = 1.0f / (smp.s[f] + 1.0f);
      float ia = 1.0f - a;
      float4 t = src[(size_t)f * frame_stride + i];
      c.x = t.x * a + c.x * ia;
      c.y = t.y * a + c.y * ia;
      c.z = t.z * a + c.z * ia;
      c.w = a * a + c.w * ia;
    }
    dst[i] = c;
    if (present) present[i] = c;
  }
}

// strips <-> dense buffer (frame-end gather): strip j of this rank = pixel rows (first + j*stride)*8 .. +8
__global__ __launch_bounds__(256) void k_pack_rows(const float4* __restrict__ img, float4* __restrict__ dense, int width, int height,
                                                   int first_group_row, int row_stride, int n_strips, int to_dense) {
  size_t per_strip = (size_t)width * 8;
  size_t n = per_strip * (size_t)n_strips;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int j = (int)(i / per_strip);
    size_t r = i - (size_t)j * per_strip;
    int row = (first_group_row + j * row_stride) * 8 + (int)(r / (size_t)width);
    int col = (int)(r % (size_t)width);
    if (row < height) {
      size_t at = (size_t)row * width + col;
      if (to_dense) dense[i] = img[at]; else const_cast<float4*>(img)[at] = dense[i];
    } else if (to_dense) {
      dense[i] = make_float4(0, 0, 0, 0);
    }
  }
}

// The same strips with THREE channels per pixel (12 B): the frame-end gather of a running mean need not move its alpha channel — after
// sample n it is the same value in every pixel, a function of the sample sequence alone (AS:40: the fragment's alpha is a itself and is
// blended like the colours) — so the root writes `alpha` itself when it de-interleaves.  A quarter of the gather's bytes less.
__global__ __launch_bounds__(256) void k_pack_rows_rgb(const float4* __restrict__ img, float* __restrict__ dense, int width, int height,
                                                       int first_group_row, int row_stride, int n_strips, int to_dense, float alpha) {
  size_t per_strip = (size_t)width * 8;
  size_t n = per_strip * (size_t)n_strips;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int j = (int)(i / per_strip);
    size_t r = i - (size_t)j * per_strip;
    int row = (first_group_row + j * row_stride) * 8 + (int)(r / (size_t)width);
    int col = (int)(r % (size_t)width);
    float* d = dense + 3 * i;
    if (row < height) {
      size_t at = (size_t)row * width + col;
      if (to_dense) { float4 v = img[at]; d[0] = v.x; d[1] = v.y; d[2] = v.z; }
      else const_cast<float4*>(img)[at] = make_float4(d[0], d[1], d[2], alpha);
    } else if (to_dense) {
      d[0] = 0; d[1] = 0; d[2] = 0;
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// host-side launchers (declared in kernels.h)
// ---------------------------------------------------------------------------------------------------
namespace urtd {

static inline int blocks_for_tiles(const FrameParams& P) {
  int waves = P.block_threads / 64;
  int ntiles = P.tiles_x * P.n_strips;
  int nblocks = (ntiles + waves - 1) / waves;
  int q = 8 * P.xcd_run;                      // the block permutation of tile_pixel() acts on windows of 8*G blocks
  return ((nblocks + q - 1) / q) * q;
}

static inline size_t stack_lds_bytes(const FrameParams& P) {
  return (size_t)(P.tlas_stack + P.blas_stack) * 64 * (size_t)(P.block_threads / 64) * sizeof(int);
}

// What the last trace launch of this process was (urt_debug_launch_info): the instantiation's name as rocprofv3 prints it, its grid
// and its dynamic LDS.  Written by the launchers below on the caller's (single) host thread; context.cpp copies it per context.
static TraceLaunchRecord g_last_trace;
const TraceLaunchRecord& last_trace_launch() { return g_last_trace; }
static void note_launch(int n_blocks, int block_threads, size_t lds, const char* fmt, ...) __attribute__((format(printf, 4, 5)));
static void note_launch(int n_blocks, int block_threads, size_t lds, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_last_trace.kernel, sizeof g_last_trace.kernel, fmt, ap);
  va_end(ap);
  g_last_trace.n_blocks = n_blocks; g_last_trace.block_threads = block_threads; g_last_trace.lds_bytes = (int)lds;
}
static const char* tf(bool b) { return b ? "true" : "false"; }

// dynamic LDS above the default 64 KiB per workgroup (very deep BVHs): the kernel's limit has to be raised first
template <typename K>
static hipError_t allow_lds(K kernel, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t launch_mega(const DevScene& S, const FrameParams& P, float4* result, DevCounters* ctr, bool count, hipStream_t st) {
  int nb = blocks_for_tiles(P);
  if (nb == 0) return hipSuccess;
  size_t lds = stack_lds_bytes(P);
  hipError_t ea = count ? allow_lds(k_mega<true>, lds) : allow_lds(k_mega<false>, lds);
  if (ea != hipSuccess) return ea;
  if (count) hipLaunchKernelGGL(k_mega<true>, dim3(nb), dim3(P.block_threads), lds, st, S, P, result, ctr);
  else hipLaunchKernelGGL(k_mega<false>, dim3(nb), dim3(P.block_threads), lds, st, S, P, result, ctr);
  note_launch(nb, P.block_threads, lds, "k_mega<%s>", tf(count));
  return hipGetLastError();
}

hipError_t launch_wavefront(const DevScene& S, const FrameParams& P, const PathQueues& Q, float4* result, DevCounters* ctr,
                            bool count, hipStream_t st) {
  int nb = blocks_for_tiles(P);
  if (nb == 0) return hipSuccess;
  size_t lds = stack_lds_bytes(P);
  size_t n_counts = (size_t)P.num_rays * (P.num_bounces + 1);
  hipError_t e = hipMemsetAsync(Q.counts, 0, n_counts * sizeof(unsigned int), st);
  if (e != hipSuccess) return e;
  e = count ? allow_lds(k_bounce<true>, lds) : allow_lds(k_bounce<false>, lds);
  if (e != hipSuccess) return e;
  size_t npix = (size_t)P.region_w * 8 * P.n_strips;
  int bt = P.block_threads;
  int nbb = (int)((npix + bt - 1) / bt);
  for (int i = 0; i < P.num_rays; i++) {
    hipLaunchKernelGGL(k_generate, dim3(nb), dim3(bt), 0, st, P, Q, (const float4*)result, i, ctr);
    for (int k = 0; k < P.num_bounces; k++) {
      if (count) hipLaunchKernelGGL(k_bounce<true>, dim3(nbb), dim3(bt), lds, st, S, P, Q, result, i, k, ctr);
      else hipLaunchKernelGGL(k_bounce<false>, dim3(nbb), dim3(bt), lds, st, S, P, Q, result, i, k, ctr);
    }
  }
  note_launch(nbb, bt, lds, "k_generate + k_bounce<%s> x %d", tf(count), P.num_rays * P.num_bounces);
  return hipGetLastError();
}

hipError_t launch_persist(const DevScene& S, const FrameParams& P, float4* result, DevCounters* ctr, unsigned int* next,
                          int n_blocks, bool count, hipStream_t st) {
  if (n_blocks <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(next, 0, kWorkShards * 32 * sizeof(unsigned int), st);
  if (e != hipSuccess) return e;
  size_t lds = stack_lds_bytes(P);
  e = count ? allow_lds(k_persist<true>, lds) : allow_lds(k_persist<false>, lds);
  if (e != hipSuccess) return e;
  if (count) hipLaunchKernelGGL(k_persist<true>, dim3(n_blocks), dim3(P.block_threads), lds, st, S, P, result, ctr, next);
  else hipLaunchKernelGGL(k_persist<false>, dim3(n_blocks), dim3(P.block_threads), lds, st, S, P, result, ctr, next);
  note_launch(n_blocks, P.block_threads, lds, "k_persist<%s>", tf(count));
  return hipGetLastError();
}

size_t sched_lds_bytes(const DevScene& S, const FrameParams& P) {
  size_t f4 = (size_t)P.top_nodes * 4;
  if (P.lds_mesh) f4 += 2 * (size_t)S.n_mesh_tlas + ((size_t)S.n_meshes + 3) / 4;
  if (P.walk_f4 > 0) f4 += (size_t)P.walk_f4 + (P.lds_small ? 3 * (size_t)S.n_small : 0);       // masked FRONT: the walk table replaces the heap (lds_mesh = 0)
  else if (P.lds_small) f4 += ((size_t)S.n_meshes + 3) / 4 + 3 * (size_t)S.n_small;
  if (P.lds_sphere) f4 += 2 * (size_t)S.n_sphere_tlas + (size_t)S.n_spheres;
  if (P.serve) f4 += 2 * (size_t)P.block_threads + 1;          // mode 5: the workgroup's mailbox (k_serve)
  return f4 * 16 + (size_t)(P.tlas_stack + P.blas_stack) * 64 * (size_t)(P.block_threads / 64) * sizeof(int);
}

template <bool COUNT, int BLOCK, int FMODE, bool MULTI, bool QN>
static hipError_t launch_sched_q(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                                 unsigned int* next, int n_blocks, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_sched<COUNT, BLOCK, FMODE, MULTI, QN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((k_sched<COUNT, BLOCK, FMODE, MULTI, QN>), dim3(n_blocks), dim3(BLOCK), lds, st, S, P, T, result, ctr, next);
  note_launch(n_blocks, BLOCK, lds, "k_sched<%s, %d, %d, %s, %s>", tf(COUNT), BLOCK, FMODE, tf(MULTI), tf(QN));
  return hipGetLastError();
}

template <bool COUNT, int BLOCK, int FMODE, bool MULTI>
static hipError_t launch_sched_t(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                                 unsigned int* next, int n_blocks, size_t lds, hipStream_t st) {
  if (!COUNT && S.blas_qnodes) return launch_sched_q<false, BLOCK, FMODE, MULTI, true>(S, P, T, result, ctr, next, n_blocks, lds, st);
  return launch_sched_q<COUNT, BLOCK, FMODE, MULTI, false>(S, P, T, result, ctr, next, n_blocks, lds, st);
}

template <bool COUNT, int BLOCK, int FMODE>
static hipError_t launch_sched_m(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                                 unsigned int* next, int n_blocks, size_t lds, hipStream_t st) {
  return P.num_rays > 1 ? launch_sched_t<COUNT, BLOCK, FMODE, true>(S, P, T, result, ctr, next, n_blocks, lds, st)
                        : launch_sched_t<COUNT, BLOCK, FMODE, false>(S, P, T, result, ctr, next, n_blocks, lds, st);
}

template <bool COUNT, int BLOCK>
static hipError_t launch_sched_b(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                                 unsigned int* next, int n_blocks, size_t lds, int front_mode, hipStream_t st) {
  if (front_mode == 3) return launch_sched_m<COUNT, BLOCK, 3>(S, P, T, result, ctr, next, n_blocks, lds, st);
  if (front_mode == 2) return launch_sched_m<COUNT, BLOCK, 2>(S, P, T, result, ctr, next, n_blocks, lds, st);
  if (front_mode == 1) return launch_sched_m<COUNT, BLOCK, 1>(S, P, T, result, ctr, next, n_blocks, lds, st);
  return launch_sched_m<COUNT, BLOCK, 0>(S, P, T, result, ctr, next, n_blocks, lds, st);
}

hipError_t launch_sched(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                        unsigned int* next, int n_blocks, int front_mode, bool count, hipStream_t st) {
  if (n_blocks <= 0) return hipSuccess;
  if (P.block_threads != 64 && P.block_threads != 256) return hipErrorInvalidValue;   // independent waves; a workgroup shares the LDS top-of-tree copy
  if (P.n_frames < 1 || P.n_frames > kMaxFramesPerLaunch) return hipErrorInvalidValue;
  if (front_mode == 2 && (!P.lds_mesh || S.n_meshes > 12 || P.tlas_stack < 2)) return hipErrorInvalidValue;
  if (front_mode == 3 && (P.walk_f4 < 20 || P.lds_mesh || P.top_nodes <= 0 || S.n_mesh_tlas > 31)) return hipErrorInvalidValue;
  if (front_mode != 3 && P.walk_f4 != 0) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(next, 0, kWorkShards * 32 * sizeof(unsigned int), st);
  if (e != hipSuccess) return e;
  size_t lds = sched_lds_bytes(S, P);
  if (P.top_nodes <= 0 && front_mode == 1) front_mode = 0;
  if (P.block_threads == 64) return count ? launch_sched_b<true, 64>(S, P, T, result, ctr, next, n_blocks, lds, front_mode, st)
                                          : launch_sched_b<false, 64>(S, P, T, result, ctr, next, n_blocks, lds, front_mode, st);
  return count ? launch_sched_b<true, 256>(S, P, T, result, ctr, next, n_blocks, lds, front_mode, st)
               : launch_sched_b<false, 256>(S, P, T, result, ctr, next, n_blocks, lds, front_mode, st);
}

template <bool COUNT, int FMODE, bool MULTI>
static hipError_t launch_serve_t(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                                 unsigned int* next, float4* mail, int n_blocks, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_serve<COUNT, 256, FMODE, MULTI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((k_serve<COUNT, 256, FMODE, MULTI>), dim3(n_blocks), dim3(256), lds, st, S, P, T, result, ctr, next, mail);
  note_launch(n_blocks, 256, lds, "k_serve<%s, 256, %d, %s>", tf(COUNT), FMODE, tf(MULTI));
  return hipGetLastError();
}

template <bool COUNT, int FMODE>
static hipError_t launch_serve_m(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                                 unsigned int* next, float4* mail, int n_blocks, size_t lds, hipStream_t st) {
  return P.num_rays > 1 ? launch_serve_t<COUNT, FMODE, true>(S, P, T, result, ctr, next, mail, n_blocks, lds, st)
                        : launch_serve_t<COUNT, FMODE, false>(S, P, T, result, ctr, next, mail, n_blocks, lds, st);
}

template <bool COUNT>
static hipError_t launch_serve_b(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                                 unsigned int* next, float4* mail, int n_blocks, size_t lds, int front_mode, hipStream_t st) {
  if (front_mode == 2) return launch_serve_m<COUNT, 2>(S, P, T, result, ctr, next, mail, n_blocks, lds, st);
  if (front_mode == 1) return launch_serve_m<COUNT, 1>(S, P, T, result, ctr, next, mail, n_blocks, lds, st);
  return launch_serve_m<COUNT, 0>(S, P, T, result, ctr, next, mail, n_blocks, lds, st);
}

hipError_t launch_serve(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                        unsigned int* next, float4* mail, int n_blocks, int front_mode, bool count, hipStream_t st) {
  if (n_blocks <= 0) return hipSuccess;
  if (P.block_threads != 256 || !P.serve || !mail) return hipErrorInvalidValue;
  if (P.n_frames < 1 || P.n_frames > kMaxFramesPerLaunch) return hipErrorInvalidValue;
  if (front_mode == 2 && (!P.lds_mesh || S.n_meshes > 12 || P.tlas_stack < 2)) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(next, 0, kWorkShards * 32 * sizeof(unsigned int), st);
  if (e != hipSuccess) return e;
  size_t lds = sched_lds_bytes(S, P);
  if (P.top_nodes <= 0 && front_mode == 1) front_mode = 0;
  return count ? launch_serve_b<true>(S, P, T, result, ctr, next, mail, n_blocks, lds, front_mode, st)
               : launch_serve_b<false>(S, P, T, result, ctr, next, mail, n_blocks, lds, front_mode, st);
}

size_t pool_lds_bytes(const FrameParams& P, int k) {
  size_t np = (size_t)64 * (size_t)k;
  size_t fields = P.num_rays > 1 ? F_COUNTN : F_COUNT1;
  return ((size_t)P.blas_stack * 64 + 64 + np + np + fields * np + (size_t)P.tlas_stack * np) * sizeof(int);
}

template <int K>
static hipError_t launch_pool_k(const DevScene& S, const FrameParams& P, float4* result, DevCounters* ctr, unsigned int* next,
                                int n_blocks, bool count, hipStream_t st) {
  size_t lds = pool_lds_bytes(P, K);
  if (lds > 64 * 1024) {            // above the default dynamic-LDS limit the kernel attribute has to be raised
    hipError_t e = count ? hipFuncSetAttribute((const void*)k_pool<true, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                         : hipFuncSetAttribute((const void*)k_pool<false, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (count) hipLaunchKernelGGL((k_pool<true, K>), dim3(n_blocks), dim3(64), lds, st, S, P, result, ctr, next);
  else hipLaunchKernelGGL((k_pool<false, K>), dim3(n_blocks), dim3(64), lds, st, S, P, result, ctr, next);
  note_launch(n_blocks, 64, lds, "k_pool<%s, %d>", tf(count), K);
  return hipGetLastError();
}

hipError_t launch_pool(const DevScene& S, const FrameParams& P, float4* result, DevCounters* ctr, unsigned int* next,
                       int n_blocks, int k, bool count, hipStream_t st) {
  if (n_blocks <= 0) return hipSuccess;
  if (P.width > 65535 || P.height > 65535 || k < 1 || k > 4) return hipErrorInvalidValue;   // pixel packed as y << 16 | x
  hipError_t e = hipMemsetAsync(next, 0, kWorkShards * 32 * sizeof(unsigned int), st);
  if (e != hipSuccess) return e;
  switch (k) {
    case 1: return launch_pool_k<1>(S, P, result, ctr, next, n_blocks, count, st);
    case 2: return launch_pool_k<2>(S, P, result, ctr, next, n_blocks, count, st);
    case 3: return launch_pool_k<3>(S, P, result, ctr, next, n_blocks, count, st);
    default: return launch_pool_k<4>(S, P, result, ctr, next, n_blocks, count, st);
  }
}

hipError_t launch_blit_add(const float4* src, float4* dst, size_t n_pixels, float sample, hipStream_t st) {
  if (n_pixels == 0) return hipSuccess;
  size_t nb = (n_pixels + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_blit_add, dim3((unsigned)nb), dim3(256), 0, st, src, dst, n_pixels, sample);
  return hipGetLastError();
}

hipError_t launch_blit_add_multi(const float4* src, size_t frame_stride, int n, const float* samples, float4* dst, float4* present,
                                 size_t n_pixels, hipStream_t st) {
  if (n_pixels == 0 || n <= 0) return hipSuccess;
  if (n > kMaxFramesPerLaunch) return hipErrorInvalidValue;
  BlendSamples smp{};
  for (int f = 0; f < n; f++) smp.s[f] = samples[f];
  size_t nb = (n_pixels + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_blit_add_multi, dim3((unsigned)nb), dim3(256), 0, st, src, frame_stride, n, smp, dst, present, n_pixels);
  return hipGetLastError();
}

hipError_t launch_pack_rows(float4* img, float4* dense, int width, int height, int first_group_row, int row_stride,
                            int n_strips, bool to_dense, hipStream_t st) {
  size_t n = (size_t)width * 8 * (size_t)n_strips;
  if (n == 0) return hipSuccess;
  size_t nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)nb), dim3(256), 0, st, (const float4*)img, dense, width, height,
                     first_group_row, row_stride, n_strips, to_dense ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_pack_rows_rgb(float4* img, float* dense, int width, int height, int first_group_row, int row_stride,
                                int n_strips, bool to_dense, float alpha, hipStream_t st) {
  size_t n = (size_t)width * 8 * (size_t)n_strips;
  if (n == 0) return hipSuccess;
  size_t nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_pack_rows_rgb, dim3((unsigned)nb), dim3(256), 0, st, (const float4*)img, dense, width, height,
                     first_group_row, row_stride, n_strips, to_dense ? 1 : 0, alpha);
  return hipGetLastError();
}

}  // namespace urtd
