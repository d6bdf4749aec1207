// oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Scalar CPU restatement of the reference's hot path, used only as the checker by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The shipped path
// (unityraytracer_amd/csrc) never includes, links or calls anything in this directory.
//
// What it restates (reference = RemyMuj/UnityRayTracer @ 2024_10_08):
//   RS = Assets/Shaders/RayTraceShader.compute  (every function; cited per function below)
//   AS = Assets/Shaders/AdditionShader.shader:9,39-41  (progressive accumulation)
// including the quirks of SURVEY.md Appendix A (A.1 RNG state, A.3 energy-before-Shade, A.4 object-
// space normals + back-face culling + strict closer-hit, A.5 `tests` never reset, A.6 Shade).
// Structure is literal: one Tracer per pixel with the shader's own mutable globals (_Pixel, _Seed),
// the same loops in the same order.  Numerics come from include/urt_math.h — the normative
// definition of the intrinsics HLSL leaves to the driver.
//
// PARITY STATUS: "parity unpinned".  The reference has no tests, golden vectors or fixtures for this
// path and cannot be executed here (HLSL inside the closed-source Unity editor;
// no HLSL/C# toolchain in the image).  This restatement is therefore the normative definition that
// the HIP path is compared against; its own known-answer tests are in tests/test_oracle_*.py.
//
// Two ways to intersect a MeshObject, selected by `mode`:
//   0  brute force over every triangle, exactly RS:237-268 (ground truth, O(tris) per ray);
//   1  the same triangle test, but candidates are enumerated through a caller-supplied triangle
//      BVH ("BLAS": built by the product and read back through urt_debug_get_blas, or by
//      oracle_build_blas below).  A conservative BVH only skips triangles that cannot win, and the
//      (t, index) rule below reproduces the brute-force winner; tests prove mode 1 == mode 0.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

#include "../include/urt_math.h"
#include "../include/urt_types.h"

using namespace urt;

extern "C" {

struct OracleScene {
  const urt_MeshObject* mesh_objects; int32_t n_mesh_objects;   // _MeshObjects  RS:64
  const float* vertices;              int32_t n_vertices;       // _Vertices     RS:65
  const int32_t* indices;             int32_t n_indices;        // _Indices      RS:66
  const float* normals;               int32_t n_normals;        // _Normals      RS:67
  const urt_Sphere* spheres;          int32_t n_spheres;        // _Spheres      RS:68
  const urt_BVHNode* mesh_bvh;        int32_t n_mesh_bvh;       // _MeshBVH      RS:70
  const urt_BVHNode* sphere_bvh;      int32_t n_sphere_bvh;     // _SphereBVH    RS:71
  const float* sky;                   int32_t sky_w, sky_h;     // _SkyboxTexture RS:9 (RGBA32F, row 0 = bottom)
  float camera_to_world[16];                                    // RS:5
  float camera_inverse_projection[16];                          // RS:6
  float pixel_offset[2];                                        // RS:7
  float seed;                                                   // RS:16
  int32_t num_bounces, num_rays;                                // RS:18-19
  int32_t width, height;                                        // Result.GetDimensions RS:438
  // mode 1 only: triangle BVH (layout documented in DESIGN.md "BLAS")
  const float* blas_nodes;            int32_t n_blas_nodes;     // 16 floats per node
  const int32_t* blas_tri_index;      int32_t n_blas_tris;      // leaf order -> index slot i (RS:243)
  const int32_t* blas_mesh_root;                                // per MeshObject
  // mode 1 only: per MeshObject, 1 = the product's object-level cull applies to it (include/urt_math.h tlas_cull; filled by
  // oracle_compute_cull_ok with the product's rule, csrc/cullflags.hip), or NULL = every popped object is intersected (the reference)
  const int32_t* mesh_cull_ok;
};

struct OracleCounters {
  uint64_t rays, tlas_nodes, blas_nodes, tri_tests, sphere_tests;
  uint64_t hit_tri, hit_sphere, hit_ground, hit_sky, pixels;
  uint64_t max_ray_steps;       // diagnostics: most BVH nodes + triangle tests any single Trace() needed
  uint64_t rays_over_256_steps; // diagnostics: Trace() calls that needed more than 256 of them
};

}  // extern "C"

namespace {

struct Ray { v3 origin, direction, energy; };                        // RS:23-27
struct RayTraceParams { v3 color_albedo, color_specular, emission; float smoothness; };  // RS:29-34
struct RayHit { v3 position; float distance; v3 normal; RayTraceParams lighting; int kind; };  // RS:36-41 (+kind for counters)

static inline v3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
static inline RayTraceParams ldparams(const urt_RayTraceParams& p) {
  RayTraceParams r;
  r.color_albedo = ld3(p.color_albedo); r.color_specular = ld3(p.color_specular);
  r.emission = ld3(p.emission); r.smoothness = p.smoothness;
  return r;
}

static int g_literal_division = 0;      // test-only variant switch, see Tracer::IntersectBVHNode

struct Tracer {
  const OracleScene& S;
  int mode;
  OracleCounters C{};
  // the shader's mutable globals (RS:15-16)
  float _PixelX = 0, _PixelY = 0;
  float _Seed = 0;

  Tracer(const OracleScene& s, int m) : S(s), mode(m) {}

  // RS:77-81
  float rand() { return rand_next(_Seed, _PixelX, _PixelY); }

  // RS:84-86
  static float sdot(v3 x, v3 y, float f = 1.0f) { return f_saturate(dot(x, y) * f); }

  // RS:103-111 with GetTangentSpace RS:89-100 inlined at the call (rows tangent, binormal, normal)
  v3 SampleHemisphere(v3 normal, float alpha) {
    float cosTheta = f_pow(rand(), 1.0f / (alpha + 1.0f));
    float sinTheta = f_sqrt(1.0f - cosTheta * cosTheta);
    float phi = (2.0f * kPI) * rand();
    float sp, cp; f_sincos(phi, sp, cp);
    v3 ts = mk3(cp * sinTheta, sp * sinTheta, cosTheta);
    // GetTangentSpace
    v3 helper = mk3(1, 0, 0);
    if (f_abs(normal.x) > 0.99f) helper = mk3(0, 0, 1);
    v3 tangent = normalize(cross(normal, helper));
    v3 binormal = normalize(cross(normal, tangent));
    // mul(row vector, float3x3(tangent, binormal, normal))  RS:110
    return mk3(f_fma(ts.z, normal.x, f_fma(ts.y, binormal.x, ts.x * tangent.x)),
               f_fma(ts.z, normal.y, f_fma(ts.y, binormal.y, ts.x * tangent.y)),
               f_fma(ts.z, normal.z, f_fma(ts.y, binormal.z, ts.x * tangent.z)));
  }

  // RS:114-120
  static Ray CreateRay(v3 o, v3 d) { Ray r; r.origin = o; r.direction = d; r.energy = mk3(1, 1, 1); return r; }

  // RS:123-139
  static RayHit CreateRayHit() {
    RayHit h;
    h.position = mk3(0, 0, 0); h.distance = URT_INF; h.normal = mk3(0, 0, 0);
    h.lighting.color_albedo = mk3(0, 0, 0); h.lighting.color_specular = mk3(0, 0, 0);
    h.lighting.smoothness = 0; h.lighting.emission = mk3(0, 0, 0);
    h.kind = 0;
    return h;
  }

  // RS:142-153
  Ray CreateCameraRay(float u, float v) {
    v3 origin = mul_m4(S.camera_to_world, 0.0f, 0.0f, 0.0f, 1.0f);
    v3 direction = mul_m4(S.camera_inverse_projection, u, v, 0.0f, 1.0f);
    direction = mul_m4(S.camera_to_world, direction.x, direction.y, direction.z, 0.0f);
    direction = normalize(direction);
    return CreateRay(origin, direction);
  }

  // RS:156-172
  void IntersectGroundPlane(const Ray& ray, RayHit& bestHit) {
    float t = -ray.origin.y / ray.direction.y;
    if (t > 0 && t < bestHit.distance) {
      bestHit.distance = t;
      bestHit.position = madd(t, ray.direction, ray.origin);
      bestHit.normal = mk3(0, 1, 0);
      bestHit.lighting.color_albedo = mk3(0.5f, 0.3f, 0.15f);
      bestHit.lighting.color_specular = mk3(0, 0, 0);
      bestHit.lighting.smoothness = 0.3f;
      bestHit.lighting.emission = mk3(0, 0, 0);
      bestHit.kind = 1;
    }
  }

  // RS:175-196
  void IntersectSphere(const Ray& ray, RayHit& bestHit, const urt_Sphere& sphere) {
    C.sphere_tests++;
    v3 c = ld3(sphere.position);
    v3 d = ray.origin - c;
    float p1 = -dot(ray.direction, d);
    float p2sqr = p1 * p1 - dot(d, d) + sphere.radius * sphere.radius;
    if (p2sqr < 0) return;
    float p2 = f_sqrt(p2sqr);
    float t = p1 - p2 > 0 ? p1 - p2 : p1 + p2;
    if (t > 0 && t < bestHit.distance) {
      bestHit.distance = t;
      bestHit.position = madd(t, ray.direction, ray.origin);
      bestHit.normal = normalize(bestHit.position - c);
      bestHit.lighting = ldparams(sphere.lighting);
      bestHit.kind = 2;
    }
  }

  // RS:199-234
  static bool IntersectTriangle_MT97(const Ray& ray, v3 vert0, v3 vert1, v3 vert2, float& t, float& u, float& v) {
    v3 edge1 = vert1 - vert0;
    v3 edge2 = vert2 - vert0;
    v3 pvec = cross(ray.direction, edge2);
    float det = dot(edge1, pvec);
    if (det < kEPSILON) return false;
    float inv_det = 1.0f / det;
    v3 tvec = ray.origin - vert0;
    u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    v3 qvec = cross(tvec, edge1);
    v = dot(ray.direction, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot(edge2, qvec) * inv_det;
    return true;
  }

  // One iteration of the loop body RS:243-266 for index slot i.  `tie_ok` is false in the literal
  // loop (strict t < best); in BVH order it says "the current best came from THIS call and from a
  // higher i", in which case an equal t must win because brute force would have met i first.
  void TestTriangle(const Ray& ray, RayHit& bestHit, const urt_MeshObject& mo, uint32_t i, int& best_i) {
    C.tri_tests++;
    const float* M = mo.localToWorldMatrix;
    const float* V = S.vertices;
    const int32_t* I = S.indices;
    v3 p0 = ld3(V + 3 * I[i]), p1 = ld3(V + 3 * I[i + 1]), p2 = ld3(V + 3 * I[i + 2]);
    v3 v0 = mul_m4(M, p0.x, p0.y, p0.z, 1.0f);
    v3 v1 = mul_m4(M, p1.x, p1.y, p1.z, 1.0f);
    v3 v2 = mul_m4(M, p2.x, p2.y, p2.z, 1.0f);
    float t, u, v;
    if (IntersectTriangle_MT97(ray, v0, v1, v2, t, u, v)) {
      bool closer = (t > 0 && t < bestHit.distance) ||
                    (t > 0 && t == bestHit.distance && best_i >= 0 && (int)i < best_i);
      if (closer) {
        best_i = (int)i;
        bestHit.distance = t;
        bestHit.position = madd(t, ray.direction, ray.origin);
        const float* N = S.normals;
        v3 n0 = ld3(N + 3 * I[i]), n1 = ld3(N + 3 * I[i + 1]), n2 = ld3(N + 3 * I[i + 2]);  // RS:259-261 (object space, A.4)
        float w = 1.0f - u - v;
        bestHit.normal = normalize((n0 * w) + (n1 * u) + (n2 * v));                        // RS:263
        bestHit.lighting = ldparams(mo.lighting);
        bestHit.kind = 3;
      }
    }
  }

  // RS:237-268
  void IntersectMeshObject(const Ray& ray, RayHit& bestHit, int mesh_index) {
    const urt_MeshObject& mo = S.mesh_objects[mesh_index];
    uint32_t offset = (uint32_t)mo.indices_offset;
    uint32_t count = offset + (uint32_t)mo.indices_count;
    int best_i = -1;   // no hit from this call yet => strict '<' only, as the literal loop
    if (mode == 0) {
      for (uint32_t i = offset; i < count; i += 3) TestTriangle(ray, bestHit, mo, i, best_i);
    } else {
      TraverseBlas(ray, bestHit, mo, S.blas_mesh_root[mesh_index], best_i);
    }
  }

  // mode 1: enumerate candidate triangles through the BVH.  Node = 16 floats:
  // c0.min c0.max c1.min c1.max (12 floats), child0, child1 (int bits), 2 pad.  child >= 0 interior
  // node index; child < 0 leaf, code = ~child, first leaf-order slot = code >> 3, count = (code & 7) + 1.
  // Root 0x7fffffff = empty mesh.  The slab test is the product's (include/urt_math.h "Slab test of the triangle BVH ..."): boxes as
  // (centre, half extent), widened per ray by pad = 2^-16 * max|origin| (on top of the pad baked in at build time), culled
  // against [0, best t] inclusive; both children hit => nearer first (ties: child0).
  void TraverseBlas(const Ray& ray, RayHit& bestHit, const urt_MeshObject& mo, int32_t root, int& best_i) {
    if (root == 0x7fffffff) return;
    v3 o = ray.origin, d = ray.direction;
    const CRay R = cray(o, d);
    int32_t stack[128];
    int sp = 0;
    int32_t cur = root;
    for (;;) {
      if (cur >= 0) {
        C.blas_nodes++;
        const float* n = S.blas_nodes + 16 * (size_t)cur;
        float tb = bestHit.distance;
        float tn[2]; bool h[2];
        for (int c = 0; c < 2; c++) {
          // the product's traversal reads (centre, half extent) boxes derived from these [lo, hi] boxes: the same derivation, the same slab arithmetic
          float cc[3], hh[3], tf;
          box_center_form(n + 6 * c, n + 6 * c + 3, cc, hh);
          cslab(cc[0], cc[1], cc[2], hh[0], hh[1], hh[2], R, tb, tn[c], tf);
          h[c] = tn[c] <= tf;
        }
        int32_t c0 = (int32_t)f_bits(n[12]), c1 = (int32_t)f_bits(n[13]);
        if (h[0] && h[1]) {
          if (tn[1] < tn[0]) { stack[sp++] = c0; cur = c1; } else { stack[sp++] = c1; cur = c0; }
        } else if (h[0]) cur = c0;
        else if (h[1]) cur = c1;
        else { if (sp == 0) break; cur = stack[--sp]; }
      } else {
        uint32_t code = ~(uint32_t)cur;
        uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
        for (uint32_t k = 0; k < cnt; k++) TestTriangle(ray, bestHit, mo, (uint32_t)S.blas_tri_index[first + k], best_i);
        if (sp == 0) break;
        cur = stack[--sp];
      }
    }
  }

  // RS:271-291.  Normative evaluation: one reciprocal per axis, (bound - origin) * rcp — see
  // DESIGN.md "normative arithmetic" (D3D `div` is itself specified to 1 ulp).
  // g_literal_division (a test-only switch, oracle_set_literal_division): evaluate RS:282-283 with the two divisions
  // written there instead — tests/test_oracle_variants.py counts the pixels that choice moves.
  static bool IntersectBVHNode(const Ray& ray, const urt_BVHNode& node) { float a, b; return IntersectBVHNode(ray, node, a, b); }
  // (t_min / t_max as compared at RS:290 are handed out for the product's object-level cull; 0, 0 for an empty node)
  static bool IntersectBVHNode(const Ray& ray, const urt_BVHNode& node, float& t_min, float& t_max) {
    t_min = 0.0f; t_max = 0.0f;
    if (node.vmin[0] == node.vmax[0] && node.vmin[1] == node.vmax[1] && node.vmin[2] == node.vmax[2]) return false;
    t_min = -kFLOAT_MAX;
    t_max = kFLOAT_MAX;
    const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z};
    const float d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
    for (int i = 0; i < 3; i++) {
      float t1, t2;
      if (g_literal_division) {
        t1 = (node.vmin[i] - o[i]) / (d[i] + kEPSILON);
        t2 = (node.vmax[i] - o[i]) / (d[i] + kEPSILON);
        t_min = f_max(t_min, f_min(t1, t2));
        t_max = f_min(t_max, f_max(t1, t2));
        continue;
      }
      float rcp = 1.0f / (d[i] + kEPSILON);
      t1 = (node.vmin[i] - o[i]) * rcp;
      t2 = (node.vmax[i] - o[i]) * rcp;
      t_min = f_max(t_min, f_min(t1, t2));
      t_max = f_min(t_max, f_max(t1, t2));
    }
    return t_max >= t_min;
  }

  // RS:294-326 — literal, including `tests` never being reset (A.5).  A structured-buffer read at
  // index -1 returns zeros in D3D: a zero MeshObject has indices_count 0, so nothing is tested.
  void IntersectMeshBVH(const Ray& ray, RayHit& bestHit) {
    int nodes[32];
    nodes[0] = 0;
    int check = 1;
    int tests = 0;
    const float t_ground = bestHit.distance;                   // Trace() calls this right after IntersectGroundPlane (RS:369-376)
    while (check > 0) {
      check--;
      int BVHIndex = nodes[check];
      urt_BVHNode node;
      if (BVHIndex >= 0 && BVHIndex < S.n_mesh_bvh) node = S.mesh_bvh[BVHIndex];
      else std::memset(&node, 0, sizeof node);                 // out-of-range read = zeros
      C.tlas_nodes++;
      float t_min, t_max;
      bool passed = IntersectBVHNode(ray, node, t_min, t_max);
      // BVH-culled mode only: the product's object-level cull (urt_math.h tlas_cull) — an object whose verified leaf box the ray passes,
      // or that lies behind the origin or beyond the ground-plane hit, by a margin, cannot hold the closest hit and is skipped.  The
      // literal mode 0 intersects it as the reference does; tests compare the two.
      bool culled = mode == 1 && S.mesh_cull_ok && node.index >= 0 && node.index < S.n_mesh_objects && S.mesh_cull_ok[node.index] &&
                    tlas_cull(t_min, t_max, t_ground);
      if (culled) { if (passed && node.index >= 0) tests++; continue; }
      if (passed) {
        if (node.index < 0) {
          nodes[check++] = BVHIndex * 2 + 1;
          nodes[check++] = BVHIndex * 2 + 2;
        } else {
          tests++;
        }
      }
      for (int i = 0; i < tests; i++) {
        OracleCounters save = C;
        if (node.index >= 0 && node.index < S.n_mesh_objects) IntersectMeshObject(ray, bestHit, node.index);
        if (i > 0) C = save;   // a repeat changes nothing (strict '<', A.5): count the work once, as the product does it once
      }
    }
  }

  // RS:329-361 — same shape.  An index of -1 reads a zero Sphere in D3D (radius 0 at the world origin),
  // which can only be hit by a ray through the exact origin; SURVEY.md A.5 fixes the net semantics as
  // "intersect object idx when idx >= 0 and (aabbHit or leafSeenBefore)", which is what this does.
  void IntersectSphereBVH(const Ray& ray, RayHit& bestHit) {
    int nodes[32];
    nodes[0] = 0;
    int check = 1;
    int tests = 0;
    while (check > 0) {
      check--;
      int BVHIndex = nodes[check];
      urt_BVHNode node;
      if (BVHIndex >= 0 && BVHIndex < S.n_sphere_bvh) node = S.sphere_bvh[BVHIndex];
      else std::memset(&node, 0, sizeof node);
      C.tlas_nodes++;
      if (IntersectBVHNode(ray, node)) {
        if (node.index < 0) {
          nodes[check++] = BVHIndex * 2 + 1;
          nodes[check++] = BVHIndex * 2 + 2;
        } else {
          tests++;
        }
      }
      for (int i = 0; i < tests; i++) {
        OracleCounters save = C;
        if (node.index >= 0 && node.index < S.n_spheres) IntersectSphere(ray, bestHit, S.spheres[node.index]);
        if (i > 0) C = save;
      }
    }
  }

  // RS:364-383
  RayHit Trace(const Ray& ray) {
    C.rays++;
    uint64_t steps0 = C.blas_nodes + C.tri_tests;
    RayHit bestHit = CreateRayHit();
    IntersectGroundPlane(ray, bestHit);
    if (S.n_mesh_objects > 0) IntersectMeshBVH(ray, bestHit);
    if (S.n_spheres > 0) IntersectSphereBVH(ray, bestHit);
    uint64_t steps = C.blas_nodes + C.tri_tests - steps0;
    if (steps > C.max_ray_steps) C.max_ray_steps = steps;
    if (steps > 256) C.rays_over_256_steps++;
    return bestHit;
  }

  // _SkyboxTexture.SampleLevel(sampler, uv, 0): bilinear, repeat, mip 0 (A.11).  Normative float
  // bilinear: texel centres at (i + 0.5)/N, weights = frac, lerp(a,b,w) = fma(w, b - a, a).
  v3 SampleSky(float u, float v) const {
    int W = S.sky_w, H = S.sky_h;
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float x0f = f_floor(x), y0f = f_floor(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    x0 %= W; if (x0 < 0) x0 += W;
    y0 %= H; if (y0 < 0) y0 += H;
    int x1 = x0 + 1; if (x1 == W) x1 = 0;
    int y1 = y0 + 1; if (y1 == H) y1 = 0;
    const float* t = S.sky;
    const float* c00 = t + 4 * ((size_t)y0 * W + x0);
    const float* c10 = t + 4 * ((size_t)y0 * W + x1);
    const float* c01 = t + 4 * ((size_t)y1 * W + x0);
    const float* c11 = t + 4 * ((size_t)y1 * W + x1);
    float r[3];
    for (int k = 0; k < 3; k++) {
      float a = f_fma(fx, c10[k] - c00[k], c00[k]);
      float b = f_fma(fx, c11[k] - c01[k], c01[k]);
      r[k] = f_fma(fy, b - a, a);
    }
    return mk3(r[0], r[1], r[2]);
  }

  // RS:386-428
  v3 Shade(Ray& ray, RayHit hit) {
    if (hit.distance < URT_INF) {
      if (hit.kind == 1) C.hit_ground++; else if (hit.kind == 2) C.hit_sphere++; else C.hit_tri++;
      hit.lighting.color_albedo = vmin3(mk3(1.0f, 1.0f, 1.0f) - hit.lighting.color_specular, hit.lighting.color_albedo);
      const float third = 1.0f / 3.0f;
      float specChance = dot(hit.lighting.color_specular, mk3(third, third, third));
      float diffChance = dot(hit.lighting.color_albedo, mk3(third, third, third));
      float sum = specChance + diffChance;
      specChance /= sum;
      diffChance /= sum;
      float roulette = rand();
      if (roulette < specChance) {
        float alpha = f_pow(1000.0f, hit.lighting.smoothness * hit.lighting.smoothness);
        ray.origin = madd(0.001f, hit.normal, hit.position);
        ray.direction = SampleHemisphere(reflect(ray.direction, hit.normal), alpha);
        float f = (alpha + 2) / (alpha + 1);
        ray.energy = ray.energy * (((1.0f / specChance) * hit.lighting.color_specular) * sdot(hit.normal, ray.direction, f));
      } else if (diffChance > 0 && roulette < specChance + diffChance) {
        ray.origin = madd(0.001f, hit.normal, hit.position);
        ray.direction = SampleHemisphere(hit.normal, 1.0f);
        ray.energy = ray.energy * ((1.0f / diffChance) * hit.lighting.color_albedo);
      } else {
        ray.energy = mk3(0, 0, 0);
      }
      return hit.lighting.emission;
    } else {
      C.hit_sky++;
      ray.energy = mk3(0, 0, 0);
      float theta = f_acos(ray.direction.y) / -kPI;
      float phi = f_atan2(ray.direction.x, -ray.direction.z) / -kPI * 0.5f;
      return SampleSky(phi, theta);
    }
  }

  // RS:431-469 for one thread id.xy
  void CSMain(uint32_t idx, uint32_t idy, float* out4) {
    _PixelX = (float)idx; _PixelY = (float)idy;   // RS:434
    _Seed = S.seed;                               // per-thread copy of the uniform (A.1)
    uint32_t width = (uint32_t)S.width, height = (uint32_t)S.height;
    v3 resultAverage = mk3(0, 0, 0);
    for (int i = 0; i < S.num_rays; i++) {
      v3 result = mk3(0, 0, 0);
      float r0 = rand();
      float r1 = rand();
      float u = ((float)idx + r0 + S.pixel_offset[0]) / (float)width * 2.0f - 1.0f;
      float v = ((float)idy + r1 + S.pixel_offset[1]) / (float)height * 2.0f - 1.0f;
      Ray ray = CreateCameraRay(u, v);
      for (int k = 0; k < S.num_bounces; k++) {
        RayHit hit = Trace(ray);
        v3 e = ray.energy;                        // read before Shade mutates it (A.3)
        v3 s = Shade(ray, hit);
        result = result + e * s;
        if (!any_nonzero(ray.energy)) break;
      }
      resultAverage = resultAverage + result;
    }
    float n = (float)S.num_rays;
    out4[0] = resultAverage.x / n; out4[1] = resultAverage.y / n; out4[2] = resultAverage.z / n; out4[3] = 1.0f;
    C.pixels++;
  }
};

static void add_counters(OracleCounters& a, const OracleCounters& b) {
  a.rays += b.rays; a.tlas_nodes += b.tlas_nodes; a.blas_nodes += b.blas_nodes; a.tri_tests += b.tri_tests;
  a.sphere_tests += b.sphere_tests; a.hit_tri += b.hit_tri; a.hit_sphere += b.hit_sphere;
  a.hit_ground += b.hit_ground; a.hit_sky += b.hit_sky; a.pixels += b.pixels;
  if (b.max_ray_steps > a.max_ray_steps) a.max_ray_steps = b.max_ray_steps;
  a.rays_over_256_steps += b.rays_over_256_steps;
}

}  // namespace

extern "C" {

// Render the pixel rectangle [x0,x1) x [y0,y1) of the W x H frame (global id.xy, row 0 = bottom) into
// out (dense (y1-y0) x (x1-x0) x 4 floats).  Rows are dealt round-robin to n_threads std::threads.
// mode: 0 brute force, 1 BLAS-culled.  counters may be NULL.  Returns 0, or 1 on bad arguments.
int oracle_render(const OracleScene* scene, int x0, int y0, int x1, int y1, int mode, int n_threads,
                  float* out, OracleCounters* counters) {
  if (!scene || !out || x0 < 0 || y0 < 0 || x1 > scene->width || y1 > scene->height || x0 > x1 || y0 > y1) return 1;
  if (mode == 1 && scene->n_mesh_objects > 0 && (!scene->blas_nodes || !scene->blas_tri_index || !scene->blas_mesh_root || scene->n_blas_nodes <= 0)) return 1;   // mode 1 walks the triangle BVH: it must be there
  if (n_threads < 1) n_threads = 1;
  std::vector<OracleCounters> cs((size_t)n_threads);
  auto work = [&](int tid) {
    Tracer T(*scene, mode);
    for (int y = y0 + tid; y < y1; y += n_threads)
      for (int x = x0; x < x1; x++)
        T.CSMain((uint32_t)x, (uint32_t)y, out + 4 * ((size_t)(y - y0) * (size_t)(x1 - x0) + (size_t)(x - x0)));
    cs[(size_t)tid] = T.C;
  };
  if (n_threads == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
    for (auto& t : th) t.join();
  }
  if (counters) {
    std::memset(counters, 0, sizeof *counters);
    for (auto& c : cs) add_counters(*counters, c);
  }
  return 0;
}

// The product's eligibility + verification rule for the object-level cull (csrc/context.cpp cull_words, csrc/cullflags.hip), restated:
// MeshObject m may be culled iff exactly one node of the mesh heap names m, that node's box is not
// empty (RS:273), m has at least one triangle, and every vertex of m's triangle records (v0, v0 + e1, v0 + e2 with e = v - v0 in
// float32, as the product stores them) lies inside that box widened by 2^-20 of its largest |coordinate|.  out: n_mesh_objects ints.
void oracle_compute_cull_ok(const OracleScene* S, int32_t* out) {
  const int nm = S->n_mesh_objects;
  for (int m = 0; m < nm; m++) out[m] = 0;
  if (nm < 1 || S->n_mesh_bvh < 1) return;
  std::vector<int> refs((size_t)nm, 0), leaf((size_t)nm, -1);
  for (int i = 0; i < S->n_mesh_bvh; i++) { int ix = S->mesh_bvh[i].index; if (ix >= 0 && ix < nm) { refs[(size_t)ix]++; leaf[(size_t)ix] = i; } }
  for (int m = 0; m < nm; m++) {
    if (refs[(size_t)m] != 1) continue;
    const urt_BVHNode& nd = S->mesh_bvh[leaf[(size_t)m]];
    if (nd.vmin[0] == nd.vmax[0] && nd.vmin[1] == nd.vmax[1] && nd.vmin[2] == nd.vmax[2]) continue;
    const urt_MeshObject& mo = S->mesh_objects[m];
    if (mo.indices_count < 3) continue;
    float M = 0.0f;
    for (int a = 0; a < 3; a++) M = f_max(M, f_max(f_abs(nd.vmin[a]), f_abs(nd.vmax[a])));
    const float tol = M * 9.5367431640625e-7f;
    bool inside = true;
    for (long i = mo.indices_offset; i + 2 < (long)mo.indices_offset + mo.indices_count && inside; i += 3) {
      v3 p[3];
      for (int k = 0; k < 3; k++) { const float* v = S->vertices + 3 * (size_t)S->indices[i + k]; p[k] = mul_m4(mo.localToWorldMatrix, v[0], v[1], v[2], 1.0f); }
      v3 q[3] = {p[0], p[0] + (p[1] - p[0]), p[0] + (p[2] - p[0])};
      for (int k = 0; k < 3; k++)
        inside = inside && q[k].x >= nd.vmin[0] - tol && q[k].x <= nd.vmax[0] + tol && q[k].y >= nd.vmin[1] - tol && q[k].y <= nd.vmax[1] + tol &&
                 q[k].z >= nd.vmin[2] - tol && q[k].z <= nd.vmax[2] + tol;
    }
    out[m] = inside ? 1 : 0;
  }
}

// AS:9,39-41 driven as RM:817-818: frag returns (T.rgb, a), a = 1/(_Sample+1); blend
// SrcAlpha OneMinusSrcAlpha on all four channels: C = src * a + C * (1 - a).
void oracle_accumulate(const float* target, float* converged, int n_pixels, float sample) {
  float a = 1.0f / (sample + 1.0f);
  float ia = 1.0f - a;
  for (int i = 0; i < n_pixels; i++) {
    const float* t = target + 4 * (size_t)i;
    float* c = converged + 4 * (size_t)i;
    c[0] = t[0] * a + c[0] * ia;
    c[1] = t[1] * a + c[1] * ia;
    c[2] = t[2] * a + c[2] * ia;
    c[3] = a * a + c[3] * ia;      // the fragment's alpha IS a
  }
}

// Single-function probes for the known-answer tests (tests/test_oracle_math.py).
// fn: 0 sin 1 cos 2 log2 3 exp2 4 pow(a,b) 5 acos 6 atan2(a,b) 7 frac 8 rand(seed=a[i], px=b[i], py=c[i])
void oracle_math_probe(int fn, const float* a, const float* b, const float* c, float* out, int n) {
  for (int i = 0; i < n; i++) {
    switch (fn) {
      case 0: out[i] = f_sin(a[i]); break;
      case 1: out[i] = f_cos(a[i]); break;
      case 2: out[i] = f_log2(a[i]); break;
      case 3: out[i] = f_exp2(a[i]); break;
      case 4: out[i] = f_pow(a[i], b[i]); break;
      case 5: out[i] = f_acos(a[i]); break;
      case 6: out[i] = f_atan2(a[i], b[i]); break;
      case 7: out[i] = f_frac(a[i]); break;
      case 8: { float s = a[i]; out[i] = rand_next(s, b[i], c[i]); } break;
      default: out[i] = 0; break;
    }
  }
}

// The triangle-BVH slab test on centre / half-extent boxes (include/urt_math.h box_center_form, cray, cslab): n boxes (lo3 hi3 each) against n
// rays (origin3 direction3 each) and best t; writes c3 h3 per box and t_near, t_far per pair (tests/test_oracle_geometry.py checks containment and
// conservativeness in exact arithmetic).
void oracle_probe_cslab(const float* boxes6, const float* rays6, const float* tbest, float* ch6, float* tnf2, int n) {
  for (int i = 0; i < n; i++) {
    float c[3], h[3];
    box_center_form(boxes6 + 6 * i, boxes6 + 6 * i + 3, c, h);
    for (int k = 0; k < 3; k++) { ch6[6 * i + k] = c[k]; ch6[6 * i + 3 + k] = h[k]; }
    const CRay R = cray(ld3(rays6 + 6 * i), ld3(rays6 + 6 * i + 3));
    cslab(c[0], c[1], c[2], h[0], h[1], h[2], R, tbest[i], tnf2[2 * i], tnf2[2 * i + 1]);
  }
}

// Exhaustive check of urt::f_div_const (include/urt_math.h) against the IEEE quotient: every float bit pattern x (both signs,
// zeros, denormals, infinities, NaNs) for the constant c.  Returns the number of x whose results differ in any bit.
unsigned long long oracle_check_div_const(float c, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  const float y = 1.0f / c;
  std::vector<unsigned long long> bad((size_t)n_threads, 0ull);
  auto work = [&](int tid) {
    unsigned long long nb = 0;
    for (uint64_t u = (uint64_t)tid; u <= 0xffffffffull; u += (uint64_t)n_threads) {
      float x = bits_f((uint32_t)u);
      float got = f_div_const(x, c, y), want = x / c;
      bool same = f_bits(got) == f_bits(want) || (got != got && want != want);   // any NaN for a NaN
      nb += same ? 0 : 1;
    }
    bad[(size_t)tid] = nb;
  };
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
  for (auto& t : th) t.join();
  unsigned long long total = 0;
  for (auto b : bad) total += b;
  return total;
}

// Geometry probes: each evaluates ONE reference function on explicit inputs.
// ray = origin(3) direction(3); returns 1 on hit and writes t,u,v.
int oracle_probe_triangle(const float* ray6, const float* v0, const float* v1, const float* v2, float* tuv) {
  Ray r = Tracer::CreateRay(ld3(ray6), ld3(ray6 + 3));
  float t = 0, u = 0, v = 0;
  bool h = Tracer::IntersectTriangle_MT97(r, ld3(v0), ld3(v1), ld3(v2), t, u, v);
  tuv[0] = t; tuv[1] = u; tuv[2] = v;
  return h ? 1 : 0;
}
int oracle_probe_aabb(const float* ray6, const urt_BVHNode* node) {
  Ray r = Tracer::CreateRay(ld3(ray6), ld3(ray6 + 3));
  return Tracer::IntersectBVHNode(r, *node) ? 1 : 0;
}
// Trace one explicit ray through the scene; out = distance, position(3), normal(3), kind.
void oracle_probe_trace(const OracleScene* scene, int mode, const float* ray6, float* out8) {
  Tracer T(*scene, mode);
  Ray r = Tracer::CreateRay(ld3(ray6), ld3(ray6 + 3));
  RayHit h = T.Trace(r);
  out8[0] = h.distance; out8[1] = h.position.x; out8[2] = h.position.y; out8[3] = h.position.z;
  out8[4] = h.normal.x; out8[5] = h.normal.y; out8[6] = h.normal.z; out8[7] = (float)h.kind;
}
// Sky lookup for a direction (the miss branch of Shade, RS:424-426).
void oracle_probe_sky(const OracleScene* scene, const float* dir3, float* rgb) {
  Tracer T(*scene, 0);
  Ray r = Tracer::CreateRay(mk3(0, 0, 0), ld3(dir3));
  RayHit h = Tracer::CreateRayHit();
  v3 c = T.Shade(r, h);
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

// An independent, deliberately simple triangle-BVH builder (median split on the longest axis of the
// centroid bounds, <= 4 triangles per leaf) in the node format above, so that the BLAS-culled mode
// can be checked without trusting the product's builder.  Returns the node count; arrays must hold
// n_nodes_cap nodes / total triangle count entries.  Boxes are padded by 2^-16 * max|coordinate|.
int oracle_build_blas(const OracleScene* scene, float* nodes, int n_nodes_cap, int32_t* tri_index, int32_t* mesh_root) {
  struct Tri { float lo[3], hi[3], c[3]; int32_t slot; };
  int n_nodes = 0, n_out = 0;
  for (int m = 0; m < scene->n_mesh_objects; m++) {
    const urt_MeshObject& mo = scene->mesh_objects[m];
    std::vector<Tri> tris;
    float ext = 0;
    for (int i = mo.indices_offset; i + 2 < mo.indices_offset + mo.indices_count; i += 3) {
      Tri t; t.slot = i;
      for (int k = 0; k < 3; k++) { t.lo[k] = URT_INF; t.hi[k] = -URT_INF; }
      for (int j = 0; j < 3; j++) {
        const float* p = scene->vertices + 3 * scene->indices[i + j];
        v3 w = mul_m4(mo.localToWorldMatrix, p[0], p[1], p[2], 1.0f);
        float wv[3] = {w.x, w.y, w.z};
        for (int k = 0; k < 3; k++) { t.lo[k] = f_min(t.lo[k], wv[k]); t.hi[k] = f_max(t.hi[k], wv[k]); ext = f_max(ext, f_abs(wv[k])); }
      }
      for (int k = 0; k < 3; k++) t.c[k] = 0.5f * (t.lo[k] + t.hi[k]);
      tris.push_back(t);
    }
    if (tris.empty()) { mesh_root[m] = 0x7fffffff; continue; }
    float pad = ext * 1.52587890625e-5f + 1e-30f;
    // recursive build with an explicit work list; returns child code
    struct Job { int lo, hi, parent, which; };
    std::vector<Job> jobs;
    jobs.push_back({0, (int)tris.size(), -1, 0});
    int base_out = n_out;
    for (size_t q = 0; q < tris.size(); q++) tri_index[n_out++] = 0;   // reserve
    while (!jobs.empty()) {
      Job j = jobs.back(); jobs.pop_back();
      int cnt = j.hi - j.lo;
      int32_t code;
      float blo[3] = {URT_INF, URT_INF, URT_INF}, bhi[3] = {-URT_INF, -URT_INF, -URT_INF};
      for (int q = j.lo; q < j.hi; q++)
        for (int k = 0; k < 3; k++) { blo[k] = f_min(blo[k], tris[q].lo[k]); bhi[k] = f_max(bhi[k], tris[q].hi[k]); }
      if (cnt <= 4) {
        for (int q = j.lo; q < j.hi; q++) tri_index[base_out + q] = tris[q].slot;
        code = (int32_t)~(uint32_t)((((uint32_t)(base_out + j.lo)) << 3) | (uint32_t)(cnt - 1));
      } else {
        float clo[3] = {URT_INF, URT_INF, URT_INF}, chi[3] = {-URT_INF, -URT_INF, -URT_INF};
        for (int q = j.lo; q < j.hi; q++)
          for (int k = 0; k < 3; k++) { clo[k] = f_min(clo[k], tris[q].c[k]); chi[k] = f_max(chi[k], tris[q].c[k]); }
        int ax = 0;
        if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
        if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
        int mid = (j.lo + j.hi) / 2;
        std::nth_element(tris.begin() + j.lo, tris.begin() + mid, tris.begin() + j.hi,
                         [ax](const Tri& a, const Tri& b) { return a.c[ax] < b.c[ax] || (a.c[ax] == b.c[ax] && a.slot < b.slot); });
        if (n_nodes >= n_nodes_cap) return -1;
        code = n_nodes++;
        float* nd = nodes + 16 * (size_t)code;
        for (int k = 0; k < 16; k++) nd[k] = 0;
        jobs.push_back({j.lo, mid, code, 0});
        jobs.push_back({mid, j.hi, code, 1});
      }
      if (j.parent < 0) mesh_root[m] = code;
      else {
        float* nd = nodes + 16 * (size_t)j.parent;
        for (int k = 0; k < 3; k++) { nd[6 * j.which + k] = blo[k] - pad; nd[6 * j.which + 3 + k] = bhi[k] + pad; }
        nd[12 + j.which] = bits_f((uint32_t)code);
      }
    }
  }
  return n_nodes;
}

// ---- literal restatements of the reference's host-side scene preparation ----------------
// UnityEngine.Vector3 arithmetic is float32 with separate multiplies and adds.
namespace {
struct U3 { float x, y, z; };
static inline U3 usub(U3 a, U3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline U3 ucross(U3 a, U3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline float usqr(U3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline U3 uld(const float* p) { return {p[0], p[1], p[2]}; }
static inline U3 umul_point_3x4(const float* m, U3 p) {       // Matrix4x4.MultiplyPoint3x4, column-major storage
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
}  // namespace

// RayTraceMaster.ComputeNormals, RM:340-368, as written: O(V * I).
void oracle_compute_normals(const float* vertices, int n_vertices, const int32_t* indices, int n_indices, float* out) {
  const float EPSILON = 3.0f * 1.401298464e-45f;                              // RM:14: float.Epsilon * 3
  for (int i = 0; i < n_vertices; i++) {
    U3 vec = {0.0f, 0.0f, 0.0f};                                              // RM:347
    U3 vi = uld(vertices + 3 * i);
    for (int listIndex = 0; listIndex < n_indices; listIndex++) {             // RM:350-351: every (vecIndex, listIndex) pair ...
      int vecIndex = indices[listIndex];
      if (usqr(usub(uld(vertices + 3 * vecIndex), vi)) <= EPSILON) {          // ... whose vertex coincides with vertex i
        int start = listIndex - (listIndex % 3);                              // RM:356
        U3 a = uld(vertices + 3 * indices[start]), b = uld(vertices + 3 * indices[start + 1]), c = uld(vertices + 3 * indices[start + 2]);
        U3 n = ucross(usub(b, a), usub(c, a));                                // RM:359
        vec = {vec.x + n.x, vec.y + n.y, vec.z + n.z};
      }
    }
    float mag = std::sqrt(vec.x * vec.x + vec.y * vec.y + vec.z * vec.z);     // Vector3.Normalize (RM:363)
    if (mag > 1e-5f) { out[3 * i] = vec.x / mag; out[3 * i + 1] = vec.y / mag; out[3 * i + 2] = vec.z / mag; }
    else { out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = 0.0f; }
  }
}

// SetupBVHLeaves(List<MeshObject>), RM:405-433, as written (including the seed from _indices[0] and the skipped first slot).
void oracle_mesh_leaf_bounds(const urt_MeshObject* meshes, int n_meshes, const float* vertices, const int32_t* indices, urt_BVHNode* out) {
  for (int m = 0; m < n_meshes; m++) {
    const urt_MeshObject& mesh = meshes[m];
    U3 lo = umul_point_3x4(mesh.localToWorldMatrix, uld(vertices + 3 * indices[0]));      // RM:415
    U3 hi = lo;                                                                            // RM:416
    for (int i = mesh.indices_offset + 1; i < mesh.indices_offset + mesh.indices_count; i++) {   // RM:421
      U3 t = umul_point_3x4(mesh.localToWorldMatrix, uld(vertices + 3 * indices[i]));
      lo = {std::min(lo.x, t.x), std::min(lo.y, t.y), std::min(lo.z, t.z)};
      hi = {std::max(hi.x, t.x), std::max(hi.y, t.y), std::max(hi.z, t.z)};
    }
    out[m].vmin[0] = lo.x; out[m].vmin[1] = lo.y; out[m].vmin[2] = lo.z;
    out[m].vmax[0] = hi.x; out[m].vmax[1] = hi.y; out[m].vmax[2] = hi.z;
    out[m].index = m;
  }
}

// SetupBVHLeaves(List<Sphere>), RM:436-455, as written (inverted boxes).
void oracle_sphere_leaf_bounds(const urt_Sphere* spheres, int n_spheres, urt_BVHNode* out) {
  for (int i = 0; i < n_spheres; i++) {
    for (int k = 0; k < 3; k++) {
      out[i].vmin[k] = spheres[i].position[k] - (-spheres[i].radius);          // RM:445
      out[i].vmax[k] = spheres[i].position[k] - spheres[i].radius;             // RM:446
    }
    out[i].index = i;
  }
}

// Test-only: 1 = the object-level slab test divides twice per axis exactly as RS:282-283 is written; 0 = the normative
// one-reciprocal form.  Process-wide; callers set it back.
void oracle_set_literal_division(int on) { g_literal_division = on ? 1 : 0; }

int oracle_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
