// urt_math.h — NORMATIVE float32 arithmetic for the path-tracing hot path.
//
// Why this file exists: the reference kernel
// (Assets/Shaders/RayTraceShader.compute, "RS") hashes `sin(x) * 43758.5453` into its random
// numbers (RS:77-81).  One ulp of difference in sin(), or one fused multiply-add the other side
// does not make, changes the random number completely, so "same pixels on CPU and GPU" is only
// definable if every intrinsic RS uses (A.10) has ONE definition with a fixed evaluation order.
// HLSL leaves that to the driver; this header pins it.  It is shared by
//   * the HIP kernels (unityraytracer_amd/csrc/*.hip, compiled by hipcc for gfx950), and
//   * host code (the BLAS builder's vertex pre-transform; the test-only oracle under oracle/),
// and uses only operations that are correctly rounded and bit-identical on x86-64 and gfx950:
//   + - * /  sqrt  fma  floor  min/max(minNum/maxNum)  integer ops and bit casts.
// No libm call, no approximate hardware op (v_rcp/v_rsq/v_sin/v_exp/v_log are NOT used).
//
// Build rule (both sides): -ffp-contract=off.  Every fused multiply-add in the normative
// semantics is spelled f_fma() here; the compiler must not add or remove one.
// (hipcc defaults to -ffp-contract=fast for device code; gcc defaults to fast as well.)
//
// Normative choices that HLSL leaves open (all documented in DESIGN.md §"normative arithmetic"):
//   dot/cross/mul(M,v)      fma chains, lowest component first (what GPU shader compilers emit)
//   normalize(v)            v * (1 / sqrt(dot(v,v)))            (normalize(0) = NaN, as A.10)
//   pow(x,y)                exp2(y * log2(x)), pow(0,y>0) = 0   (HLSL definition, A.10)
//   sin/cos                 3-term Cody–Waite reduction by pi/2 with fma + minimax polynomials
//   acos                    argument clamped to [-1,1] (normalize() can return 1+1ulp)
//   atan2(0,0)              0
//   min/max                 IEEE minNum/maxNum (a NaN operand is ignored) = v_min_f32/v_max_f32
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define URT_HD __host__ __device__ __forceinline__
#else
#define URT_HD inline __attribute__((always_inline))
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace urt {

// ---------------------------------------------------------------------------------------------
// constants (RS:12-14)
// ---------------------------------------------------------------------------------------------
static constexpr float kPI        = 3.14159265f;      // RS:12
static constexpr float kEPSILON   = 1e-8f;            // RS:13
static constexpr float kFLOAT_MAX = 3.402823466e+38f; // RS:14
#define URT_INF (__builtin_inff())

// ---------------------------------------------------------------------------------------------
// scalar primitives
// ---------------------------------------------------------------------------------------------
URT_HD float f_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
URT_HD float f_sqrt(float x) { return __builtin_sqrtf(x); }   // IEEE correctly rounded both sides
URT_HD float f_floor(float x) { return __builtin_floorf(x); }
URT_HD float f_abs(float x) { return __builtin_fabsf(x); }

URT_HD uint32_t f_bits(float x) { return __builtin_bit_cast(uint32_t, x); }
URT_HD float bits_f(uint32_t u) { return __builtin_bit_cast(float, u); }

// minNum / maxNum: if exactly one operand is NaN the other is returned.  The sign of a zero
// result is unspecified (only ever compared, never divided by).
URT_HD float f_min(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fminf(a, b);
#else
  return a < b ? a : (b != b ? a : b);
#endif
}
URT_HD float f_max(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fmaxf(a, b);
#else
  return a > b ? a : (b != b ? a : b);
#endif
}
URT_HD float f_saturate(float x) { return f_min(f_max(x, 0.0f), 1.0f); }  // RS:85 (NaN -> 0)
URT_HD float f_frac(float x) { return x - f_floor(x); }                    // RS:78

// round-to-nearest-even to an integer-valued float, valid for |x| < 2^22
URT_HD float f_rint_small(float x) { return (x + 12582912.0f) - 12582912.0f; }

// ---------------------------------------------------------------------------------------------
// sin / cos   (call sites RS:78, RS:107).  Valid for |x| < ~1e5 (rand() reaches ~1.5e4, A.1).
// ---------------------------------------------------------------------------------------------
URT_HD void f_sincos(float x, float& s_out, float& c_out) {
  // k = nearest integer to x * 2/pi ; r = x - k*pi/2 in three exact-product steps
  float k = f_rint_small(x * 0.636619747f);
  float r = f_fma(-k, 1.57079601e+00f, x);   // 0x1.921fb0p+0
  r = f_fma(-k, 3.13916473e-07f, r);         // 0x1.5110b4p-22
  r = f_fma(-k, 5.39030253e-15f, r);         // 0x1.846988p-48
  float z = r * r;
  // sin(r) on [-pi/4, pi/4]
  float p = 2.86567956e-6f;
  p = f_fma(p, z, -1.98559923e-4f);
  p = f_fma(p, z, 8.33338592e-3f);
  p = f_fma(p, z, -1.66666672e-1f);
  float sr = f_fma(p, r * z, r);
  // cos(r) on [-pi/4, pi/4]
  float q = 2.44677067e-5f;
  q = f_fma(q, z, -1.38877297e-3f);
  q = f_fma(q, z, 4.16666567e-2f);
  q = f_fma(q, z, -0.5f);
  float cr = f_fma(q, z, 1.0f);
  int i = (int)k;
  float s = (i & 1) ? cr : sr;
  float c = (i & 1) ? sr : cr;
  s_out = (i & 2) ? -s : s;
  c_out = ((i + 1) & 2) ? -c : c;
}
URT_HD float f_sin(float x) { float s, c; f_sincos(x, s, c); return s; }
URT_HD float f_cos(float x) { float s, c; f_sincos(x, s, c); return c; }

// ---------------------------------------------------------------------------------------------
// log2 / exp2 / pow   (call sites RS:104, RS:401)
// ---------------------------------------------------------------------------------------------
URT_HD float f_log2(float x) {
  if (!(x > 0.0f)) return (x == 0.0f) ? -URT_INF : __builtin_nanf("");
  if (x == URT_INF) return x;
  int e = 0;
  if (x < 1.17549435e-38f) { x = x * 16777216.0f; e = -24; }
  uint32_t u = f_bits(x);
  e += (int)(u >> 23) - 127;
  u = (u & 0x007fffffu) | 0x3f800000u;          // m in [1,2)
  float m = bits_f(u);
  if (m > 1.41421354f) { m = m * 0.5f; e += 1; } // m in (0.7071, 1.4142]
  float f = m - 1.0f;
  float s = f / (2.0f + f);
  float z = s * s;
  // ln(m) = 2 atanh(s) = 2s + s*z*(2/3 + z*(2/5 + z*(2/7 + z*2/9)))
  float p = 0.222222222f;
  p = f_fma(p, z, 0.285714286f);
  p = f_fma(p, z, 0.4f);
  p = f_fma(p, z, 0.666666667f);
  float lnm = f_fma(s * z, p, s + s);
  return f_fma(lnm, 1.44269504f, (float)e);
}

URT_HD float f_exp2(float x) {
  if (x != x) return x;
  if (x > 128.0f) return URT_INF;
  if (x < -150.0f) return 0.0f;
  float n = f_rint_small(x);
  float f = x - n;                                // [-0.5, 0.5], exact
  // 2^f, degree-7 polynomial (Taylor in f*ln2; truncation < 6e-9 on the interval)
  float p = 1.52527338e-5f;
  p = f_fma(p, f, 1.54035304e-4f);
  p = f_fma(p, f, 1.33335581e-3f);
  p = f_fma(p, f, 9.61812911e-3f);
  p = f_fma(p, f, 5.55041087e-2f);
  p = f_fma(p, f, 2.40226507e-1f);
  p = f_fma(p, f, 6.93147181e-1f);
  p = f_fma(p, f, 1.0f);
  int ni = (int)n;
  int n1 = ni >> 1;
  int n2 = ni - n1;                               // both in [-75, 64]: normal scale factors
  float s1 = bits_f((uint32_t)(n1 + 127) << 23);
  float s2 = bits_f((uint32_t)(n2 + 127) << 23);
  return (p * s1) * s2;
}

// HLSL pow(x, y) = exp2(y * log2(x)); pow(0, y>0) = 0 (A.10)
URT_HD float f_pow(float x, float y) { return f_exp2(y * f_log2(x)); }

// ---------------------------------------------------------------------------------------------
// acos / atan2   (call sites RS:424-425)
// ---------------------------------------------------------------------------------------------
URT_HD float f_asin_poly(float x, float z) {   // asin(x) for |x| <= 0.5, z = x*x
  float p = 4.2163199048e-2f;
  p = f_fma(p, z, 2.4181311049e-2f);
  p = f_fma(p, z, 4.5470025998e-2f);
  p = f_fma(p, z, 7.4953002686e-2f);
  p = f_fma(p, z, 1.6666752422e-1f);
  return f_fma(p * z, x, x);
}
URT_HD float f_acos(float x) {
  x = f_min(f_max(x, -1.0f), 1.0f);
  float a = f_abs(x);
  if (a <= 0.5f) return 1.57079637f - f_asin_poly(x, x * x);
  float z = (1.0f - a) * 0.5f;
  float s = f_sqrt(z);
  float r = f_asin_poly(s, z);
  r = r + r;
  return (x > 0.0f) ? r : (3.14159274f - r);
}

URT_HD float f_atan2(float y, float x) {
  float ax = f_abs(x), ay = f_abs(y);
  float mx = f_max(ax, ay), mn = f_min(ax, ay);
  float r;
  if (mx == 0.0f) {
    r = 0.0f;
  } else {
    float a = mn / mx;                           // [0, 1]
    float off = 0.0f;
    if (a > 0.414213562f) { a = (a - 1.0f) / (a + 1.0f); off = 0.785398163f; }
    float z = a * a;
    float p = 8.05374449538e-2f;
    p = f_fma(p, z, -1.38776856032e-1f);
    p = f_fma(p, z, 1.99777106478e-1f);
    p = f_fma(p, z, -3.33329491539e-1f);
    r = f_fma(p * z, a, a) + off;
    if (ay > ax) r = 1.57079637f - r;
  }
  if (f_bits(x) >> 31) r = 3.14159274f - r;       // x negative (incl. -0)
  return (f_bits(y) >> 31) ? -r : r;
}

// ---------------------------------------------------------------------------------------------
// vectors
// ---------------------------------------------------------------------------------------------
struct v2 { float x, y; };
struct v3 { float x, y, z; };

URT_HD v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
URT_HD v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
URT_HD v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
URT_HD v3 operator*(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
URT_HD v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
URT_HD v3 operator*(float s, v3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
URT_HD v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }

URT_HD float dot2(float ax, float ay, float bx, float by) { return f_fma(ay, by, ax * bx); }
URT_HD float dot(v3 a, v3 b) { return f_fma(a.z, b.z, f_fma(a.y, b.y, a.x * b.x)); }
URT_HD v3 cross(v3 a, v3 b) {
  return mk3(f_fma(a.y, b.z, -(a.z * b.y)),
             f_fma(a.z, b.x, -(a.x * b.z)),
             f_fma(a.x, b.y, -(a.y * b.x)));
}
URT_HD v3 normalize(v3 a) { float inv = 1.0f / f_sqrt(dot(a, a)); return a * inv; }
// o + t*d  (RS:163,191,253: "ray.origin + t * ray.direction")
URT_HD v3 madd(float t, v3 d, v3 o) { return mk3(f_fma(t, d.x, o.x), f_fma(t, d.y, o.y), f_fma(t, d.z, o.z)); }
// reflect(i, n) = i - 2 n dot(i,n)   (RS:403, A.10)
URT_HD v3 reflect(v3 i, v3 n) { float k = 2.0f * dot(i, n); return mk3(f_fma(-k, n.x, i.x), f_fma(-k, n.y, i.y), f_fma(-k, n.z, i.z)); }
URT_HD v3 vmin3(v3 a, v3 b) { return mk3(f_min(a.x, b.x), f_min(a.y, b.y), f_min(a.z, b.z)); }
URT_HD bool any_nonzero(v3 a) { return (a.x != 0.0f) || (a.y != 0.0f) || (a.z != 0.0f); }  // HLSL any(), RS:457

// mul(M, float4(v, w)).xyz with M = 16 floats in Unity Matrix4x4 memory order (column-major:
// m[col*4+row], A.9).  Each row is a 4-term fma chain.
URT_HD v3 mul_m4(const float* m, float x, float y, float z, float w) {
  v3 r;
  r.x = f_fma(m[12], w, f_fma(m[8], z, f_fma(m[4], y, m[0] * x)));
  r.y = f_fma(m[13], w, f_fma(m[9], z, f_fma(m[5], y, m[1] * x)));
  r.z = f_fma(m[14], w, f_fma(m[10], z, f_fma(m[6], y, m[2] * x)));
  return r;
}

// 1/d for the triangle-BVH slab test (not a reference function: the reference has no triangle BVH).
// |d| < 1e-18 (in particular d == +-0: a diffuse bounce whose rand() returned exactly 0 leaves exactly
// tangent to the surface) is treated as +-1e-18, so the fma form t = b * (1/d) - o * (1/d) never becomes
// inf - inf = NaN, which would switch that axis off and walk a whole slab of the mesh.
URT_HD float blas_rcp(float d) {
  float r = 1.0f / d;
  if (f_abs(d) < 1e-18f) r = (f_bits(d) >> 31) ? -1e18f : 1e18f;
  return r;
}

// ---------------------------------------------------------------------------------------------
// Slab test of the triangle BVH on CENTRE / HALF-EXTENT boxes (not a reference function: the reference has no triangle BVH).
// The builders and the refit produce [lo, hi] child boxes; the traversal reads a derived copy in which every box is (c, h) with
// [c - h, c + h] containing [lo, hi] (box_center_form: h rounded up).  Per axis
//     t_mid = c / d - o / d = fma(c, idir, b),   dh = (h + pad) / |d| = fma(h, |idir|, pa),   t_near = t_mid - dh,  t_far = t_mid + dh
// so the near / far planes need no per-axis min / max (half-rate instructions on gfx950, profiles/r03_logs/r3_valu_table_microbench.log):
// twelve add / sub instead of twelve min / max per node step, same [0, best t] clamp.  pad = 2^-16 max|origin| widens every box per ray on top
// of the build-time pad (2^-16 of the mesh extent), which together cover the rounding of these few operations (2^-22 (|o| + |c| + h) / |d|)
// and of the Moller-Trumbore test by a factor of 64 — as the [lo, hi] form did.  Shared by the kernels and the oracle's culled mode.
// ---------------------------------------------------------------------------------------------
struct CRay { v3 idir, b, pa; };        // 1/d (blas_rcp), -(o / d), pad / |d|
URT_HD CRay cray(v3 o, v3 d) {
  CRay R;
  float pad = f_max(f_max(f_abs(o.x), f_abs(o.y)), f_abs(o.z)) * 1.52587890625e-5f;
  R.idir = mk3(blas_rcp(d.x), blas_rcp(d.y), blas_rcp(d.z));
  R.b = mk3(-(o.x * R.idir.x), -(o.y * R.idir.y), -(o.z * R.idir.z));
  R.pa = mk3(pad * f_abs(R.idir.x), pad * f_abs(R.idir.y), pad * f_abs(R.idir.z));
  return R;
}
// [lo, hi] -> (c, h); an inverted (empty) box becomes one no ray enters
URT_HD void box_center_form(const float lo[3], const float hi[3], float c[3], float h[3]) {
  bool empty = false;
  for (int a = 0; a < 3; a++) empty = empty || !(lo[a] <= hi[a]);
  for (int a = 0; a < 3; a++) {
    float m = 0.5f * lo[a] + 0.5f * hi[a];
    float r = f_max(m - lo[a], hi[a] - m) * 1.0000005f + 1e-37f;
    c[a] = empty ? 0.0f : m;
    h[a] = empty ? -3.0e38f : r;
  }
}
// t_near (clamped to 0) and t_far (clamped to tbest) of one box; the box is entered iff tn <= tf
URT_HD void cslab(float cx, float cy, float cz, float hx, float hy, float hz, const CRay& R, float tbest, float& tn, float& tf) {
  float mx = f_fma(cx, R.idir.x, R.b.x), my = f_fma(cy, R.idir.y, R.b.y), mz = f_fma(cz, R.idir.z, R.b.z);
  float dx = f_fma(hx, f_abs(R.idir.x), R.pa.x), dy = f_fma(hy, f_abs(R.idir.y), R.pa.y), dz = f_fma(hz, f_abs(R.idir.z), R.pa.z);
  tn = f_max(f_max(f_max(mx - dx, my - dy), mz - dz), 0.0f);
  tf = f_min(f_min(f_min(mx + dx, my + dy), mz + dz), tbest);
}

// ---------------------------------------------------------------------------------------------
// Object-level culling (not a reference function).  The reference intersects EVERY MeshObject whose heap leaf is popped once a first
// leaf box was hit (`tests` is never reset, RS:296-326, A.5) — including objects whose box the ray misses, that lie behind its origin
// or beyond the ground-plane hit (the slab test RS:271-291 has no t-range check).  A triangle of such an object cannot be the closest
// hit, so the product skips the object when the reference's OWN slab values say so BY A MARGIN: t_min / t_max are the values RS:287-288
// computes for the leaf's box, t_ground the hit distance after IntersectGroundPlane (+inf: none).  kappa = 1/64 of the t-values'
// magnitude covers the rounding of the slab test and of Moller-Trumbore and the 2^-20 relative slack with which the library verifies
// that the leaf's box really contains the object's triangles (csrc/cullflags.hip: only such leaves are ever culled); NaN / inf compare
// false (no cull).  Counted, not assumed: tests compare the culled oracle with the literal brute force (tests/test_oracle_variants.py).
// Shared by the kernels and the oracle's BVH-culled mode, so traversal counters stay equal event for event.
// ---------------------------------------------------------------------------------------------
URT_HD bool tlas_cull(float t_min, float t_max, float t_ground) {
  const float kappa = 0.015625f;
  float s = kappa * (f_abs(t_min) + f_abs(t_max));
  bool miss = (t_min - t_max) > s;                                       // the ray's line passes the box
  bool behind = t_max < -s;                                              // the box lies behind the origin
  bool beyond = (t_min - t_ground) > kappa * (f_abs(t_min) + f_abs(t_ground));   // the box starts beyond the ground-plane hit
  return miss || behind || beyond;
}

// ---------------------------------------------------------------------------------------------
// rand()  (RS:77-81, A.1).  State: pixel (float2 of absolute pixel coordinates) + running seed.
// ---------------------------------------------------------------------------------------------
// x / c for a CONSTANT c whose correctly rounded reciprocal is y, without the divider: q = RN(x y), r = x - q c (exact, one
// fma), q' = RN(q + r y) (Markstein's correction step).  tests/test_oracle_math.py checks q' against the IEEE quotient for
// EVERY float x, for the constants it is used with (17 and 100 below, -PI in the HIP sky lookup): c = 17 agrees everywhere, c = 100
// everywhere above 4.8e-38, c = -PI above 3.1e-32; zeros (sign), the bottom of the range (|x| < 1e-30), infinities and NaN take the divider.  Ten instructions fewer per division on the GPU, same value.
URT_HD float f_div_const(float x, float c, float y) {
  float ax = f_abs(x);
  if (!(ax >= 1e-30f && ax <= 3.0e38f)) return x / c;
  float q = x * y;
  float r = f_fma(-q, c, x);
  return f_fma(r, y, q);
}

URT_HD float rand_next(float& seed, float px, float py) {
  float a = f_div_const(seed + f_div_const(seed, 17.0f, 1.0f / 17.0f), 100.0f, 1.0f / 100.0f);   // (seed + seed / 17) / 100, RS:78
  float d = dot2(px, py, 12.9898f, 78.233f);
  float r = f_frac(f_sin(a * d) * 43758.5453f);
  seed = seed + 0.5f;
  return r;
}

}  // namespace urt
