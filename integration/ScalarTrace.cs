// ScalarTrace.cs — scalar C# re-execution of the trace loop of Assets/Shaders/RayTraceShader.compute ("RS") and of the
// AdditionShader blend ("AS"), for the host cores of the machine the GPU library runs on (BASELINE.json north_star: "a scalar
// C# re-execution of the identical trace loop on the box's own host cores (core count stated) is timed in the same run as the
// reported, non-target baseline").
//
// It is the C# twin of oracle/oracle.cpp (mode 0): the same functions in the same order on the same NORMATIVE float32
// arithmetic as include/urt_math.h — sin/cos/pow/acos/atan2 are restated here with the same constants and fma placement, so
// that on a runtime with strict float32 and a real fused multiply-add (.NET Core 3.0+ on x64: MathF.FusedMultiplyAdd) the
// pixels are bit-identical to the oracle and to the HIP kernels.  Like the reference it tests EVERY triangle of a mesh whose
// object-level box is reached (RS:243) — that is the algorithm being baselined, not this library's BVH.
//
// SOURCE ONLY: the build image of this repository has no C#/.NET/Mono toolchain, so this file has not been compiled or run
// there; the timed CPU baseline bench.py reports is the C++ twin (oracle/, "kind": "port").  Usage inside the Unity project:
//     var t = new ScalarTrace.Scene { ... lists straight from RayTraceMaster (_spheres, _meshObjects, _vertices, ...) ... };
//     float[] rgba = ScalarTrace.Render(t, width, height, out long rays, out double seconds);   // all cores, rows in parallel
using System;
using System.Diagnostics;
using System.Threading.Tasks;

public static class ScalarTrace {
    // ---- scene: the buffers RM uploads (RM:738-745) and the uniforms of SetShaderParameters (RM:772-795) ------------------
    public struct Params { public float ar, ag, ab, sr, sg, sb, er, eg, eb, smoothness; }         // RayTraceParams, 40 B (RM:48-53)
    public struct MeshObject { public float[] localToWorld; public int indicesOffset, indicesCount; public Params lighting; }   // RM:82-86 (matrix column-major)
    public struct Sphere { public float px, py, pz, radius; public Params lighting; }             // RM:116-119
    public struct BVHNode { public float minx, miny, minz, maxx, maxy, maxz; public int index; }  // RM:148-152
    public sealed class Scene {
        public MeshObject[] meshObjects = new MeshObject[0];
        public float[] vertices = new float[0], normals = new float[0];                            // xyz triples
        public int[] indices = new int[0];
        public Sphere[] spheres = new Sphere[0];
        public BVHNode[] meshBVH = new BVHNode[0], sphereBVH = new BVHNode[0];
        public float[] sky = new float[4]; public int skyW = 1, skyH = 1;                          // RGBA32F, row 0 = bottom, bilinear + repeat
        public float[] cameraToWorld = new float[16], cameraInverseProjection = new float[16];    // Matrix4x4 memory order
        public float pixelOffsetX = 0.5f, pixelOffsetY = 0.5f, seed = 0.5f;
        public int numBounces = 8, numRays = 1;
    }

    // ---- normative arithmetic (include/urt_math.h) ----------------------------------------------------------------------------
    const float PI = 3.14159265f, EPSILON = 1e-8f, FLOAT_MAX = 3.402823466e+38f;                  // RS:12-14
    static float Fma(float a, float b, float c) {
#if NETCOREAPP3_0_OR_GREATER || NET5_0_OR_GREATER
        return MathF.FusedMultiplyAdd(a, b, c);
#else
        return (float)((double)a * b + c);   // product exact in double; the final double->float step can differ from a true fma in rare ties
#endif
    }
    static float Min(float a, float b) { return a < b ? a : (b != b ? a : b); }                     // minNum / maxNum
    static float Max(float a, float b) { return a > b ? a : (b != b ? a : b); }
    static float Saturate(float x) { return Min(Max(x, 0.0f), 1.0f); }
    static float Floor(float x) { return (float)Math.Floor(x); }
    static float Sqrt(float x) { return (float)Math.Sqrt(x); }                                     // correctly rounded for float inputs
    static float RintSmall(float x) { return (x + 12582912.0f) - 12582912.0f; }
    static int Bits(float f) { return BitConverter.ToInt32(BitConverter.GetBytes(f), 0); }
    static float FromBits(int i) { return BitConverter.ToSingle(BitConverter.GetBytes(i), 0); }

    static void SinCos(float x, out float s, out float c) {
        float k = RintSmall(x * 0.636619747f);
        float r = Fma(-k, 1.57079601e+00f, x);
        r = Fma(-k, 3.13916473e-07f, r);
        r = Fma(-k, 5.39030253e-15f, r);
        float z = r * r;
        float p = 2.86567956e-6f; p = Fma(p, z, -1.98559923e-4f); p = Fma(p, z, 8.33338592e-3f); p = Fma(p, z, -1.66666672e-1f);
        float sr = Fma(p, r * z, r);
        float q = 2.44677067e-5f; q = Fma(q, z, -1.38877297e-3f); q = Fma(q, z, 4.16666567e-2f); q = Fma(q, z, -0.5f);
        float cr = Fma(q, z, 1.0f);
        int i = (int)k;
        float s0 = (i & 1) != 0 ? cr : sr, c0 = (i & 1) != 0 ? sr : cr;
        s = (i & 2) != 0 ? -s0 : s0;
        c = ((i + 1) & 2) != 0 ? -c0 : c0;
    }
    static float Sin(float x) { float s, c; SinCos(x, out s, out c); return s; }
    static float Log2(float x) {
        if (!(x > 0.0f)) return x == 0.0f ? float.NegativeInfinity : float.NaN;
        if (float.IsPositiveInfinity(x)) return x;
        int e = 0;
        if (x < 1.17549435e-38f) { x = x * 16777216.0f; e = -24; }
        int u = Bits(x);
        e += (int)((uint)u >> 23) - 127;
        float m = FromBits((u & 0x007fffff) | 0x3f800000);
        if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
        float f = m - 1.0f, s = f / (2.0f + f), z = s * s;
        float p = 0.222222222f; p = Fma(p, z, 0.285714286f); p = Fma(p, z, 0.4f); p = Fma(p, z, 0.666666667f);
        float lnm = Fma(s * z, p, s + s);
        return Fma(lnm, 1.44269504f, (float)e);
    }
    static float Exp2(float x) {
        if (x != x) return x;
        if (x > 128.0f) return float.PositiveInfinity;
        if (x < -150.0f) return 0.0f;
        float n = RintSmall(x), f = x - n;
        float p = 1.52527338e-5f; p = Fma(p, f, 1.54035304e-4f); p = Fma(p, f, 1.33335581e-3f); p = Fma(p, f, 9.61812911e-3f);
        p = Fma(p, f, 5.55041087e-2f); p = Fma(p, f, 2.40226507e-1f); p = Fma(p, f, 6.93147181e-1f); p = Fma(p, f, 1.0f);
        int ni = (int)n, n1 = ni >> 1, n2 = ni - n1;
        return (p * FromBits((n1 + 127) << 23)) * FromBits((n2 + 127) << 23);
    }
    static float Pow(float x, float y) { return Exp2(y * Log2(x)); }                               // HLSL pow
    static float AsinPoly(float x, float z) {
        float p = 4.2163199048e-2f; p = Fma(p, z, 2.4181311049e-2f); p = Fma(p, z, 4.5470025998e-2f); p = Fma(p, z, 7.4953002686e-2f);
        p = Fma(p, z, 1.6666752422e-1f);
        return Fma(p * z, x, x);
    }
    static float Acos(float x) {
        x = Min(Max(x, -1.0f), 1.0f);
        float a = Math.Abs(x);
        if (a <= 0.5f) return 1.57079637f - AsinPoly(x, x * x);
        float z = (1.0f - a) * 0.5f, s = Sqrt(z), r = AsinPoly(s, z);
        r = r + r;
        return x > 0.0f ? r : (3.14159274f - r);
    }
    static float Atan2(float y, float x) {
        float ax = Math.Abs(x), ay = Math.Abs(y), mx = Max(ax, ay), mn = Min(ax, ay), r;
        if (mx == 0.0f) r = 0.0f;
        else {
            float a = mn / mx, off = 0.0f;
            if (a > 0.414213562f) { a = (a - 1.0f) / (a + 1.0f); off = 0.785398163f; }
            float z = a * a;
            float p = 8.05374449538e-2f; p = Fma(p, z, -1.38776856032e-1f); p = Fma(p, z, 1.99777106478e-1f); p = Fma(p, z, -3.33329491539e-1f);
            r = Fma(p * z, a, a) + off;
            if (ay > ax) r = 1.57079637f - r;
        }
        if (Bits(x) < 0) r = 3.14159274f - r;
        return Bits(y) < 0 ? -r : r;
    }

    struct V3 { public float x, y, z; public V3(float x, float y, float z) { this.x = x; this.y = y; this.z = z; } }
    static V3 Add(V3 a, V3 b) { return new V3(a.x + b.x, a.y + b.y, a.z + b.z); }
    static V3 Sub(V3 a, V3 b) { return new V3(a.x - b.x, a.y - b.y, a.z - b.z); }
    static V3 Mul(V3 a, V3 b) { return new V3(a.x * b.x, a.y * b.y, a.z * b.z); }
    static V3 Mul(V3 a, float s) { return new V3(a.x * s, a.y * s, a.z * s); }
    static float Dot(V3 a, V3 b) { return Fma(a.z, b.z, Fma(a.y, b.y, a.x * b.x)); }
    static V3 Cross(V3 a, V3 b) { return new V3(Fma(a.y, b.z, -(a.z * b.y)), Fma(a.z, b.x, -(a.x * b.z)), Fma(a.x, b.y, -(a.y * b.x))); }
    static V3 Normalize(V3 a) { float inv = 1.0f / Sqrt(Dot(a, a)); return Mul(a, inv); }
    static V3 Madd(float t, V3 d, V3 o) { return new V3(Fma(t, d.x, o.x), Fma(t, d.y, o.y), Fma(t, d.z, o.z)); }
    static V3 Reflect(V3 i, V3 n) { float k = 2.0f * Dot(i, n); return new V3(Fma(-k, n.x, i.x), Fma(-k, n.y, i.y), Fma(-k, n.z, i.z)); }
    static V3 MulM4(float[] m, float x, float y, float z, float w) {                                // mul(M, float4(v, w)).xyz, column-major storage
        return new V3(Fma(m[12], w, Fma(m[8], z, Fma(m[4], y, m[0] * x))), Fma(m[13], w, Fma(m[9], z, Fma(m[5], y, m[1] * x))),
                      Fma(m[14], w, Fma(m[10], z, Fma(m[6], y, m[2] * x))));
    }

    // ---- the shader, one instance per pixel (its mutable globals _Pixel / _Seed are fields: RS:15-16) ------------------------------
    struct Ray { public V3 origin, direction, energy; }
    struct RayHit { public V3 position, normal; public float distance; public Params lighting; }

    sealed class Tracer {
        readonly Scene S;
        float pixelX, pixelY, seed;
        public long rays;
        public Tracer(Scene s) { S = s; }

        float Rand() {                                                                              // RS:77-81
            float a = (seed + seed / 17.0f) / 100.0f;
            float d = Fma(pixelY, 78.233f, pixelX * 12.9898f);
            float v = Sin(a * d) * 43758.5453f;
            float r = v - Floor(v);
            seed = seed + 0.5f;
            return r;
        }
        V3 SampleHemisphere(V3 normal, float alpha) {                                               // RS:103-111 with GetTangentSpace RS:89-100
            float cosTheta = Pow(Rand(), 1.0f / (alpha + 1.0f));
            float sinTheta = Sqrt(1.0f - cosTheta * cosTheta);
            float phi = (2.0f * PI) * Rand();
            float sp, cp; SinCos(phi, out sp, out cp);
            V3 ts = new V3(cp * sinTheta, sp * sinTheta, cosTheta);
            V3 helper = Math.Abs(normal.x) > 0.99f ? new V3(0, 0, 1) : new V3(1, 0, 0);
            V3 tangent = Normalize(Cross(normal, helper)), binormal = Normalize(Cross(normal, tangent));
            return new V3(Fma(ts.z, normal.x, Fma(ts.y, binormal.x, ts.x * tangent.x)), Fma(ts.z, normal.y, Fma(ts.y, binormal.y, ts.x * tangent.y)),
                          Fma(ts.z, normal.z, Fma(ts.y, binormal.z, ts.x * tangent.z)));
        }
        Ray CreateCameraRay(float u, float v) {                                                     // RS:142-153
            Ray r;
            r.origin = MulM4(S.cameraToWorld, 0, 0, 0, 1);
            V3 d = MulM4(S.cameraInverseProjection, u, v, 0, 1);
            r.direction = Normalize(MulM4(S.cameraToWorld, d.x, d.y, d.z, 0));
            r.energy = new V3(1, 1, 1);
            return r;
        }
        void IntersectGroundPlane(ref Ray ray, ref RayHit best) {                                   // RS:156-172
            float t = -ray.origin.y / ray.direction.y;
            if (t > 0 && t < best.distance) {
                best.distance = t; best.position = Madd(t, ray.direction, ray.origin); best.normal = new V3(0, 1, 0);
                best.lighting = new Params { ar = 0.5f, ag = 0.3f, ab = 0.15f, smoothness = 0.3f };
            }
        }
        void IntersectSphere(ref Ray ray, ref RayHit best, ref Sphere s) {                          // RS:175-196
            V3 c = new V3(s.px, s.py, s.pz), d = Sub(ray.origin, c);
            float p1 = -Dot(ray.direction, d);
            float p2sqr = p1 * p1 - Dot(d, d) + s.radius * s.radius;
            if (p2sqr < 0) return;
            float p2 = Sqrt(p2sqr);
            float t = p1 - p2 > 0 ? p1 - p2 : p1 + p2;
            if (t > 0 && t < best.distance) {
                best.distance = t; best.position = Madd(t, ray.direction, ray.origin); best.normal = Normalize(Sub(best.position, c)); best.lighting = s.lighting;
            }
        }
        static bool IntersectTriangleMT97(ref Ray ray, V3 v0, V3 v1, V3 v2, out float t, out float u, out float v) {   // RS:199-234
            t = u = v = 0;
            V3 e1 = Sub(v1, v0), e2 = Sub(v2, v0), pvec = Cross(ray.direction, e2);
            float det = Dot(e1, pvec);
            if (det < EPSILON) return false;
            float inv = 1.0f / det;
            V3 tvec = Sub(ray.origin, v0);
            u = Dot(tvec, pvec) * inv;
            if (u < 0.0f || u > 1.0f) return false;
            V3 qvec = Cross(tvec, e1);
            v = Dot(ray.direction, qvec) * inv;
            if (v < 0.0f || u + v > 1.0f) return false;
            t = Dot(e2, qvec) * inv;
            return true;
        }
        V3 Vtx(float[] a, int i) { return new V3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
        void IntersectMeshObject(ref Ray ray, ref RayHit best, ref MeshObject mo) {                 // RS:237-268: every triangle of the mesh
            int end = mo.indicesOffset + mo.indicesCount;
            for (int i = mo.indicesOffset; i + 2 < end; i += 3) {
                int i0 = S.indices[i], i1 = S.indices[i + 1], i2 = S.indices[i + 2];
                V3 p0 = Vtx(S.vertices, i0), p1 = Vtx(S.vertices, i1), p2 = Vtx(S.vertices, i2);
                V3 v0 = MulM4(mo.localToWorld, p0.x, p0.y, p0.z, 1), v1 = MulM4(mo.localToWorld, p1.x, p1.y, p1.z, 1), v2 = MulM4(mo.localToWorld, p2.x, p2.y, p2.z, 1);
                float t, u, v;
                if (IntersectTriangleMT97(ref ray, v0, v1, v2, out t, out u, out v) && t > 0 && t < best.distance) {
                    best.distance = t; best.position = Madd(t, ray.direction, ray.origin);
                    V3 n0 = Vtx(S.normals, i0), n1 = Vtx(S.normals, i1), n2 = Vtx(S.normals, i2);  // object-space normals (RS:259-261)
                    float w = 1.0f - u - v;
                    best.normal = Normalize(Add(Add(Mul(n0, w), Mul(n1, u)), Mul(n2, v)));
                    best.lighting = mo.lighting;
                }
            }
        }
        static bool IntersectBVHNode(ref Ray ray, ref BVHNode n) {                                  // RS:271-291 (normative: one reciprocal per axis)
            if (n.minx == n.maxx && n.miny == n.maxy && n.minz == n.maxz) return false;
            float tMin = -FLOAT_MAX, tMax = FLOAT_MAX, rcp, t1, t2;
            rcp = 1.0f / (ray.direction.x + EPSILON); t1 = (n.minx - ray.origin.x) * rcp; t2 = (n.maxx - ray.origin.x) * rcp;
            tMin = Max(tMin, Min(t1, t2)); tMax = Min(tMax, Max(t1, t2));
            rcp = 1.0f / (ray.direction.y + EPSILON); t1 = (n.miny - ray.origin.y) * rcp; t2 = (n.maxy - ray.origin.y) * rcp;
            tMin = Max(tMin, Min(t1, t2)); tMax = Min(tMax, Max(t1, t2));
            rcp = 1.0f / (ray.direction.z + EPSILON); t1 = (n.minz - ray.origin.z) * rcp; t2 = (n.maxz - ray.origin.z) * rcp;
            tMin = Max(tMin, Min(t1, t2)); tMax = Min(tMax, Max(t1, t2));
            return tMax >= tMin;
        }
        // RS:294-361: explicit-stack walk of the implicit heap; `tests` is never reset (once a leaf was reached, every later popped node has
        // its object intersected); an object index of -1 / out of range is a zero-size read in D3D and intersects nothing.
        void WalkHeap(ref Ray ray, ref RayHit best, BVHNode[] heap, bool meshes) {
            int[] nodes = new int[32];
            nodes[0] = 0;
            int check = 1, tests = 0;
            while (check > 0) {
                check--;
                int bi = nodes[check];
                BVHNode node = (bi >= 0 && bi < heap.Length) ? heap[bi] : new BVHNode();
                if (IntersectBVHNode(ref ray, ref node)) {
                    if (node.index < 0) { nodes[check++] = bi * 2 + 1; nodes[check++] = bi * 2 + 2; }
                    else tests++;
                }
                if (tests > 0 && node.index >= 0) {                                                  // repeats are idempotent (strict <): once is enough
                    if (meshes) { if (node.index < S.meshObjects.Length) IntersectMeshObject(ref ray, ref best, ref S.meshObjects[node.index]); }
                    else if (node.index < S.spheres.Length) IntersectSphere(ref ray, ref best, ref S.spheres[node.index]);
                }
            }
        }
        RayHit Trace(ref Ray ray) {                                                                 // RS:364-383
            rays++;
            RayHit best = new RayHit { distance = float.PositiveInfinity };
            IntersectGroundPlane(ref ray, ref best);
            if (S.meshObjects.Length > 0) WalkHeap(ref ray, ref best, S.meshBVH, true);
            if (S.spheres.Length > 0) WalkHeap(ref ray, ref best, S.sphereBVH, false);
            return best;
        }
        V3 SampleSky(float u, float v) {                                                            // SampleLevel(sampler, uv, 0): bilinear, repeat
            int W = S.skyW, H = S.skyH;
            float x = u * W - 0.5f, y = v * H - 0.5f, x0f = Floor(x), y0f = Floor(y), fx = x - x0f, fy = y - y0f;
            int x0 = (int)x0f % W; if (x0 < 0) x0 += W;
            int y0 = (int)y0f % H; if (y0 < 0) y0 += H;
            int x1 = x0 + 1 == W ? 0 : x0 + 1, y1 = y0 + 1 == H ? 0 : y0 + 1;
            float[] r = new float[3];
            for (int k = 0; k < 3; k++) {
                float c00 = S.sky[4 * (y0 * W + x0) + k], c10 = S.sky[4 * (y0 * W + x1) + k], c01 = S.sky[4 * (y1 * W + x0) + k], c11 = S.sky[4 * (y1 * W + x1) + k];
                float a = Fma(fx, c10 - c00, c00), b = Fma(fx, c11 - c01, c01);
                r[k] = Fma(fy, b - a, a);
            }
            return new V3(r[0], r[1], r[2]);
        }
        V3 Shade(ref Ray ray, RayHit hit) {                                                         // RS:386-428
            if (hit.distance < float.PositiveInfinity) {
                Params L = hit.lighting;
                V3 spec = new V3(L.sr, L.sg, L.sb);
                V3 albedo = new V3(Min(1.0f - spec.x, L.ar), Min(1.0f - spec.y, L.ag), Min(1.0f - spec.z, L.ab));
                const float third = 1.0f / 3.0f;
                V3 thirds = new V3(third, third, third);
                float specChance = Dot(spec, thirds), diffChance = Dot(albedo, thirds), sum = specChance + diffChance;
                specChance /= sum; diffChance /= sum;
                float roulette = Rand();
                if (roulette < specChance) {
                    float alpha = Pow(1000.0f, L.smoothness * L.smoothness);
                    ray.origin = Madd(0.001f, hit.normal, hit.position);
                    ray.direction = SampleHemisphere(Reflect(ray.direction, hit.normal), alpha);
                    float f = (alpha + 2) / (alpha + 1);
                    ray.energy = Mul(ray.energy, Mul(Mul(spec, 1.0f / specChance), Saturate(Dot(hit.normal, ray.direction) * f)));
                } else if (diffChance > 0 && roulette < specChance + diffChance) {
                    ray.origin = Madd(0.001f, hit.normal, hit.position);
                    ray.direction = SampleHemisphere(hit.normal, 1.0f);
                    ray.energy = Mul(ray.energy, Mul(albedo, 1.0f / diffChance));
                } else ray.energy = new V3(0, 0, 0);
                return new V3(L.er, L.eg, L.eb);
            }
            ray.energy = new V3(0, 0, 0);
            float theta = Acos(ray.direction.y) / -PI, phi = Atan2(ray.direction.x, -ray.direction.z) / -PI * 0.5f;
            return SampleSky(phi, theta);
        }
        public void Pixel(int x, int y, int width, int height, float[] result) {                    // CSMain, RS:431-469
            pixelX = x; pixelY = y; seed = S.seed;
            V3 avg = new V3(0, 0, 0);
            for (int i = 0; i < S.numRays; i++) {
                V3 res = new V3(0, 0, 0);
                float r0 = Rand(), r1 = Rand();
                float u = (pixelX + r0 + S.pixelOffsetX) / width * 2.0f - 1.0f, v = (pixelY + r1 + S.pixelOffsetY) / height * 2.0f - 1.0f;
                Ray ray = CreateCameraRay(u, v);
                for (int k = 0; k < S.numBounces; k++) {
                    RayHit hit = Trace(ref ray);
                    V3 e0 = ray.energy;                                                             // energy is read BEFORE Shade updates it (RS:455)
                    V3 s = Shade(ref ray, hit);
                    res = Add(res, Mul(e0, s));
                    if (ray.energy.x == 0 && ray.energy.y == 0 && ray.energy.z == 0) break;         // !any(energy), RS:457
                }
                avg = Add(avg, res);
            }
            float n = S.numRays;
            int at = 4 * (y * width + x);
            result[at] = avg.x / n; result[at + 1] = avg.y / n; result[at + 2] = avg.z / n; result[at + 3] = 1.0f;
        }
    }

    /// One frame on all host cores (rows in parallel, one Tracer per row): RGBA32F, row 0 = bottom.  Reports the Trace() invocations
    /// (the unit of the Mrays/s metric, RS:454) and the wall time; print Environment.ProcessorCount next to the figure.
    public static float[] Render(Scene scene, int width, int height, out long rays, out double seconds) {
        float[] result = new float[4 * width * height];
        long total = 0;
        Stopwatch sw = Stopwatch.StartNew();
        Parallel.For(0, height, () => 0L, (y, state, local) => {
            Tracer t = new Tracer(scene);
            for (int x = 0; x < width; x++) t.Pixel(x, y, width, height, result);
            return local + t.rays;
        }, local => System.Threading.Interlocked.Add(ref total, local));
        sw.Stop();
        rays = total; seconds = sw.Elapsed.TotalSeconds;
        return result;
    }

    /// AdditionShader (AS:9,39-41) as driven by RM:817-818: converged = target * a + converged * (1 - a), a = 1 / (sample + 1), all four channels.
    public static void Accumulate(float[] target, float[] converged, float sample) {
        float a = 1.0f / (sample + 1.0f), ia = 1.0f - a;
        for (int i = 0; i < target.Length; i += 4) {
            converged[i] = target[i] * a + converged[i] * ia;
            converged[i + 1] = target[i + 1] * a + converged[i + 1] * ia;
            converged[i + 2] = target[i + 2] * a + converged[i + 2] * ia;
            converged[i + 3] = a * a + converged[i + 3] * ia;
        }
    }

    public static string Report(long rays, double seconds) {
        return string.Format("scalar C# trace loop: {0:F3} Mrays/s on {1} logical cores ({2} rays in {3:F2} s)", rays / seconds / 1e6, Environment.ProcessorCount, rays, seconds);
    }
}
