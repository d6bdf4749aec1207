/* frame_loop.c — the reference's frame protocol (RayTraceMaster.cs "RM": RebuildTrees -> SetShaderParameters -> Render, RM:725-866)
 * written against include/urt.h in plain C99: what a native host (or the P/Invoke shim of integration/UrtNative.cs) does, with no
 * Python and no C++ in between.  A small scene — four spheres over the ground plane and one quad mesh — is uploaded through the
 * ComputeBuffer calls, N frames are dispatched and accumulated, and the running mean is read back and written as PFM + PNG.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/frame_loop.c -o /tmp/frame_loop -Lunityraytracer_amd -lunityraytracer_amd \
 *       -Wl,-rpath,$PWD/unityraytracer_amd && /tmp/frame_loop 64 /tmp/frame        (needs an MI355X; there is no CPU path)
 *
 * tests/test_gpu_c_host.py builds and runs it and compares its image bit for bit with the same protocol driven from Python.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "urt.h"

#define W 160
#define H 96
#define CHECK(ctx, call)                                                                  \
  do {                                                                                    \
    int rc__ = (call);                                                                    \
    if (rc__ != URT_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, urt_last_error(ctx)); return 2; } \
  } while (0)

/* splitmix64 frame uniforms: the documented stand-in for UnityEngine.Random.value (RM:777-778), same sequence as
 * unityraytracer_amd/scenes.py frame_uniforms (three draws per frame; frame 0 is the fixed fixture 0.5, 0.5, 0.5) */
static unsigned long long sm_state;
static float sm_value(void) {
  unsigned long long z = (sm_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (float)((double)(z >> 40) / 16777216.0);
}

static void material(urt_RayTraceParams* m, float ar, float ag, float ab, float sr, float sg, float sb, float e, float smooth) {
  m->color_albedo[0] = ar; m->color_albedo[1] = ag; m->color_albedo[2] = ab;
  m->color_specular[0] = sr; m->color_specular[1] = sg; m->color_specular[2] = sb;
  m->emission[0] = m->emission[1] = m->emission[2] = e;
  m->smoothness = smooth;
}

int main(int argc, char** argv) {
  int frames = argc > 1 ? atoi(argv[1]) : 16;
  const char* out = argc > 2 ? argv[2] : "/tmp/frame";
  urt_context* ctx = NULL;
  if (urt_context_create(0, &ctx) != URT_OK) { fprintf(stderr, "urt_context_create: %s\n", urt_last_error(NULL)); return 1; }

  /* ---- the lists RebuildObjectLists produces (RM:262-336) ---- */
  urt_Sphere spheres[4];
  const float sx[4] = {-3.0f, -1.0f, 1.2f, 3.2f}, sr[4] = {0.9f, 0.6f, 0.8f, 0.5f};
  memset(spheres, 0, sizeof spheres);
  for (int k = 0; k < 4; k++) {
    spheres[k].position[0] = sx[k]; spheres[k].position[1] = sr[k]; spheres[k].position[2] = (float)(k % 2) * 1.5f;
    spheres[k].radius = sr[k];
  }
  material(&spheres[0].lighting, 0.8f, 0.2f, 0.2f, 0.04f, 0.04f, 0.04f, 0.0f, 0.2f);
  material(&spheres[1].lighting, 0.0f, 0.0f, 0.0f, 0.9f, 0.9f, 0.9f, 0.0f, 0.95f);
  material(&spheres[2].lighting, 0.2f, 0.7f, 0.3f, 0.3f, 0.3f, 0.3f, 0.0f, 0.6f);
  material(&spheres[3].lighting, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 3.0f, 0.0f);
  /* one quad (two triangles) standing behind the spheres, facing the camera (-z) */
  const float vertices[12] = {-4, 0, 4,  4, 0, 4,  4, 3, 4,  -4, 3, 4};
  const int32_t indices[6] = {0, 2, 1, 0, 3, 2};
  float normals[12];
  CHECK(ctx, urt_host_compute_normals(vertices, 4, indices, 6, normals));                 /* RM:340-368 */
  urt_MeshObject mesh;
  memset(&mesh, 0, sizeof mesh);
  mesh.localToWorldMatrix[0] = mesh.localToWorldMatrix[5] = mesh.localToWorldMatrix[10] = mesh.localToWorldMatrix[15] = 1.0f;
  mesh.indices_offset = 0; mesh.indices_count = 6;
  material(&mesh.lighting, 0.6f, 0.6f, 0.7f, 0.1f, 0.1f, 0.1f, 0.0f, 0.3f);
  /* object-level heaps (RM:405-722): leaf boxes + the heap in the reference's array contract */
  urt_BVHNode mesh_leaf[1], sphere_leaf[4], mesh_bvh[1], sphere_bvh[7];
  CHECK(ctx, urt_host_mesh_leaf_bounds(&mesh, 1, vertices, 4, indices, 6, 0, mesh_leaf));
  CHECK(ctx, urt_host_sphere_leaf_bounds(spheres, 4, 0, sphere_leaf));
  CHECK(ctx, urt_host_build_object_bvh(mesh_leaf, 1, mesh_bvh, urt_host_object_bvh_length(1)));
  CHECK(ctx, urt_host_build_object_bvh_pairing(sphere_leaf, 4, sphere_bvh, urt_host_object_bvh_length(4)));   /* the reference's own builder */

  /* ---- RebuildTrees: CreateComputeBuffer x 7 (RM:738-745) ---- */
  struct { const char* name; const void* data; int count, stride; } bufs[7] = {
      {"_MeshObjects", &mesh, 1, URT_STRIDE_MESHOBJECT}, {"_Vertices", vertices, 4, URT_STRIDE_VEC3}, {"_Indices", indices, 6, URT_STRIDE_INDEX},
      {"_Normals", normals, 4, URT_STRIDE_VEC3}, {"_Spheres", spheres, 4, URT_STRIDE_SPHERE}, {"_MeshBVH", mesh_bvh, 1, URT_STRIDE_BVHNODE},
      {"_SphereBVH", sphere_bvh, 7, URT_STRIDE_BVHNODE}};
  urt_handle handles[7];
  for (int k = 0; k < 7; k++) {
    CHECK(ctx, urt_buffer_create(ctx, bufs[k].count, bufs[k].stride, &handles[k]));
    CHECK(ctx, urt_buffer_set_data(ctx, handles[k], bufs[k].data, bufs[k].count));
  }
  /* a small analytic sky (RGBA32F, row 0 = bottom) */
  enum { SW = 64, SH = 32 };
  static float sky[SW * SH * 4];
  for (int y = 0; y < SH; y++)
    for (int x = 0; x < SW; x++) {
      float t = (float)y / (float)(SH - 1);
      float* p = sky + 4 * (y * SW + x);
      p[0] = 0.3f + 0.5f * t; p[1] = 0.4f + 0.5f * t; p[2] = 0.6f + 0.4f * t; p[3] = 1.0f;
    }
  urt_handle sky_tex, target, converged, destination;
  CHECK(ctx, urt_texture_create(ctx, SW, SH, &sky_tex));
  CHECK(ctx, urt_texture_set_pixels(ctx, sky_tex, sky));
  CHECK(ctx, urt_texture_create(ctx, W, H, &target));                                      /* InitRenderTexture, RM:824-845 */
  CHECK(ctx, urt_texture_create(ctx, W, H, &converged));
  CHECK(ctx, urt_texture_create(ctx, W, H, &destination));                                 /* OnRenderImage's `destination` (RM:848) */

  /* camera of Scene1 (position (0,1,-10), identity rotation, vertical fov 81 degrees): Unity's GL-convention matrices */
  float c2w[16] = {1, 0, 0, 0,  0, 1, 0, 0,  0, 0, -1, 0,  0, 1, -10, 1};                 /* TR * diag(1,1,-1), column-major */
  const float near_ = 0.3f, far_ = 1000.0f, th = 0.85408069f /* tan(40.5 deg) */, aspect = (float)W / (float)H;
  float invp[16];
  memset(invp, 0, sizeof invp);
  invp[0] = aspect * th; invp[5] = th; invp[11] = (near_ - far_) / (2.0f * far_ * near_); invp[14] = -1.0f; invp[15] = (far_ + near_) / (2.0f * far_ * near_);

  sm_state = 0x5EEDULL;
  for (int frame = 0; frame < frames; frame++) {
    /* ---- SetShaderParameters (RM:772-795) ---- */
    float off[4] = {0.5f, 0.5f, 0, 0}, seed = 0.5f;
    if (frame > 0) { off[0] = sm_value(); off[1] = sm_value(); seed = sm_value(); }
    CHECK(ctx, urt_shader_set_matrix(ctx, "_CameraToWorld", c2w));
    CHECK(ctx, urt_shader_set_matrix(ctx, "_CameraInverseProjection", invp));
    CHECK(ctx, urt_shader_set_texture(ctx, 0, "_SkyboxTexture", sky_tex));
    CHECK(ctx, urt_shader_set_vector(ctx, "_PixelOffset", off));
    CHECK(ctx, urt_shader_set_float(ctx, "_Seed", seed));
    CHECK(ctx, urt_shader_set_int(ctx, "_numBounces", 6));
    CHECK(ctx, urt_shader_set_int(ctx, "_numRays", 1));
    CHECK(ctx, urt_shader_set_int(ctx, "_MeshBVH_len", 1));                                /* accepted and ignored (static const, RS:73-74) */
    CHECK(ctx, urt_shader_set_int(ctx, "_SphereBVH_len", 7));
    for (int k = 0; k < 7; k++) CHECK(ctx, urt_shader_set_buffer(ctx, 0, bufs[k].name, handles[k]));
    /* ---- Render (RM:798-821) ---- */
    CHECK(ctx, urt_shader_set_texture(ctx, 0, "Result", target));
    CHECK(ctx, urt_shader_dispatch(ctx, 0, (W + 7) / 8, (H + 7) / 8, 1));
    CHECK(ctx, urt_blit_add(ctx, target, converged, (float)frame));                         /* RM:817-818: _Sample = _currentSample */
    CHECK(ctx, urt_blit(ctx, converged, destination));                                      /* RM:819: present; then _currentSample++ */
  }
  static float image[W * H * 4];
  CHECK(ctx, urt_texture_get_pixels(ctx, destination, image));                              /* submits the batched frames (all of them: the presents were queued too) and waits */
  urt_counters c;
  CHECK(ctx, urt_get_counters(ctx, &c));
  char path[1024];
  /* the same present as an 8-bit sRGB back buffer would hold it (RM:819 into the camera's target), converted on the GPU and read back
   * without stalling the stream: begin ... (a real host would submit the next frames here) ... end */
  uint64_t ticket = 0;
  const void* rgba8 = NULL;
  size_t rgba8_bytes = 0;
  CHECK(ctx, urt_texture_read_begin_format(ctx, destination, URT_FORMAT_RGBA8_SRGB, &ticket));
  CHECK(ctx, urt_texture_read_end_format(ctx, ticket, &rgba8, &rgba8_bytes));
  snprintf(path, sizeof path, "%s.rgba8", out);
  FILE* f8 = fopen(path, "wb");
  if (!f8 || rgba8_bytes != (size_t)W * H * 4 || fwrite(rgba8, rgba8_bytes, 1, f8) != 1) { fprintf(stderr, "cannot write %s\n", path); return 3; }
  fclose(f8);
  snprintf(path, sizeof path, "%s.pfm", out);
  CHECK(ctx, urt_host_write_pfm(path, image, W, H));
  snprintf(path, sizeof path, "%s.png", out);
  CHECK(ctx, urt_host_write_png(path, image, W, H));
  snprintf(path, sizeof path, "%s.rgba32f", out);
  FILE* f = fopen(path, "wb");
  if (!f || fwrite(image, sizeof image, 1, f) != 1) { fprintf(stderr, "cannot write %s\n", path); return 3; }
  fclose(f);
  printf("%d frames of %dx%d: %llu rays in %llu launches (%llu dispatches), watchdog %u\n", frames, W, H, (unsigned long long)c.rays,
         (unsigned long long)c.launches, (unsigned long long)c.dispatches, c.watchdog_trips);
  for (int k = 0; k < 7; k++) CHECK(ctx, urt_buffer_release(ctx, handles[k]));
  CHECK(ctx, urt_texture_release(ctx, target));
  CHECK(ctx, urt_texture_release(ctx, converged));
  CHECK(ctx, urt_texture_release(ctx, destination));
  CHECK(ctx, urt_texture_release(ctx, sky_tex));
  urt_context_destroy(ctx);
  return 0;
}
