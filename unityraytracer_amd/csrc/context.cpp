// context.cpp — implementation of the C ABI in include/urt.h.
//
// Stands in for the UnityEngine GPU objects RayTraceMaster.cs drives:
//   ComputeBuffer   -> Buffer   (host copy kept; the device form is DERIVED at the next dispatch)
//   RenderTexture   -> Texture  (RGBA32F device image)
//   ComputeShader   -> the uniform/binding table in urt_context + dispatch of the HIP kernels
//   Graphics.Blit   -> urt_blit / urt_blit_add
// There is deliberately no CPU path: without a HIP device context creation fails.
#include "experiments.h"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/urt.h"
#include "../../include/urt_math.h"
#include "blas_builder.h"
#include "kernels.h"
#include "lbvh.h"
#include "refit.h"
#include "qnodes.h"
#include "cullflags.h"
#include "present.h"
#include "urt_device.h"

#include <chrono>

using namespace urtd;

namespace {

struct Buffer {
  int count = 0, stride = 0;
  std::vector<uint8_t> host;   // SetData copy (RM:250): the caller keeps ownership of its list
  bool has_data = false;
};

struct Texture {
  int w = 0, h = 0;
  float4* dev = nullptr;        // where the CURRENT contents live: `own`, or a frame slot of the context's slab (a Result
                                // texture is renamed to a fresh slot by every batched dispatch)
  float4* own = nullptr;        // the allocation made at creation (or the caller's memory when external)
  bool external = false;
  bool ptr_exposed = false;     // urt_texture_get_info handed out the device pointer: never renamed again
  // What has written the image since its zero-filled creation.  A dispatch that covers only part of the image may be renamed
  // to a (zero-filled) slab slot only while the pixels outside its region are still the zeros of creation, i.e. while
  // nothing but dispatches of that SAME region has written the image.
  bool other_writes = false;    // SetPixels / Blit destination / unpack_rows
  int n_regions = 0;            // 0 none yet, 1 = every dispatch so far had region `rg`, 2 = mixed
  int rg[4] = {0, 0, 0, 0};     // region_w, region_h, first_group_row, row_stride
};

enum BindSlot { B_MESHOBJECTS, B_VERTICES, B_INDICES, B_NORMALS, B_SPHERES, B_MESHBVH, B_SPHEREBVH, B_COUNT };
const char* const kBindNames[B_COUNT] = {"_MeshObjects", "_Vertices", "_Indices", "_Normals", "_Spheres", "_MeshBVH", "_SphereBVH"};
const int kBindStride[B_COUNT] = {URT_STRIDE_MESHOBJECT, URT_STRIDE_VEC3, URT_STRIDE_INDEX, URT_STRIDE_VEC3,
                                  URT_STRIDE_SPHERE, URT_STRIDE_BVHNODE, URT_STRIDE_BVHNODE};

std::string g_create_error;   // urt_last_error(NULL)

// process-wide cache for urt_debug_build_blas / urt_debug_get_blas
BlasResult g_debug_blas;

}  // namespace

struct urt_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  std::string err;
  std::unordered_map<urt_handle, Buffer> buffers;
  std::unordered_map<urt_handle, Texture> textures;
  urt_handle next_id = 1;

  urt_handle bound[B_COUNT] = {0, 0, 0, 0, 0, 0, 0};
  urt_handle t_sky = 0, t_result = 0;
  float c2w[16] = {0}, invp[16] = {0};
  float pixel_off[2] = {0, 0};
  float seed = 0;
  int num_bounces = 0, num_rays = 0;     // shader uniforms default to 0 until SetInt (RM:780-781)

  // derived device scene
  bool scene_dirty = true;
  // what made it dirty: SetData on a bound buffer sets the slot's bit; anything else (binding changes, options) asks for a full
  // preparation.  When only _MeshObjects / _MeshBVH / _Spheres / _SphereBVH contents changed, the scene is updated in place
  // (prepare_incremental: moved MeshObjects are refitted on the GPU, csrc/refit.hip)
  unsigned int dirty_slots = 0;
  bool dirty_full = true;
  int opt_refit = 1;                        // 0 = always prepare from scratch
  int opt_qnodes = 0;                       // 32-byte quantized nodes in the traversal loop: 0 = off (default: measured -1.3 % on C3 / C3D, +1.3 % on C4 / C5 — the loop waits on the latency of ONE dependent fetch per step, not on its width), 1 = on, -1 = on unless a MeshObject is only a few grid cells wide
  float4* qbuf = nullptr;                   // frame + quantized nodes of the prepared scene (in scene_allocs)
  int sched_groups = 0;                     // kernel_mode 3: workgroups per CU the last configuration counts on when fewer than the default fit (0 = default)
  float4* cbuf = nullptr;                   // centre / half-extent copy of the nodes of the prepared scene (in scene_allocs): DevScene::blas_cnodes
  float qnode_quality = 0;                  // smallest MeshObject extent in grid cells (csrc/qnodes.hip)
  std::vector<uint8_t> prev_mesh_objects;   // the _MeshObjects records of the prepared scene
  std::vector<int32_t> h_mesh_root, h_small_first;
  struct RefitAux {                         // device-resident, part of the prepared scene (scene_allocs)
    const float* vertices = nullptr; const int32_t* indices = nullptr;      // copies of _Vertices / _Indices
    int32_t* parent = nullptr; int32_t* node_mesh = nullptr; int32_t* depth = nullptr;
    float4* cbox = nullptr; unsigned int* ext = nullptr;
    float* matrices = nullptr; int32_t* moved = nullptr;
    bool ready = false;
  } refit;
  size_t cap_materials = 0, cap_mesh_tlas = 0, cap_sphere_tlas = 0, cap_sphere_pr = 0;   // float4 capacities of the arrays updated in place
  uint64_t refitted_meshes = 0, incremental_preps = 0;
  DevScene ds{};
  std::vector<void*> scene_allocs;
  int tlas_stack = 2, blas_stack = 2;
  BlasCache blas_cache;                     // per-MeshObject BVHs of the previous scene (reused when a MeshObject is unchanged)
  int n_blas_nodes = 0;                     // interior nodes of the triangle-BVH forest (all meshes)
  unsigned int watchdog_steps = 1u << 16;
  float4* zero_sky = nullptr;

  // wavefront queues
  PathQueues q{};
  size_t q_capacity = 0, counts_capacity = 0;

  DevCounters* d_counters = nullptr;        // kCounterShards shards
  unsigned int* d_next = nullptr;           // persistent mode: frame work counter
  float4* d_mail = nullptr; size_t mail_slots = 0;   // kernel_mode 5: posted rays (2 float4 per thread of the resident grid)
  // frame tables of the batched launches: kTableSlots pinned host images + device copies, used round-robin; a slot is reused
  // once the copy of its previous use has left the host image (event)
  static constexpr int kTableSlots = 4;
  FrameUniforms* h_tables = nullptr; FrameUniforms* d_tables = nullptr;
  hipEvent_t table_ev[kTableSlots] = {nullptr, nullptr, nullptr, nullptr};
  unsigned int table_next = 0;
  uint64_t pixels_dispatched = 0;
  int n_cus = 256;

  uint64_t dispatches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timing;   // unresolved event pairs
  float trace_ms = 0;

  int opt_count_stats = 0, opt_time_dispatch = 0, opt_kernel_mode = 3;
  int opt_block_threads = 64, opt_xcd_run = 0 /* auto */, opt_work_shards = 64, opt_frame_group = 64, opt_refill_min = 16, opt_waves_per_cu = 0 /* auto */, opt_blas_min = 0 /* auto */, opt_blas_exit = 0 /* auto */;
  int opt_pool_k = 2, opt_pool_refill = 32, opt_pool_blas_min = 48, opt_pool_blas_exit = 8, opt_pool_inloop = 16, opt_pool_other_min = 24;   // kernel_mode 4
  int opt_sched_block = 0;                  // kernel_mode 3: threads per workgroup (64 or 256; 0 = 256 when there is a BVH top to share)
  int opt_stack_pad = 0;                    // test hook: extra (unused) entries per traversal stack, to reach the > 64 KiB LDS launch path
  int opt_shade_min = 32, opt_sky_min = 32; // kernel_mode 3
  int opt_serve_refill = 16;                // kernel_mode 5: idle lanes of the traversal service that trigger a claim of waiting rays
  int opt_front_list = -1;                  // kernel_mode 3: listed FRONT for scenes of <= 12 MeshObjects (-1 auto = on, 0 off)
  int opt_shade_split = -1;                 // kernel_mode 3: -1 = auto (= split: measured better or equal on C2-C5), 0 = surface hits and misses shaded in one trip
  int opt_tile_order = -1;                  // persistent modes: order in which the frame's tiles are handed out: 0 bottom strip first, 1 top strip first (a launch then ENDS with the
                                            // bottom rows), -1 = auto: top first for scenes without triangle meshes (C2: -3.6 % in bench.py, -9 .. -12 % for launches of 1 - 20 frames),
                                            // bottom first otherwise (C3, driver's 20-frame launch: top first +1.5 %; 64-frame launches, C3D, C4, C5: +-0.4 %) —
                                            // profiles/r03_logs/r3_probe_tile_order.log; any order draws the same pixels
  int opt_lds_tlas = 1;                     // kernel_mode 3: object-level heaps, roots and spheres in LDS when small
  int opt_top_front = -1;                   // kernel_mode 3: top-of-forest walk inside the object-level phase (-1 = when the scene has several meshes)
  int opt_top_nodes = -1;                   // kernel_mode 3: triangle-BVH nodes kept in LDS (0 = none; -1 = auto: 64, or with the masked object-level phase twice the number of
                                            // MeshObjects that have a BVH, rounded up to a power of two — there the top is walked lane by lane inside that phase and only its first level pays)

  // ---- frame batching (kernel_mode 3) --------------------------------------------------------------------------------
  // A 1080p frame is small for this chip: ~40 % of its kernel time is the drain of the last long paths.
  // So dispatches are DEFERRED: consecutive frames that differ only in their per-frame uniforms (camera, _PixelOffset,
  // _Seed) are collected and traced by ONE persistent launch whose lanes move on to the next frame's pixels as soon as
  // the current frame is handed out.  Each frame's Result goes to its own slot of a slab (the Result texture is renamed
  // per dispatch), the AdditionShader blits that follow the dispatches are deferred with them and run in order after the
  // launch.  Everything else that could observe the images flushes first, so the in-order semantics of RM:806-820
  // stay exactly observable.
  int opt_blas_builder = -1;                // -1 = auto (default): 0 below kGpuBuildTriangles triangles, 3 from there on; 0 = binned SAH on host threads, 1 = Karras radix tree built on the GPU,
                                            // 2 = the same tree built top-down within a depth budget, 3 = binned SAH on the GPU (csrc/lbvh.hip): the host's trees at a fifth of the time on big scenes
  float last_prepare_ms = 0;                // host wall time of the last scene preparation (buffers -> device scene)
  int n_scene_tris = 0;                     // triangles of the prepared scene
  int walk_f4 = 0;                          // float4s of the masked-walk table behind the mesh heap's device copy (0 = none: heap > 31 nodes)
  int scene_max_depth = 0;
  int opt_frames_per_launch = 0;            // 0 = auto (own stream: 64 frames per launch, fewer when the Result slots would exceed 8 GiB; caller's stream: 1), 1 = off, 2..64
  uint64_t scene_epoch = 0;                 // bumps at every scene preparation
  struct PostOp { int kind; int frame; urt_handle tex; urt_handle dst; float sample; int first_row, row_stride; void* dense; };   // kind 0 = blit_add(tex@frame -> dst), 1 = pack_rows(tex -> dense), 2 = blit(tex -> dst), the present of RM:819
  struct Pending {
    int n = 0, limit = 1;
    urt_handle tex = 0;                     // the Result texture of the batch
    uint64_t scene_epoch = 0;
    DevScene S{};
    FrameParams P{};                        // frame 0's; the frames agree on everything but the table entries
    FrameTable T{};
    int front_mode = 0; bool count = false;
    std::vector<PostOp> ops;
  } pend;
  float4* slab = nullptr;                   // slab_frames x slab_stride float4: Result slots of the batched frames
  size_t slab_stride = 0;
  int slab_frames = 0;
  urt_handle slab_tex = 0;                  // the texture whose `dev` may point into the slab
  std::vector<hipEvent_t> event_pool;       // recycled timing events
  hipEvent_t ev_switch = nullptr;           // orders the old stream before the new one in urt_context_set_stream
  uint64_t launches = 0;                    // trace-kernel launches (a batched launch counts once)
  // a wave that left a persistent kernel through one of its caps has not written its pixels: the kernels raise this host-mapped
  // word (kernels.hip report_watchdog) and the next synchronising call fails with URT_ERR_WATCHDOG
  unsigned int* h_trip_flag = nullptr;      // pinned, device-visible
  unsigned int* d_trip_flag = nullptr;      // its device address
  int opt_watchdog_cap = 0;                 // test hook: scheduler trips per wave (0 = auto, scaled with the launch)
  int slab_frames_max = 0;                  // largest batch the Result slab could be allocated for (after out-of-memory retries)
  urt_launch_info last_launch{};            // the last trace launch of this context (urt_debug_launch_info)
  int last_builder = 0;                     // the triangle-BVH builder the last full scene preparation used (0..3)
  // pipelined readback (urt_texture_read_begin / _end): kReadSlots snapshots in flight, each a device copy + a pinned host image
  static constexpr int kReadSlots = 3;
  struct ReadSlot { float4* dev = nullptr; float4* host = nullptr; size_t pixels = 0; size_t bytes = 0; int format = 0; hipEvent_t snap = nullptr, done = nullptr; bool busy = false; uint64_t ticket = 0; } rslot[kReadSlots];
  float* srgb_first = nullptr;                // device: first float of every 8-bit sRGB code (csrc/present.hip), made at the first RGBA8 readback
  hipStream_t copy_stream = nullptr;
  uint64_t read_next = 0;
  int opt_lbvh_slack = 6;                   // blas_builder 2: levels of slack in the depth budget (csrc/lbvh.hip k_td_level)
  int opt_front_cull = 1;                   // object-level cull (urt_math.h tlas_cull; csrc/cullflags.hip): 0 = every popped object is intersected, as the reference does
  int32_t* d_mesh_leaf = nullptr;           // per MeshObject: its heap leaf, or < 0 (in scene_allocs)
  size_t cap_mesh_leaf = 0;
  size_t slab_oom_stride = 0;               // image size (pixels) for which not even two slots could be allocated
  // Overlapped launches (option "overlap_launches", flush_pending): a host that SUBMITS every frame (urt_flush, a present into an external
  // texture) produces one-frame launches, and a one-frame launch is mostly ramp and drain.  Small launches therefore alternate between two trace streams and take their Result slots
  // round-robin from the slab, so that launch L+1 fills the wave slots launch L's draining waves give back; the blends / presents /
  // readbacks stay on the main stream, in program order, each behind its own launch.
  static constexpr int kOverlapFrames = 8;  // launches of up to this many frames take part
  int opt_overlap = 1;
  hipStream_t trace_q[2] = {nullptr, nullptr};
  hipEvent_t trace_done[2] = {nullptr, nullptr}, pre_ev[2] = {nullptr, nullptr}, dep_ev = nullptr;
  unsigned int* d_next2 = nullptr;          // the second launch in flight needs work counters of its own
  unsigned int trace_parity = 0;
  bool main_touched = true;                 // something other than the frame loop's own blends / presents / readbacks was enqueued on the main stream since the last launch
  int slab_cursor = 0, prev_base = 0, prev_n = 0;
  uint64_t overlapped_launches = 0;
};

namespace { inline hipStream_t touch(urt_context* ctx) { ctx->main_touched = true; return ctx->stream; } }

namespace {

int fail(urt_context* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg; else g_create_error = msg;
  return code;
}

#define URT_HIP(ctx, expr)                                                                         \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return fail(ctx, e__ == hipErrorOutOfMemory ? URT_ERR_OUT_OF_MEMORY : URT_ERR_HIP,           \
                  std::string(#expr) + ": " + hipGetErrorString(e__));                            \
  } while (0)

#define URT_GUARD_BEGIN try {
#define URT_GUARD_END(ctx)                                                                         \
  } catch (const std::bad_alloc&) { return fail(ctx, URT_ERR_OUT_OF_MEMORY, "host allocation failed"); } \
  catch (const std::exception& ex) { return fail(ctx, URT_ERR_INVALID_ARGUMENT, ex.what()); }        \
  catch (...) { return fail(ctx, URT_ERR_INVALID_ARGUMENT, "unknown exception"); }

// Scheduler trips a wave of a persistent kernel may make before it gives up (kernels.hip).  A frame needs 1e3-1e5; the cap
// grows with what the launch carries: frames x (rays x bounces / 8).
unsigned int sched_trip_cap(urt_context* ctx, const FrameParams& P, int n_frames) {
  if (ctx->opt_watchdog_cap > 0) return (unsigned int)ctx->opt_watchdog_cap;
  uint64_t per = std::max<uint64_t>(1, (uint64_t)std::max(1, P.num_rays) * (uint64_t)std::max(1, P.num_bounces) / 8u);
  uint64_t cap = (1ull << 24) * (uint64_t)std::max(1, n_frames) * per;
  return (unsigned int)std::min<uint64_t>(cap, 0xfffffff0ull);
}

// After the stream has been waited for: did a wave of the work just completed leave through a cap?
int check_watchdog(urt_context* ctx) {
  if (!ctx->h_trip_flag) return URT_OK;
  unsigned int n = __atomic_exchange_n(ctx->h_trip_flag, 0u, __ATOMIC_ACQ_REL);
  if (n == 0) return URT_OK;
  return fail(ctx, URT_ERR_WATCHDOG, std::to_string(n) + " wave(s) of a trace launch hit the kernel's iteration cap and left pixels unwritten "
                                     "(urt_counters.watchdog_trips): the images written since the last successful synchronisation are incomplete");
}

void free_scene(urt_context* ctx) {
  if (!ctx->scene_allocs.empty()) (void)hipStreamSynchronize(touch(ctx));   // queued kernels may still read them
  for (void* p : ctx->scene_allocs) (void)hipFree(p);
  ctx->scene_allocs.clear();
  ctx->ds = DevScene{};
  ctx->refit = urt_context::RefitAux{};
  ctx->qbuf = nullptr; ctx->cbuf = nullptr; ctx->d_mesh_leaf = nullptr; ctx->cap_mesh_leaf = 0;
  ctx->cap_materials = ctx->cap_mesh_tlas = ctx->cap_sphere_tlas = ctx->cap_sphere_pr = 0;
  ctx->slab_oom_stride = 0;                                // device memory came back: the next batch may try the Result slots again
}

template <typename T>
int upload(urt_context* ctx, const std::vector<T>& v, const float4** out) {
  *out = nullptr;
  if (v.empty()) return URT_OK;
  void* d = nullptr;
  URT_HIP(ctx, hipMalloc(&d, v.size() * sizeof(T)));
  ctx->scene_allocs.push_back(d);
  // synchronous on purpose: `v` is a short-lived staging vector, and a pageable-memory hipMemcpyAsync may
  // still be reading it after this function returns
  URT_HIP(ctx, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const float4*)d;
  return URT_OK;
}

const Buffer* bound_buffer(urt_context* ctx, int slot) {
  urt_handle h = ctx->bound[slot];
  if (!h) return nullptr;
  auto it = ctx->buffers.find(h);
  if (it == ctx->buffers.end() || !it->second.has_data || it->second.count == 0) return nullptr;
  return &it->second;
}

// Everything Shade (RS:388-419) derives from the material ALONE is evaluated here, once per material, with the normative
// arithmetic of include/urt_math.h in the shader's own operation order (the same functions the oracle evaluates per hit, so the
// bits are the same): the clamped albedo, the two normalised roulette chances and their sum, the Phong exponent
// alpha = pow(1000, smoothness^2), 1/(alpha+1), (alpha+2)/(alpha+1) and the two energy factors (1/chance) * colour.
// Per hit the kernel then loads 64 bytes and skips two dot products, a pow and six IEEE divisions.
//   [0] (1/diffChance) * albedo', specChance      [1] (1/specChance) * specular, specChance + diffChance
//   [2] emission, diffChance                       [3] alpha, 1/(alpha+1), (alpha+2)/(alpha+1), 0
constexpr int kMatFloats = 16;
void pack_material(const urt_RayTraceParams& m, float* dst) {
  using namespace urt;
  v3 albedo = mk3(m.color_albedo[0], m.color_albedo[1], m.color_albedo[2]);
  v3 spec = mk3(m.color_specular[0], m.color_specular[1], m.color_specular[2]);
  albedo = vmin3(mk3(1.0f, 1.0f, 1.0f) - spec, albedo);                          // RS:390
  const float third = 1.0f / 3.0f;
  float specChance = dot(spec, mk3(third, third, third));                        // RS:391-392
  float diffChance = dot(albedo, mk3(third, third, third));
  float sum = specChance + diffChance;                                           // RS:393-395
  specChance /= sum;
  diffChance /= sum;
  float alpha = f_pow(1000.0f, m.smoothness * m.smoothness);                     // RS:401
  v3 ks = (1.0f / specChance) * spec;                                            // RS:405
  v3 kd = (1.0f / diffChance) * albedo;                                          // RS:411
  dst[0] = kd.x; dst[1] = kd.y; dst[2] = kd.z; dst[3] = specChance;
  dst[4] = ks.x; dst[5] = ks.y; dst[6] = ks.z; dst[7] = specChance + diffChance;
  dst[8] = m.emission[0]; dst[9] = m.emission[1]; dst[10] = m.emission[2]; dst[11] = diffChance;
  dst[12] = alpha; dst[13] = 1.0f / (alpha + 1.0f); dst[14] = (alpha + 2) / (alpha + 1); dst[15] = 0.0f;   // RS:104, 404
}

int heap_levels(int n) { int l = 0; while (n > 0) { l++; n >>= 1; } return l; }   // floor(log2 n) + 1

// "Masked" object-level walk (kernels.hip front_masked): for a mesh heap of <= 31 nodes the walk RS:294-326 is evaluated without
// a stack.  Which nodes a ray pops depends only on the slab tests of their ancestors, and the pop order (children pushed 2i+1
// then 2i+2, so the right child is popped first) is a static pre-order of the heap.  The heap is therefore re-indexed in that
// order ("position"): the right child of the node at position p sits at p + 1, the left child at p + 2^(h-1), h = levels below and
// including p.  One bit per position: H = slab test passed, P = popped (root; children of a popped, hit, interior node — a shift
// per level), objects to test = popped leaves from the first popped-and-hit leaf on (`tests` is never reset, A.5), in position
// order = pop order.  The table appended to the device copy of the heap (float4 units; layout shared with kernels.hip):
//   [0]  n_eval, levels, interior mask, exist mask          [1] leaf_any mask, leaf_valid mask, 0, 0
//   [2]  depth masks d = 0..3                                [3] left-child shifts d = 0..3
//   [4 .. 20)  per position p = 0..31: int2 {triangle-BVH root of the MeshObject, first triangle in the LDS copy of the single-leaf
//              MeshObjects or -1}
//   [20 .. 20 + 2 * n_eval)  the nodes whose slab test can matter (inside the buffer, non-empty bounds, every ancestor an interior
//              non-empty node), in position order: vmin.xyz, position bit of the parent (0 = root) | vmax.xyz, position bit
constexpr int kWalkHeaderF4 = 20;
bool build_walk_table(const Buffer* heap, int n_meshes, const std::vector<int32_t>& mesh_root, const std::vector<int32_t>& small_first,
                      std::vector<float>& out) {
  out.clear();
  if (!heap || heap->count < 1 || heap->count > 31) return false;
  const int n = heap->count, D = heap_levels(n);            // complete tree of D levels holds the array
  const int N = (1 << D) - 1;
  std::vector<int> pos((size_t)N, -1), depth((size_t)N, 0);
  {   // right-first pre-order positions of the complete tree's slots
    std::vector<int> stack{0};
    int next = 0;
    while (!stack.empty()) {
      int i = stack.back(); stack.pop_back();
      pos[(size_t)i] = next++;
      if (2 * i + 2 < N) { depth[(size_t)(2 * i + 1)] = depth[(size_t)(2 * i + 2)] = depth[(size_t)i] + 1; stack.push_back(2 * i + 1); stack.push_back(2 * i + 2); }
    }
  }
  auto node = [&](int i) { urt_BVHNode nd; std::memcpy(&nd, heap->host.data() + (size_t)i * URT_STRIDE_BVHNODE, sizeof nd); return nd; };
  uint32_t imask = 0, exist = 0, leaf_any = 0, leaf_valid = 0, dm[4] = {0, 0, 0, 0};
  int32_t ls[4] = {0, 0, 0, 0};
  std::vector<int32_t> pos_tab(64, 0);
  for (int p = 0; p < 32; p++) { pos_tab[(size_t)(2 * p)] = kEmptyMeshRoot; pos_tab[(size_t)(2 * p + 1)] = -1; }   // (positions that are never tested)
  std::vector<char> live((size_t)N, 0);                      // slab test can matter
  struct Ev { int p, parent_p; urt_BVHNode nd; };
  std::vector<Ev> ev;
  for (int i = 0; i < N; i++) {
    const int p = pos[(size_t)i], d = depth[(size_t)i];
    if (d < D - 1 && d < 4) { dm[d] |= 1u << p; ls[d] = 1 << (D - d - 1); }
    if (i >= n) continue;
    urt_BVHNode nd = node(i);
    exist |= 1u << p;
    if (nd.index < 0) imask |= 1u << p; else leaf_any |= 1u << p;
    bool nonempty = !(nd.vmin[0] == nd.vmax[0] && nd.vmin[1] == nd.vmax[1] && nd.vmin[2] == nd.vmax[2]);      // RS:273
    bool parent_ok = i == 0 || (live[(size_t)((i - 1) / 2)] && node((i - 1) / 2).index < 0);
    live[(size_t)i] = nonempty && parent_ok;
    if (nd.index >= 0 && nd.index < n_meshes && mesh_root[(size_t)nd.index] != kEmptyMeshRoot) {      // a MeshObject without triangles is never tested
      leaf_valid |= 1u << p;
      pos_tab[(size_t)(2 * p)] = mesh_root[(size_t)nd.index];
      pos_tab[(size_t)(2 * p + 1)] = small_first.empty() ? -1 : small_first[(size_t)nd.index];
    }
    if (live[(size_t)i]) ev.push_back(Ev{p, i == 0 ? -1 : pos[(size_t)((i - 1) / 2)], nd});
  }
  std::sort(ev.begin(), ev.end(), [](const Ev& a, const Ev& b) { return a.p < b.p; });
  out.assign((size_t)(kWalkHeaderF4 + 2 * ev.size()) * 4, 0.0f);
  auto put = [&](size_t word, int32_t v) { std::memcpy(&out[word], &v, 4); };
  put(0, (int32_t)ev.size()); put(1, D); put(2, (int32_t)imask); put(3, (int32_t)exist);
  put(4, (int32_t)leaf_any); put(5, (int32_t)leaf_valid);
  for (int d = 0; d < 4; d++) { put(8 + (size_t)d, (int32_t)dm[d]); put(12 + (size_t)d, ls[d]); }
  for (size_t k = 0; k < 64; k++) put(16 + k, pos_tab[k]);
  for (size_t e = 0; e < ev.size(); e++) {
    float* o = out.data() + (size_t)(kWalkHeaderF4 + 2 * e) * 4;
    o[0] = ev[e].nd.vmin[0]; o[1] = ev[e].nd.vmin[1]; o[2] = ev[e].nd.vmin[2];
    int32_t pbit = ev[e].parent_p < 0 ? 0 : (int32_t)(1u << ev[e].parent_p); std::memcpy(&o[3], &pbit, 4);
    o[4] = ev[e].nd.vmax[0]; o[5] = ev[e].nd.vmax[1]; o[6] = ev[e].nd.vmax[2];
    int32_t bit = (int32_t)(1u << ev[e].p); std::memcpy(&o[7], &bit, 4);
  }
  return true;
}

void pack_tlas(const Buffer* b, std::vector<float>& out, const std::vector<int32_t>* cull_words) {
  out.clear();
  if (!b) return;
  out.resize((size_t)b->count * 8);
  for (int i = 0; i < b->count; i++) {
    urt_BVHNode nd;
    std::memcpy(&nd, b->host.data() + (size_t)i * URT_STRIDE_BVHNODE, sizeof nd);
    float* o = out.data() + (size_t)i * 8;
    o[0] = nd.vmin[0]; o[1] = nd.vmin[1]; o[2] = nd.vmin[2]; std::memcpy(&o[3], &nd.index, 4);
    o[4] = nd.vmax[0]; o[5] = nd.vmax[1]; o[6] = nd.vmax[2]; o[7] = 0;
    if (cull_words && (size_t)i < cull_words->size()) std::memcpy(&o[7], &(*cull_words)[(size_t)i], 4);      // (0 = never culled)
  }
}

// right-first pre-order position of every slot of the complete tree that holds an n-node heap (the pop order of RS:294-326; build_walk_table)
void heap_positions(int n, std::vector<int>& pos, std::vector<int>& depth, int* levels, int* slots) {
  const int D = heap_levels(n), N = (1 << D) - 1;
  pos.assign((size_t)N, -1); depth.assign((size_t)N, 0);
  std::vector<int> stack{0};
  int next = 0;
  while (!stack.empty()) {
    int i = stack.back(); stack.pop_back();
    pos[(size_t)i] = next++;
    if (2 * i + 2 < N) { depth[(size_t)(2 * i + 1)] = depth[(size_t)(2 * i + 2)] = depth[(size_t)i] + 1; stack.push_back(2 * i + 1); stack.push_back(2 * i + 2); }
  }
  *levels = D; *slots = N;
}

// Object-level cull (urt_math.h tlas_cull): the cull word of every heap node — non-zero for the leaves that are ELIGIBLE: a MeshObject
// with triangles that exactly one leaf of the heap names.  (A lone MeshObject gains too: a ray that leaves it behind, or meets the ground
// first, skips the round trip through the triangle-BVH phase — C3 -1.5 %, C3D -2.5 %, profiles/r04_logs/r4_ab_front_cull.log.)
// The word is the leaf's position bit of the masked walk (heaps of <= 31 nodes) or 1.
// csrc/cullflags.hip then clears the word of every leaf whose box does not contain its object's triangles.  mesh_leaf[m] = that leaf, or -1.
void cull_words(const urt_context* ctx, const Buffer* heap, int n_meshes, const std::vector<int32_t>& mesh_root, std::vector<int32_t>& words,
                std::vector<int32_t>& mesh_leaf) {
  const int n = heap ? heap->count : 0;
  words.assign((size_t)n, 0);
  mesh_leaf.assign((size_t)std::max(0, n_meshes), -1);
  if (!ctx->opt_front_cull || n_meshes < 1 || n < 1) return;
  std::vector<int> refs((size_t)n_meshes, 0);
  auto node = [&](int i) { urt_BVHNode nd; std::memcpy(&nd, heap->host.data() + (size_t)i * URT_STRIDE_BVHNODE, sizeof nd); return nd; };
  for (int i = 0; i < n; i++) { urt_BVHNode nd = node(i); if (nd.index >= 0 && nd.index < n_meshes) refs[(size_t)nd.index]++; }
  std::vector<int> pos, depth; int D = 0, N = 0;
  if (n <= 31) heap_positions(n, pos, depth, &D, &N);
  for (int i = 0; i < n; i++) {
    urt_BVHNode nd = node(i);
    if (nd.index < 0 || nd.index >= n_meshes || refs[(size_t)nd.index] != 1) continue;
    if ((size_t)nd.index >= mesh_root.size() || mesh_root[(size_t)nd.index] == kEmptyMeshRoot) continue;
    if (nd.vmin[0] == nd.vmax[0] && nd.vmin[1] == nd.vmax[1] && nd.vmin[2] == nd.vmax[2]) continue;       // RS:273: never passes the slab test, its t values are not computed
    words[(size_t)i] = n <= 31 ? (int32_t)(1u << pos[(size_t)i]) : 1;
    mesh_leaf[(size_t)nd.index] = i;
  }
}

// Upload mesh_leaf and run the verification pass over the prepared scene's triangle records (after a build and after every refit).
int verify_cull_flags(urt_context* ctx, const std::vector<int32_t>& words, const std::vector<int32_t>& mesh_leaf) {
  DevScene& S = ctx->ds;
  bool any = false;
  for (int32_t w : words) any = any || w != 0;
  S.cull_any = 0;
  if (!any || S.n_mesh_tlas <= 0 || !S.mesh_tlas) return URT_OK;
  S.cull_any = 1;
  if (mesh_leaf.size() > ctx->cap_mesh_leaf || !ctx->d_mesh_leaf) {
    void* p = nullptr;
    URT_HIP(ctx, hipMalloc(&p, std::max<size_t>(16, mesh_leaf.size() * sizeof(int32_t))));
    ctx->scene_allocs.push_back(p);
    ctx->d_mesh_leaf = (int32_t*)p; ctx->cap_mesh_leaf = mesh_leaf.size();
  }
  URT_HIP(ctx, hipMemcpy(ctx->d_mesh_leaf, mesh_leaf.data(), mesh_leaf.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  int* mask = ctx->walk_f4 > 0 ? (int*)const_cast<float4*>(S.mesh_tlas + 2 * (size_t)S.n_mesh_tlas) + 6 : nullptr;      // header word [6] of the walk table behind the heap
  URT_HIP(ctx, update_cull_flags(const_cast<float4*>(S.mesh_tlas), S.n_mesh_tlas, ctx->d_mesh_leaf, S.n_meshes, S.tri_verts, ctx->n_scene_tris, mask, touch(ctx)));
  return URT_OK;
}

int requantize(urt_context* ctx);
// (Re)derive what the trace kernels read from the [lo, hi] nodes of the prepared scene — after a build and after every refit: the
// centre / half-extent copy (always) and the quantized copy (option qnodes).
int rederive_nodes(urt_context* ctx) {
  DevScene& S = ctx->ds;
  S.blas_cnodes = nullptr;
  if (ctx->cbuf && ctx->n_blas_nodes > 0) {
    URT_HIP(ctx, center_nodes(S.blas_nodes, ctx->n_blas_nodes, ctx->cbuf, touch(ctx)));
    S.blas_cnodes = ctx->cbuf;
  }
  return requantize(ctx);
}
// (Re)derive the quantized nodes from the float nodes of the prepared scene and decide whether the traversal loop uses them.
int requantize(urt_context* ctx) {
  DevScene& S = ctx->ds;
  S.blas_qnodes = nullptr;
  if (ctx->opt_qnodes == 0 || !ctx->qbuf || ctx->n_blas_nodes <= 0) return URT_OK;
  URT_HIP(ctx, quantize_nodes(S.blas_nodes, ctx->n_blas_nodes, S.mesh_root, S.n_meshes, ctx->qbuf, touch(ctx)));
  float4 f0;
  URT_HIP(ctx, hipMemcpyAsync(&f0, ctx->qbuf, sizeof f0, hipMemcpyDeviceToHost, touch(ctx)));
  URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));
  ctx->qnode_quality = f0.w;
  // one grid for the whole forest: a MeshObject that spans only a few hundred cells would have boxes of a few cells — every ray through
  // it would walk most of its tree.  Such scenes keep the float nodes (auto); "qnodes" = 1 insists.
  if (ctx->opt_qnodes == 1 || f0.w >= 1024.0f) S.blas_qnodes = ctx->qbuf;
  return URT_OK;
}

// Update a small device array of the prepared scene: in place while it fits its allocation, else a new allocation (the old one
// stays in scene_allocs until the next full preparation).  The stream has been waited for.
int update_array(urt_context* ctx, const std::vector<float>& v, const float4** dev, size_t* cap_f4) {
  size_t need = (v.size() + 3) / 4;
  if (need == 0) { *dev = nullptr; return URT_OK; }
  if (*dev && need <= *cap_f4) {
    URT_HIP(ctx, hipMemcpy(const_cast<float4*>(*dev), v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return URT_OK;
  }
  int rc = upload(ctx, v, dev);
  if (rc == URT_OK) *cap_f4 = need;
  return rc;
}

bool build_walk_table(const Buffer* heap, int n_meshes, const std::vector<int32_t>& mesh_root, const std::vector<int32_t>& small_first, std::vector<float>& out);
void pack_tlas(const Buffer* b, std::vector<float>& out, const std::vector<int32_t>* cull_words = nullptr);
void cull_words(const urt_context* ctx, const Buffer* heap, int n_meshes, const std::vector<int32_t>& mesh_root, std::vector<int32_t>& words, std::vector<int32_t>& mesh_leaf);
int verify_cull_flags(urt_context* ctx, const std::vector<int32_t>& words, const std::vector<int32_t>& mesh_leaf);

// The dynamic-scene path (RM:215-230: a moved object makes the reference re-upload every buffer).  When the only contents that changed
// since the scene was prepared are those of _MeshObjects / _MeshBVH / _Spheres / _SphereBVH — same counts, same index ranges per
// MeshObject — the device scene is UPDATED: materials, object-level heaps and sphere tables are re-packed (a few KB), and every
// MeshObject whose localToWorldMatrix changed keeps its triangle BVH's topology: its triangle records and boxes are recomputed on the
// GPU (csrc/refit.hip).  Returns 1 when the change is not of that kind (the caller prepares from scratch).
int prepare_incremental(urt_context* ctx) {
  const unsigned int small = (1u << B_MESHOBJECTS) | (1u << B_MESHBVH) | (1u << B_SPHERES) | (1u << B_SPHEREBVH);
  if (!ctx->opt_refit || ctx->dirty_full || ctx->scene_allocs.empty() || (ctx->dirty_slots & ~small)) return 1;
  DevScene& S = ctx->ds;
  const Buffer* bm = bound_buffer(ctx, B_MESHOBJECTS);
  const Buffer* bs = bound_buffer(ctx, B_SPHERES);
  const Buffer* bmt = bound_buffer(ctx, B_MESHBVH);
  const Buffer* bst = bound_buffer(ctx, B_SPHEREBVH);
  const int n_meshes = bm ? bm->count : 0, n_spheres = bs ? bs->count : 0;
  if (n_meshes != S.n_meshes || n_spheres != S.n_spheres) return 1;
  if ((size_t)n_meshes * URT_STRIDE_MESHOBJECT != ctx->prev_mesh_objects.size()) return 1;
  auto t_begin = std::chrono::steady_clock::now();
  std::vector<int32_t> moved((size_t)n_meshes, 0);
  std::vector<float> matrices((size_t)n_meshes * 16, 0.0f);
  int n_moved = 0;
  for (int m = 0; m < n_meshes; m++) {
    urt_MeshObject a, b;
    std::memcpy(&a, ctx->prev_mesh_objects.data() + (size_t)m * URT_STRIDE_MESHOBJECT, sizeof a);
    std::memcpy(&b, bm->host.data() + (size_t)m * URT_STRIDE_MESHOBJECT, sizeof b);
    if (a.indices_offset != b.indices_offset || a.indices_count != b.indices_count) return 1;
    std::memcpy(&matrices[(size_t)m * 16], b.localToWorldMatrix, 64);
    if (std::memcmp(a.localToWorldMatrix, b.localToWorldMatrix, 64) != 0 && b.indices_count >= 3) { moved[(size_t)m] = 1; n_moved++; }
  }
  if (n_moved > 0 && (!ctx->refit.ready || ctx->n_scene_tris <= 0)) return 1;
  URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));           // frames in flight read the arrays that are about to change
  int rc;
  // materials: spheres, mesh objects, ground plane (pack_material: what Shade derives from the material alone)
  {
    std::vector<float> mats((size_t)(n_meshes + n_spheres + 1) * kMatFloats);
    urt_RayTraceParams ground{};
    ground.color_albedo[0] = 0.5f; ground.color_albedo[1] = 0.3f; ground.color_albedo[2] = 0.15f; ground.smoothness = 0.3f;
    pack_material(ground, mats.data() + (size_t)(n_meshes + n_spheres) * kMatFloats);
    for (int m = 0; m < n_meshes; m++) {
      urt_MeshObject mo; std::memcpy(&mo, bm->host.data() + (size_t)m * URT_STRIDE_MESHOBJECT, sizeof mo);
      pack_material(mo.lighting, mats.data() + (size_t)(n_spheres + m) * kMatFloats);
    }
    std::vector<float> pr((size_t)n_spheres * 4);
    for (int i = 0; i < n_spheres; i++) {
      urt_Sphere sp; std::memcpy(&sp, bs->host.data() + (size_t)i * URT_STRIDE_SPHERE, sizeof sp);
      pr[4 * (size_t)i] = sp.position[0]; pr[4 * (size_t)i + 1] = sp.position[1]; pr[4 * (size_t)i + 2] = sp.position[2]; pr[4 * (size_t)i + 3] = sp.radius;
      pack_material(sp.lighting, mats.data() + (size_t)i * kMatFloats);
    }
    if ((rc = update_array(ctx, mats, &S.materials, &ctx->cap_materials))) return rc;
    if (n_spheres > 0 && (rc = update_array(ctx, pr, &S.sphere_pr, &ctx->cap_sphere_pr))) return rc;
  }
  // object-level heaps (+ the masked-walk table of a small mesh heap)
  std::vector<int32_t> cw, mesh_leaf;
  {
    std::vector<float> t, walk;
    cull_words(ctx, bmt, n_meshes, ctx->h_mesh_root, cw, mesh_leaf);
    pack_tlas(bmt, t, &cw);
    ctx->walk_f4 = 0;
    if (n_meshes > 0 && build_walk_table(bmt, n_meshes, ctx->h_mesh_root, ctx->h_small_first, walk)) {
      ctx->walk_f4 = (int)(walk.size() / 4);
      t.insert(t.end(), walk.begin(), walk.end());
    }
    if ((rc = update_array(ctx, t, &S.mesh_tlas, &ctx->cap_mesh_tlas))) return rc;
    S.n_mesh_tlas = bmt ? bmt->count : 0;
    pack_tlas(bst, t);
    if ((rc = update_array(ctx, t, &S.sphere_tlas, &ctx->cap_sphere_tlas))) return rc;
    S.n_sphere_tlas = bst ? bst->count : 0;
    int lv = std::max(heap_levels(S.n_mesh_tlas), heap_levels(S.n_sphere_tlas));
    if (lv + 1 > 32)
      return fail(ctx, URT_ERR_SCENE, "object-level BVH deeper than the reference's 32-entry traversal stack (RS:73-74)");
    ctx->tlas_stack = std::max(2, lv + 1);
  }
  if (n_moved > 0) {
    URT_HIP(ctx, hipMemcpy(ctx->refit.matrices, matrices.data(), matrices.size() * sizeof(float), hipMemcpyHostToDevice));
    URT_HIP(ctx, hipMemcpy(ctx->refit.moved, moved.data(), moved.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    URT_HIP(ctx, refit_moved(const_cast<float4*>(S.blas_nodes), ctx->n_blas_nodes, const_cast<float4*>(S.tri_verts), ctx->n_scene_tris,
                             ctx->refit.vertices, ctx->refit.indices, ctx->refit.depth, std::max(0, ctx->scene_max_depth - 1), ctx->refit.node_mesh,
                             ctx->refit.matrices, ctx->refit.moved, ctx->refit.ext, n_meshes, ctx->refit.cbox, touch(ctx)));
    ctx->refitted_meshes += (uint64_t)n_moved;
    if ((rc = rederive_nodes(ctx))) return rc;
  }
  if ((rc = verify_cull_flags(ctx, cw, mesh_leaf))) return rc;       // against the (refitted) triangle records
  ctx->prev_mesh_objects.assign(bm ? bm->host.begin() : ctx->prev_mesh_objects.begin(), bm ? bm->host.begin() + (ptrdiff_t)((size_t)n_meshes * URT_STRIDE_MESHOBJECT) : ctx->prev_mesh_objects.begin());
  ctx->scene_dirty = false; ctx->dirty_slots = 0; ctx->dirty_full = false;
  ctx->scene_epoch++;
  ctx->incremental_preps++;
  ctx->last_prepare_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return URT_OK;
}

// Derive the device scene from the bound ComputeBuffers (runs at the first dispatch after a change;
// the reference pays the equivalent in RebuildTrees -> SetData, RM:725-746).
int prepare_scene(urt_context* ctx) {
  {
    int rc = prepare_incremental(ctx);
    if (rc != 1) return rc;                                 // updated in place (or failed)
  }
  free_scene(ctx);
  DevScene& S = ctx->ds;
  const Buffer* bm = bound_buffer(ctx, B_MESHOBJECTS);
  const Buffer* bv = bound_buffer(ctx, B_VERTICES);
  const Buffer* bi = bound_buffer(ctx, B_INDICES);
  const Buffer* bn = bound_buffer(ctx, B_NORMALS);
  const Buffer* bs = bound_buffer(ctx, B_SPHERES);
  const Buffer* bmt = bound_buffer(ctx, B_MESHBVH);
  const Buffer* bst = bound_buffer(ctx, B_SPHEREBVH);

  int rc;
  int n_meshes = bm ? bm->count : 0;
  int n_spheres = bs ? bs->count : 0;
  std::vector<float> mats((size_t)(n_meshes + n_spheres + 1) * kMatFloats);   // spheres first, then mesh objects, then the ground plane
  {
    urt_RayTraceParams ground{};                                                 // RS:164-170: hard-coded material of the y = 0 plane
    ground.color_albedo[0] = 0.5f; ground.color_albedo[1] = 0.3f; ground.color_albedo[2] = 0.15f;
    ground.smoothness = 0.3f;
    pack_material(ground, mats.data() + (size_t)(n_meshes + n_spheres) * kMatFloats);
  }
  // meshes: the triangle BVH ("BLAS") of every MeshObject, by the host SAH builder or by the GPU LBVH builder
  auto t_begin = std::chrono::steady_clock::now();
  std::vector<int32_t> mesh_root_host, small_first;
  int blas_max_depth = 0;
  size_t n_blas_nodes = 0, n_tris = 0;
  if (n_meshes > 0) {
    for (int m = 0; m < n_meshes; m++) {
      urt_MeshObject mo;
      std::memcpy(&mo, bm->host.data() + (size_t)m * URT_STRIDE_MESHOBJECT, sizeof mo);
      pack_material(mo.lighting, mats.data() + (size_t)(n_spheres + m) * kMatFloats);
    }
    const float4* p;
    // auto: the host builder up to kGpuBuildTriangles triangles (C3's 69,600: 12 ms on the host, 6 on the GPU — and small scenes are what the
    // per-MeshObject host cache is good at), the GPU's binned SAH beyond (C4 300 k: 32 -> 10 ms, C5 983 k: 80 -> 15 ms; same trees, same frames)
    constexpr long kGpuBuildTriangles = 200000;
    int builder = ctx->opt_blas_builder;
    if (builder < 0) {
      long tris = 0;
      for (int m = 0; m < n_meshes; m++) { urt_MeshObject mo; std::memcpy(&mo, bm->host.data() + (size_t)m * URT_STRIDE_MESHOBJECT, sizeof mo); tris += std::max(0, mo.indices_count) / 3; }
      builder = tris >= kGpuBuildTriangles ? 3 : 0;
    }
    ctx->last_builder = builder;
    if (builder >= 1) {
      // device copies of the buffers exactly as SetData delivered them; the whole build runs on the GPU (csrc/lbvh.hip)
      std::vector<int32_t> offs((size_t)n_meshes), cnts((size_t)n_meshes);
      for (int m = 0; m < n_meshes; m++) {
        urt_MeshObject mo;
        std::memcpy(&mo, bm->host.data() + (size_t)m * URT_STRIDE_MESHOBJECT, sizeof mo);
        offs[(size_t)m] = mo.indices_offset; cnts[(size_t)m] = mo.indices_count;
      }
      void* raw = nullptr;
      size_t b_mo = ((size_t)n_meshes * URT_STRIDE_MESHOBJECT + 255) & ~(size_t)255;
      size_t b_v = bv ? (((size_t)bv->count * 12 + 255) & ~(size_t)255) : 0, b_i = bi ? (((size_t)bi->count * 4 + 255) & ~(size_t)255) : 0;
      size_t b_n = bn ? (((size_t)bn->count * 12 + 255) & ~(size_t)255) : 0;
      URT_HIP(ctx, hipMalloc(&raw, b_mo + b_v + b_i + b_n + 256));
      char* rb = (char*)raw;
      hipError_t e = hipMemcpy(rb, bm->host.data(), (size_t)n_meshes * URT_STRIDE_MESHOBJECT, hipMemcpyHostToDevice);
      if (e == hipSuccess && bv) e = hipMemcpy(rb + b_mo, bv->host.data(), (size_t)bv->count * 12, hipMemcpyHostToDevice);
      if (e == hipSuccess && bi) e = hipMemcpy(rb + b_mo + b_v, bi->host.data(), (size_t)bi->count * 4, hipMemcpyHostToDevice);
      if (e == hipSuccess && bn) e = hipMemcpy(rb + b_mo + b_v + b_i, bn->host.data(), (size_t)bn->count * 12, hipMemcpyHostToDevice);
      if (e != hipSuccess) { (void)hipFree(raw); return fail(ctx, URT_ERR_HIP, std::string("scene upload: ") + hipGetErrorString(e)); }
      LbvhInput in;
      in.mesh_objects = (const uint8_t*)rb; in.n_meshes = n_meshes;
      in.vertices = bv ? (const float*)(rb + b_mo) : nullptr; in.n_vertices = bv ? bv->count : 0;
      in.indices = bi ? (const int32_t*)(rb + b_mo + b_v) : nullptr; in.n_indices = bi ? bi->count : 0;
      in.normals = bn ? (const float*)(rb + b_mo + b_v + b_i) : nullptr; in.n_normals = bn ? bn->count : 0;
      in.h_offsets = offs.data(); in.h_counts = cnts.data(); in.leaf_max = get_blas_leaf_max(); in.depth_budget = builder == 2; in.depth_slack = ctx->opt_lbvh_slack; in.sah = builder == 3;
      LbvhOutput o;
      std::string err;
      rc = lbvh_build(in, touch(ctx), o, err);
      if (rc) { (void)hipFree(raw); return fail(ctx, rc, err); }
      ctx->scene_allocs.push_back(raw);                     // _Vertices / _Indices stay resident: a moved MeshObject is refitted from them
      ctx->refit.vertices = in.vertices; ctx->refit.indices = in.indices;
      for (void* a : o.allocs) ctx->scene_allocs.push_back(a);
      S.mesh_root = o.mesh_root; S.blas_nodes = o.nodes; S.tri_verts = o.tri_verts; S.tri_norms = o.tri_norms;
      mesh_root_host = o.h_mesh_root; blas_max_depth = o.max_depth; n_blas_nodes = (size_t)o.n_nodes; n_tris = (size_t)o.n_tris;
    } else {
      BlasResult blas;
      std::string err;
      if (!build_blas(bm->host.data(), n_meshes, bv ? (const float*)bv->host.data() : nullptr, bv ? bv->count : 0,
                      bi ? (const int32_t*)bi->host.data() : nullptr, bi ? bi->count : 0,
                      bn ? (const float*)bn->host.data() : nullptr, bn ? bn->count : 0, blas, err, &ctx->blas_cache))
        return fail(ctx, URT_ERR_SCENE, err);
      if ((rc = upload(ctx, blas.mesh_root, &p))) return rc; S.mesh_root = (const int32_t*)p;
      if ((rc = upload(ctx, blas.nodes, &p))) return rc; S.blas_nodes = p;
      if ((rc = upload(ctx, blas.tri_verts, &p))) return rc; S.tri_verts = p;
      if ((rc = upload(ctx, blas.tri_norms, &p))) return rc; S.tri_norms = p;
      mesh_root_host = blas.mesh_root; blas_max_depth = blas.max_depth; n_blas_nodes = blas.nodes.size() / kBlasNodeFloats; n_tris = blas.tri_slot.size();
    }
    {   // single-leaf MeshObjects: where their triangles sit in the LDS copy (kernels.hip k_sched prologue)
      small_first.assign((size_t)n_meshes, -1);
      int n_small = 0;
      for (int m = 0; m < n_meshes; m++) {
        int32_t r = mesh_root_host[(size_t)m];
        if (r < 0 && r != (int32_t)0x80000000) { small_first[(size_t)m] = n_small; n_small += (int)((~(uint32_t)r) & 7u) + 1; }
      }
      if (n_small > 0 && n_small <= 64) {
        if ((rc = upload(ctx, small_first, &p))) return rc;
        S.mesh_small_first = (const int32_t*)p; S.n_small = n_small;
      } else small_first.assign((size_t)n_meshes, -1);

    }
  }
  S.n_meshes = n_meshes;
  // spheres
  if (n_spheres > 0) {
    std::vector<float> pr((size_t)n_spheres * 4);
    for (int i = 0; i < n_spheres; i++) {
      urt_Sphere sp;
      std::memcpy(&sp, bs->host.data() + (size_t)i * URT_STRIDE_SPHERE, sizeof sp);
      pr[4 * (size_t)i] = sp.position[0]; pr[4 * (size_t)i + 1] = sp.position[1]; pr[4 * (size_t)i + 2] = sp.position[2];
      pr[4 * (size_t)i + 3] = sp.radius;
      pack_material(sp.lighting, mats.data() + (size_t)i * kMatFloats);
    }
    const float4* p;
    if ((rc = upload(ctx, pr, &p))) return rc; S.sphere_pr = p;
  }
  {
    const float4* p;
    if ((rc = upload(ctx, mats, &p))) return rc; S.materials = p; ctx->cap_materials = mats.size() / 4;
  }
  S.n_spheres = n_spheres;
  ctx->cap_sphere_pr = (size_t)n_spheres;
  // object-level BVHs
  std::vector<float> t;
  const float4* p;
  std::vector<int32_t> cw, mesh_leaf;
  cull_words(ctx, bmt, n_meshes, mesh_root_host, cw, mesh_leaf);
  pack_tlas(bmt, t, &cw);
  ctx->walk_f4 = 0;
  {   // the masked-walk table of a small mesh heap rides behind the heap's device copy (kernels.hip front_masked)
    std::vector<float> walk;
    if (n_meshes > 0 && build_walk_table(bmt, n_meshes, mesh_root_host, small_first, walk)) {
      ctx->walk_f4 = (int)(walk.size() / 4);
      t.insert(t.end(), walk.begin(), walk.end());
    }
  }
  if ((rc = upload(ctx, t, &p))) return rc; S.mesh_tlas = p; S.n_mesh_tlas = bmt ? bmt->count : 0; ctx->cap_mesh_tlas = t.size() / 4;
  pack_tlas(bst, t);
  if ((rc = upload(ctx, t, &p))) return rc; S.sphere_tlas = p; S.n_sphere_tlas = bst ? bst->count : 0; ctx->cap_sphere_tlas = t.size() / 4;

  // traversal stack budgets (per lane, LDS)
  int lv = std::max(heap_levels(S.n_mesh_tlas), heap_levels(S.n_sphere_tlas));
  if (lv + 1 > 32)
    return fail(ctx, URT_ERR_SCENE, "object-level BVH deeper than the reference's 32-entry traversal stack (RS:73-74)");
  ctx->tlas_stack = std::max(2, lv + 1);
  if (n_blas_nodes >= (1u << 26)) return fail(ctx, URT_ERR_SCENE, "triangle BVH larger than 2^26 nodes (4 GiB)");   // kernels address nodes by 32-bit byte offsets
  if ((uint64_t)n_tris * 48ull >= (1ull << 32)) return fail(ctx, URT_ERR_SCENE, "more than 2^32 / 48 triangles (4 GiB of triangle records)");   // 32-bit byte offsets as well
  ctx->blas_stack = std::max(2, blas_max_depth + 1) + 1;      // + the sentinel entry below the stack (kernels.hip blas_node_eval_ptr)
  ctx->n_blas_nodes = (int)std::min<size_t>(0x7fffffff, n_blas_nodes);
  ctx->n_scene_tris = (int)n_tris; ctx->scene_max_depth = blas_max_depth;
  ctx->d_mesh_leaf = nullptr; ctx->cap_mesh_leaf = 0;         // (freed with the previous scene's allocations)
  if ((rc = verify_cull_flags(ctx, cw, mesh_leaf))) return rc;
  // a ray with NaN components passes every slab test and walks the whole tree once: (nodes + leaves) trips per lane, and the
  // majority vote can make a lane wait a trip for every trip it runs; 8x that is a bound no correct traversal reaches
  ctx->watchdog_steps = (unsigned int)std::min<size_t>(0x7fffffffu, 8 * (n_blas_nodes + n_tris) + 4096);
  if ((size_t)(ctx->tlas_stack + ctx->blas_stack) * 64 * 4 * sizeof(int) > 150 * 1024)   // 4-wave workgroup; a CU has 160 KiB
    return fail(ctx, URT_ERR_SCENE, "traversal stacks exceed the LDS of a compute unit");
  if (n_blas_nodes > 0) {                                     // the copy of the nodes the trace kernels traverse: child boxes as (centre, half extent)
    void* c = nullptr;
    URT_HIP(ctx, hipMalloc(&c, 4 * n_blas_nodes * sizeof(float4)));
    ctx->scene_allocs.push_back(c);
    ctx->cbuf = (float4*)c;
  }
  if (ctx->opt_qnodes != 0 && n_blas_nodes > 0) {             // 32-byte quantized nodes for the traversal loop (csrc/qnodes.hip)
    void* q = nullptr;
    URT_HIP(ctx, hipMalloc(&q, (2 + 2 * n_blas_nodes) * sizeof(float4)));
    ctx->scene_allocs.push_back(q);
    ctx->qbuf = (float4*)q;
  }
  if ((rc = rederive_nodes(ctx))) return rc;
  // what a later in-place update needs (prepare_incremental): the records this scene was prepared from, and — when it has triangle
  // BVHs — device copies of _Vertices / _Indices plus every node's parent and MeshObject (csrc/refit.hip)
  ctx->prev_mesh_objects.assign(bm ? bm->host.begin() : ctx->prev_mesh_objects.end(), bm ? bm->host.begin() + (ptrdiff_t)((size_t)n_meshes * URT_STRIDE_MESHOBJECT) : ctx->prev_mesh_objects.end());
  if (!bm) ctx->prev_mesh_objects.clear();
  ctx->h_mesh_root = mesh_root_host; ctx->h_small_first = small_first;
  if (ctx->opt_refit && n_meshes > 0 && n_tris > 0 && bv && bi) {
    urt_context::RefitAux& R = ctx->refit;
    auto dev_alloc = [&](void** ptr, size_t bytes) -> int {
      URT_HIP(ctx, hipMalloc(ptr, std::max<size_t>(bytes, 16)));
      ctx->scene_allocs.push_back(*ptr);
      return URT_OK;
    };
    if (!R.vertices) {                                        // (the GPU builder has left its copies in place)
      void *dv = nullptr, *di = nullptr;
      if ((rc = dev_alloc(&dv, (size_t)bv->count * 12))) return rc;
      if ((rc = dev_alloc(&di, (size_t)bi->count * 4))) return rc;
      URT_HIP(ctx, hipMemcpy(dv, bv->host.data(), (size_t)bv->count * 12, hipMemcpyHostToDevice));
      URT_HIP(ctx, hipMemcpy(di, bi->host.data(), (size_t)bi->count * 4, hipMemcpyHostToDevice));
      R.vertices = (const float*)dv; R.indices = (const int32_t*)di;
    }
    size_t nn = std::max<size_t>(1, n_blas_nodes);
    if ((rc = dev_alloc((void**)&R.parent, nn * 4))) return rc;
    if ((rc = dev_alloc((void**)&R.node_mesh, nn * 4))) return rc;
    if ((rc = dev_alloc((void**)&R.cbox, nn * 64))) return rc;
    if ((rc = dev_alloc((void**)&R.depth, nn * 4))) return rc;
    if ((rc = dev_alloc((void**)&R.ext, (size_t)n_meshes * 4))) return rc;
    if ((rc = dev_alloc((void**)&R.matrices, (size_t)n_meshes * 64))) return rc;
    if ((rc = dev_alloc((void**)&R.moved, (size_t)n_meshes * 4))) return rc;
    URT_HIP(ctx, refit_prepare(S.blas_nodes, (int)n_blas_nodes, S.tri_verts, R.parent, R.node_mesh, R.depth, touch(ctx)));
    R.ready = true;
  }
  ctx->scene_dirty = false; ctx->dirty_slots = 0; ctx->dirty_full = false;
  ctx->scene_epoch++;
  ctx->last_prepare_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return URT_OK;
}

int ensure_queues(urt_context* ctx, size_t n_paths, size_t n_counts) {
  if (n_paths > ctx->q_capacity) {
    for (int a = 0; a < 2; a++)
      for (int r = 0; r < 4; r++) {
        if (ctx->q.s[a][r]) (void)hipFree(ctx->q.s[a][r]);
        ctx->q.s[a][r] = nullptr;
      }
    ctx->q_capacity = 0;
    for (int a = 0; a < 2; a++)
      for (int r = 0; r < 4; r++) URT_HIP(ctx, hipMalloc((void**)&ctx->q.s[a][r], n_paths * sizeof(float4)));
    ctx->q_capacity = n_paths;
  }
  if (n_counts > ctx->counts_capacity) {
    if (ctx->q.counts) (void)hipFree(ctx->q.counts);
    ctx->q.counts = nullptr; ctx->counts_capacity = 0;
    URT_HIP(ctx, hipMalloc((void**)&ctx->q.counts, n_counts * sizeof(unsigned int)));
    ctx->counts_capacity = n_counts;
  }
  return URT_OK;
}

Texture* find_texture(urt_context* ctx, urt_handle h) {
  auto it = ctx->textures.find(h);
  return it == ctx->textures.end() ? nullptr : &it->second;
}

int resolve_timing(urt_context* ctx) {
  for (auto& pr : ctx->timing) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) ctx->trace_ms += ms;
    ctx->event_pool.push_back(pr.first);
    ctx->event_pool.push_back(pr.second);
  }
  ctx->timing.clear();
  return URT_OK;
}

int take_event(urt_context* ctx, hipEvent_t* out) {
  if (!ctx->event_pool.empty()) { *out = ctx->event_pool.back(); ctx->event_pool.pop_back(); return URT_OK; }
  URT_HIP(ctx, hipEventCreate(out));
  return URT_OK;
}

// ---- Result renaming: the slab of frame slots ---------------------------------------------------------------------------
bool in_slab(urt_context* ctx, const Texture& t) {
  return ctx->slab && t.dev >= ctx->slab && t.dev < ctx->slab + ctx->slab_stride * (size_t)ctx->slab_frames;
}

// Give a texture its own storage back (its current contents are copied out of the slab slot they live in).
int detach_from_slab(urt_context* ctx, Texture& t) {
  if (!in_slab(ctx, t)) return URT_OK;
  URT_HIP(ctx, hipMemcpyAsync(t.own, t.dev, (size_t)t.w * t.h * sizeof(float4), hipMemcpyDeviceToDevice, touch(ctx)));
  t.dev = t.own;
  return URT_OK;
}

// Slab of `frames` zero-filled slots for texture `h` (all work queued so far stays ordered before its first use: same stream).
int ensure_slab(urt_context* ctx, urt_handle h, Texture& t, int frames) {
  size_t stride = (size_t)t.w * (size_t)t.h;
  if (ctx->slab && ctx->slab_tex == h && ctx->slab_stride == stride && ctx->slab_frames >= frames) return URT_OK;
  // two slots of this size did not fit last time: not tried per frame — but again after a release in this context (free_scene,
  // urt_texture_release clear the mark) and every 256th dispatch (another context on the card may have given memory back);
  // urt_debug_launch_info reports the degradation (slab_frames_max, slab_out_of_memory)
  if (!ctx->slab && ctx->slab_oom_stride == stride && (ctx->dispatches & 255u) != 0) return URT_OK;
  if (ctx->slab_tex) {                                   // somebody's current contents may live in the old slab
    auto it = ctx->textures.find(ctx->slab_tex);
    if (it != ctx->textures.end()) { int rc = detach_from_slab(ctx, it->second); if (rc) return rc; }
    ctx->slab_tex = 0;
  }
  if (!ctx->slab || ctx->slab_stride * (size_t)ctx->slab_frames < stride * (size_t)frames) {
    if (ctx->slab) {
      URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));   // queued kernels may still use the old slab
      (void)hipFree(ctx->slab);
      ctx->slab = nullptr; ctx->slab_frames = 0; ctx->slab_stride = 0;
    }
    // out of memory (several contexts on one card, a huge image): halve the batch until the slots fit; one frame = no slab at
    // all (the caller then renders unbatched, straight into the texture)
    hipError_t e = hipErrorOutOfMemory;
    while (frames >= 2) {
      e = hipMalloc((void**)&ctx->slab, stride * (size_t)frames * sizeof(float4));
      if (e != hipErrorOutOfMemory) break;
      (void)hipGetLastError();
      ctx->slab = nullptr;
      frames /= 2;
    }
    if (e == hipErrorOutOfMemory) { ctx->slab = nullptr; ctx->slab_frames = 0; ctx->slab_stride = 0; ctx->slab_frames_max = 1; ctx->slab_oom_stride = stride; return URT_OK; }
    URT_HIP(ctx, e);
    ctx->slab_frames = frames;
    ctx->slab_frames_max = frames;
  } else {
    ctx->slab_frames = (int)(ctx->slab_stride * (size_t)ctx->slab_frames / stride);    // same bytes, re-cut for this image size
  }
  ctx->slab_stride = stride;
  if (ctx->slab_frames < 2) return URT_OK;               // (re-cut for a larger image: no room for two slots -> unbatched)
  URT_HIP(ctx, hipMemsetAsync(ctx->slab, 0, stride * (size_t)ctx->slab_frames * sizeof(float4), touch(ctx)));   // a new RenderTexture is zero-filled
  ctx->slab_tex = h;
  return URT_OK;
}

// Persistent kernels: a work-counter shard hands out RUNS of consecutive 8x8 tiles (and the waves of a workgroup share a shard),
// so neighbours on the chip work on neighbouring tiles.  Run length when "xcd_run" is 0 (auto): the largest power of two <= 8 that
// leaves every shard >= 256 runs per launch.  Measured with the frames of a launch interleaved (profiles/r02_logs/r2_run_by_launch.log, C3):
// one frame per launch 0.62 ms at 1 vs 0.82 at 64 (few runs per shard: the shards run dry unevenly); 16 frames 0.24 vs 0.33;
// 64 frames 0.221 at 4-8 vs 0.224 at 64.  (Before the interleaving, runs of 64 were the gain: r2_xcd_run.log.)
int auto_run_length(const FrameParams& P, int n_frames) {
  long runs = (long)P.tiles_x * P.n_strips * std::max(1, n_frames) / ((long)std::max(1, P.n_shards) * 256L);
  int g = 1;
  while (g < 8 && 2L * g <= runs) g *= 2;
  return g;
}

// One attempt at `groups` workgroups per CU; *degraded = an LDS feature the scene qualifies for had to be given up (or the stacks alone do not fit)
static int configure_sched_at(urt_context* ctx, const DevScene& S, FrameParams& P, bool top_in_front, size_t groups, bool* degraded) {
  // independent waves; the waves of a workgroup share one LDS copy of the top of the triangle-BVH forest, which shrinks
  // until the workgroups fit the 160 KiB of a CU next to their traversal stacks
  int t = std::min(std::min(ctx->opt_top_nodes >= 0 ? ctx->opt_top_nodes : 64, (int)kTopOrderNodes), ctx->n_blas_nodes);
  // small object-level tables (<= 256 entries) also live in LDS: their walk is a chain of dependent fetches
  P.lds_mesh = ctx->opt_lds_tlas && S.n_mesh_tlas > 0 && S.n_mesh_tlas <= 256 && S.n_meshes <= 256;
  P.lds_small = P.lds_mesh && S.n_small > 0;
  P.lds_sphere = ctx->opt_lds_tlas && S.n_sphere_tlas > 0 && S.n_sphere_tlas <= 256 && S.n_spheres <= 256;
  const bool wanted_tables = P.lds_mesh || P.lds_sphere;
  P.top_nodes = t;
  const size_t budget = 156 * 1024;                          // (a little of the 160 KiB goes to allocation granules)
  while (P.top_nodes > 0 && sched_lds_bytes(S, P) * groups > budget) P.top_nodes /= 2;
  if (ctx->opt_top_nodes < 0 && P.top_nodes == 64) {
    // auto: what is left of the workgroup's share of the LDS holds more of the forest's top, 16 nodes (1 KiB) at a time — C3 / C3D: 96 nodes,
    // -0.6 % per frame in 64-frame launches (profiles/r04_logs/r4_sweep_top_nodes.log); every node there is a fetch at LDS latency
    const int most = std::min((int)kTopOrderNodes, ctx->n_blas_nodes);
    while (P.top_nodes + 16 <= most) {
      P.top_nodes += 16;
      if (sched_lds_bytes(S, P) * groups > budget) { P.top_nodes -= 16; break; }
    }
  }
  if (sched_lds_bytes(S, P) * groups > budget) { P.lds_mesh = 0; P.lds_sphere = 0; P.lds_small = 0; }
  *degraded = (t > 0 && P.top_nodes == 0) || (wanted_tables && !(P.lds_mesh || P.lds_sphere)) || sched_lds_bytes(S, P) * groups > budget;
  // listed FRONT (kernels.hip front_listed): scenes of a few MeshObjects whose heap is in LDS; the list of objects a ray has to test
  // (<= 12 ids of 5 bits) lives in the first two entries of the lane's object-level stack, so it costs no LDS
  bool listed = top_in_front && P.top_nodes > 0 && P.lds_mesh && S.n_meshes <= 12 && ctx->opt_front_list != 0;
  // masked FRONT (kernels.hip front_masked): mesh heaps of <= 31 nodes are walked with mask arithmetic instead of a stack; the walk
  // table takes the heap's place in LDS.  "front_list" 2 forces the listed form (A/B), -1 / 1 prefer the masked one.
  bool masked = top_in_front && t > 0 && ctx->opt_lds_tlas && ctx->walk_f4 > 0 && !P.serve && ctx->opt_front_list != 0 && ctx->opt_front_list != 2;
  if (masked) {
    // The masked walk keeps no object-level stack for the mesh heap: the lane's `tl` column only serves the sphere heap's walk.  The
    // entries that frees (C4, C5: 4 of 6, i.e. 4 KiB per workgroup) go to the LDS copy of the top of the forest, which is sized again for this layout.
    FrameParams Q = P;
    Q.lds_mesh = 0; Q.walk_f4 = ctx->walk_f4; Q.lds_small = S.n_small > 0;
    Q.tlas_stack = std::max(2, heap_levels(S.n_sphere_tlas) + 1);
    Q.top_nodes = t;
    if (ctx->opt_top_nodes < 0) {
      // measured (profiles/r03_logs/r3_sweep_top_masked.log): C4 (3 big MeshObjects) 2.97 / 2.99 / 3.01 / 3.06 ms at a top of 4 / 8 / 16 / 64 nodes,
      // C5 (12) 1.57 / 1.52 / 1.50 / 1.495 / 1.50 at 4 / 8 / 16 / 32 / 64: the roots and about one more level
      int big = 0;
      for (int32_t r : ctx->h_mesh_root) big += r >= 0 && r != kEmptyMeshRoot;
      int want = 4;
      while (want < 2 * big && want < 64) want *= 2;
      Q.top_nodes = std::min(t, want);
    }
    while (Q.top_nodes > 0 && sched_lds_bytes(S, Q) * groups > budget) Q.top_nodes /= 2;
    if (Q.top_nodes > 0 && sched_lds_bytes(S, Q) * groups <= budget) { P = Q; *degraded = false; return 3; }
    *degraded = true;
  }
  return listed ? 2 : (top_in_front && P.top_nodes > 0) ? 1 : 0;
}

// kernel_mode 3: what lives in the workgroup's LDS next to the stacks (fills P.top_nodes, P.lds_*, P.block_threads, P.list_base,
// P.tlas_stack), how many workgroups per CU the launch counts on (ctx->sched_groups) and how FRONT treats MeshObjects (returns the
// front mode of kernels.h launch_sched).  5 waves per SIMD (what 96 VGPRs allow) = 5 workgroups of 4 waves per CU is the target; a scene
// whose traversal stacks are too deep for that (a GPU-built Morton tree of 100 k triangles is 30 levels: 34 KiB of stacks per workgroup)
// keeps its LDS features — the masked object-level phase, the tables, the top of the forest — at 4 or 3 workgroups per CU instead
// of losing them at a nominal 5 that the hardware would not make resident anyway (GPU-built trees: C4 6.71 -> 3.61 ms, C3 0.284 -> 0.264, C3D 0.502 -> 0.443; profiles/r03_logs/r3_lbvh_groups.log).
int configure_sched(urt_context* ctx, const DevScene& S, FrameParams& P, bool top_in_front) {
  {
    const int t = std::min(std::min(ctx->opt_top_nodes >= 0 ? ctx->opt_top_nodes : 64, (int)kTopOrderNodes), ctx->n_blas_nodes);
    const bool lds_mesh = ctx->opt_lds_tlas && S.n_mesh_tlas > 0 && S.n_mesh_tlas <= 256 && S.n_meshes <= 256;
    const bool lds_sphere = ctx->opt_lds_tlas && S.n_sphere_tlas > 0 && S.n_sphere_tlas <= 256 && S.n_spheres <= 256;
    const bool shared = t > 0 || lds_mesh || lds_sphere;
    P.block_threads = ctx->opt_sched_block > 0 ? ctx->opt_sched_block : (shared ? 256 : 64);   // nothing to share: single waves
    if (P.serve) P.block_threads = 256;                        // kernel_mode 5: the waves of a workgroup share the traversal service
  }
  const int wpc_default = P.serve ? 16 : 20;                 // what the kernel's registers allow (k_serve: 128 VGPRs, k_sched: 96)
  const size_t per = (size_t)(P.block_threads / 64);
  const size_t groups = std::max<size_t>(1, (size_t)(ctx->opt_waves_per_cu > 0 ? ctx->opt_waves_per_cu : wpc_default) / per);   // workgroups per CU that should fit
  ctx->sched_groups = 0;
  const FrameParams P0 = P;
  bool degraded = false;
  int mode = configure_sched_at(ctx, S, P, top_in_front, groups, &degraded);
  if (degraded && ctx->opt_waves_per_cu <= 0 && per == 4) {
    for (size_t g = groups - 1; g >= 3 && g + 2 >= groups; g--) {
      FrameParams Q = P0; bool d2 = false;
      int m2 = configure_sched_at(ctx, S, Q, top_in_front, g, &d2);
      if (!d2) { P = Q; mode = m2; ctx->sched_groups = (int)g; break; }
    }
  }
  return mode;
}

// What urt_debug_launch_info reports: taken right after a launcher of kernels.hip returned (one host thread per context)
void record_launch(urt_context* ctx, int kernel_mode, int front_mode, const FrameParams& P, bool count, int waves_per_cu) {
  urt_launch_info& I = ctx->last_launch;
  const TraceLaunchRecord& R = last_trace_launch();
  std::memset(&I, 0, sizeof I);
  std::snprintf(I.kernel, sizeof I.kernel, "%s", R.kernel);
  I.kernel_mode = kernel_mode; I.front_mode = front_mode; I.count_stats = count ? 1 : 0;
  I.n_blocks = R.n_blocks; I.block_threads = R.block_threads; I.lds_bytes = R.lds_bytes;
  I.n_frames = P.n_frames; I.frame_group = P.frame_group; I.xcd_run = P.xcd_run; I.tile_order = P.tile_order;
  I.top_nodes = P.top_nodes; I.waves_per_cu = waves_per_cu;
  I.tlas_stack = P.tlas_stack; I.blas_stack = P.blas_stack;
  I.lds_tables = (P.lds_mesh ? 1 : 0) | (P.lds_sphere ? 2 : 0) | (P.lds_small ? 4 : 0) | (P.walk_f4 > 0 ? 8 : 0);
  I.slab_frames = ctx->slab_frames; I.slab_frames_max = ctx->slab_frames_max; I.slab_out_of_memory = ctx->slab_oom_stride != 0 ? 1 : 0;
  I.experiment = URT_ABI_SIGN < 0 ? 1 : 0;
}

// Launch the phase-scheduled trace kernel for P.n_frames frames (uniforms T) into result + f * P.frame_stride.
static constexpr int kAutoFrames = 64;     // frames per launch when "frames_per_launch" is 0 (auto) on the library's own stream

int launch_sched_frames(urt_context* ctx, const DevScene& S, const FrameParams& P, const FrameTable& T, float4* result,
                        int front_mode, bool count, hipStream_t st = nullptr, unsigned int* next = nullptr) {
  if (!st) { st = touch(ctx); next = ctx->d_next; }       // the main stream; flush_pending may pass one of its trace streams and that stream's work counters
  // the launch's frame table -> device memory, in stream order (pinned staging slot: the copy does not wait for the stream)
  if (!ctx->h_tables) {
    URT_HIP(ctx, hipHostMalloc((void**)&ctx->h_tables, sizeof(FrameUniforms) * kMaxFramesPerLaunch * urt_context::kTableSlots, hipHostMallocDefault));
    URT_HIP(ctx, hipMalloc((void**)&ctx->d_tables, sizeof(FrameUniforms) * kMaxFramesPerLaunch * urt_context::kTableSlots));
    for (int k = 0; k < urt_context::kTableSlots; k++) URT_HIP(ctx, hipEventCreateWithFlags(&ctx->table_ev[k], hipEventDisableTiming));
  }
  const unsigned int slot = ctx->table_next++ % (unsigned int)urt_context::kTableSlots;
  if (ctx->table_next > (unsigned int)urt_context::kTableSlots) URT_HIP(ctx, hipEventSynchronize(ctx->table_ev[slot]));   // (four launches ago: long done)
  FrameUniforms* h_slot = ctx->h_tables + (size_t)slot * kMaxFramesPerLaunch;
  FrameUniforms* d_table = ctx->d_tables + (size_t)slot * kMaxFramesPerLaunch;
  std::memcpy(h_slot, T.f, sizeof(FrameUniforms) * (size_t)P.n_frames);
  URT_HIP(ctx, hipMemcpyAsync(d_table, h_slot, sizeof(FrameUniforms) * (size_t)P.n_frames, hipMemcpyHostToDevice, st));
  URT_HIP(ctx, hipEventRecord(ctx->table_ev[slot], st));
  int waves_per_block = P.block_threads / 64;
  long want = ((long)P.tiles_x * P.n_strips * P.n_frames + waves_per_block - 1) / waves_per_block;
  // resident waves per CU: every slot the registers allow (k_sched: 96 VGPRs -> 5 waves/SIMD = 20 per CU).  While the
  // frame's work counter was one address, fewer and fatter waves were faster at 1080p (12 per CU); since it is sharded
  // (kernels.hip wave_fetch_pixels) the full 20 win at every frame size measured (profiles/README.md).
  int wpc = ctx->opt_waves_per_cu;
  if (wpc <= 0) wpc = P.serve ? 16 : 20;
  if (ctx->sched_groups > 0) wpc = ctx->sched_groups * waves_per_block;       // deep stacks: fewer workgroups per CU, LDS features kept (configure_sched)
  long resident = (long)ctx->n_cus * wpc / waves_per_block;
  int nb = (int)std::max(1L, std::min(want, resident));
  if (P.serve) {                                             // mailbox of the posted rays: 32 B per thread of the grid
    size_t slots = (size_t)nb * (size_t)P.block_threads;
    if (slots > ctx->mail_slots) {
      if (ctx->d_mail) { URT_HIP(ctx, hipStreamSynchronize(touch(ctx))); (void)hipFree(ctx->d_mail); ctx->d_mail = nullptr; ctx->mail_slots = 0; }
      URT_HIP(ctx, hipMalloc((void**)&ctx->d_mail, slots * 2 * sizeof(float4)));
      ctx->mail_slots = slots;
    }
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->opt_time_dispatch) {
    int rc = take_event(ctx, &e0); if (rc) return rc;
    rc = take_event(ctx, &e1); if (rc) return rc;
    URT_HIP(ctx, hipEventRecord(e0, st));
  }
  hipError_t le = P.serve ? launch_serve(S, P, d_table, result, ctx->d_counters, next, ctx->d_mail, nb, front_mode, count, st)
                          : launch_sched(S, P, d_table, result, ctx->d_counters, next, nb, front_mode, count, st);
  if (ctx->opt_time_dispatch) {
    (void)hipEventRecord(e1, st);
    ctx->timing.emplace_back(e0, e1);
  }
  ctx->launches++;
  if (le != hipSuccess) return fail(ctx, URT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(le));
  record_launch(ctx, P.serve ? 5 : 3, front_mode, P, count, wpc);
  return URT_OK;
}

// Submit the deferred frames: ONE trace launch, then the deferred blits in program order.  Runs of AdditionShader blits of
// consecutive frames into one image are fused into a single pass (same per-pixel operations in the same order).
int flush_pending(urt_context* ctx) {
  urt_context::Pending& B = ctx->pend;
  if (B.n == 0) return URT_OK;
  int n = B.n;
  B.n = 0;                                               // whatever happens below, the batch is gone
  std::vector<urt_context::PostOp> ops;
  ops.swap(B.ops);
  URT_HIP(ctx, hipSetDevice(ctx->device));
  FrameParams P = B.P;
  P.n_frames = n;
  P.sched_trips = sched_trip_cap(ctx, P, n);
  P.frame_group = std::max(1, std::min(P.frame_group, n));
  if (ctx->opt_xcd_run <= 0) P.xcd_run = auto_run_length(P, n);
  P.frame_stride = (unsigned int)ctx->slab_stride;
  // Small launches (a host that presents every frame) overlap: see urt_context "Overlapped launches".  Launch L goes to trace stream
  // L mod 2 and takes slots [base, base + n) round-robin; it waits for
  //   - pre_ev of launch L-1: everything the main stream held when L-1 was submitted — the blends / presents of L-2 and older (the last
  //     readers of any slot L may reuse), scene uploads, texture writes — but NOT launch L-1 itself nor its blends, whose slots are others;
  //   - or, when anything but the frame loop's own work went to the main stream since (main_touched: SetData, a scene preparation, a
  //     blit outside a batch, a gather, ...) or the slots would collide, for the main stream as it is now — which has waited for L-1.
  // The main stream waits for the launch before its deferred blits, so "the main stream is idle" still means "everything is done".
  int base = 0;
  bool reading = false;                                  // a pipelined readback in flight: the host paces itself on FINISHED frames, and two launches sharing
  for (const auto& r : ctx->rslot) reading = reading || r.busy;   // the chip finish later than one after the other (measured: +6 % C3, +21 % C2 with two tickets in flight)
  const bool eligible = (ctx->opt_overlap == 2 || (ctx->opt_overlap == 1 && !reading)) && ctx->stream == ctx->own_stream && !P.serve && !ctx->opt_time_dispatch &&
                        n <= urt_context::kOverlapFrames && ctx->slab_frames >= 2 * urt_context::kOverlapFrames && ctx->d_next2;
  if (eligible) {
    base = ctx->slab_cursor + n <= ctx->slab_frames ? ctx->slab_cursor : 0;
    if (!ctx->trace_q[0]) {
      for (int k = 0; k < 2; k++) {
        URT_HIP(ctx, hipStreamCreateWithFlags(&ctx->trace_q[k], hipStreamNonBlocking));
        URT_HIP(ctx, hipEventCreateWithFlags(&ctx->trace_done[k], hipEventDisableTiming));
        URT_HIP(ctx, hipEventCreateWithFlags(&ctx->pre_ev[k], hipEventDisableTiming));
      }
      URT_HIP(ctx, hipEventCreateWithFlags(&ctx->dep_ev, hipEventDisableTiming));
    }
    const unsigned int k = ctx->trace_parity++ & 1u;
    const bool disjoint = base >= ctx->prev_base + ctx->prev_n || base + n <= ctx->prev_base;
    if (ctx->main_touched || !disjoint) {
      URT_HIP(ctx, hipEventRecord(ctx->dep_ev, ctx->stream));
      URT_HIP(ctx, hipStreamWaitEvent(ctx->trace_q[k], ctx->dep_ev, 0));
    } else {
      URT_HIP(ctx, hipStreamWaitEvent(ctx->trace_q[k], ctx->pre_ev[k ^ 1u], 0));
      ctx->overlapped_launches++;
    }
    const bool narrow = !(ctx->main_touched || !disjoint);
    int rc = launch_sched_frames(ctx, B.S, P, B.T, ctx->slab + (size_t)base * ctx->slab_stride, B.front_mode, B.count, ctx->trace_q[k], k ? ctx->d_next2 : ctx->d_next);
    if (rc) { ctx->main_touched = true; return rc; }
    ctx->last_launch.trace_stream = 1 + (int)k; ctx->last_launch.slab_base = base; ctx->last_launch.overlapped = narrow ? 1 : 0;
    URT_HIP(ctx, hipEventRecord(ctx->trace_done[k], ctx->trace_q[k]));
    URT_HIP(ctx, hipEventRecord(ctx->pre_ev[k], ctx->stream));
    URT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->trace_done[k], 0));
    ctx->main_touched = false;
  } else {
    int rc = launch_sched_frames(ctx, B.S, P, B.T, ctx->slab, B.front_mode, B.count);     // on the main stream (marks it touched)
    if (rc) return rc;
  }
  ctx->prev_base = base; ctx->prev_n = n; ctx->slab_cursor = base + n;
  if (base) {                                            // the Result texture names the LAST frame's slot (do_dispatch named it assuming slot 0)
    Texture* rt = find_texture(ctx, B.tex);
    if (rt && in_slab(ctx, *rt)) rt->dev = ctx->slab + (size_t)(base + n - 1) * ctx->slab_stride;
  }
  const float4* const slots = ctx->slab + (size_t)base * ctx->slab_stride;
  size_t i = 0;
  while (i < ops.size()) {
    const urt_context::PostOp& op = ops[i];
    if (op.kind == 0) {
      Texture* d = find_texture(ctx, op.dst);
      if (!d) return fail(ctx, URT_ERR_INVALID_HANDLE, "deferred Blit: destination texture was released");
      // A run of AdditionShader blends of consecutive frames into one image, each possibly followed by the present of that image
      // (RM:818-819: Blit(_target, _converged, mat); Blit(_converged, destination)) — ONE pass.  Of the presents only the last
      // is observable: every call that could observe `destination` submits this work first, so an earlier present is overwritten
      // unseen (as-if rule, include/urt.h); the run ends at its last present so that the image presented is the mean it was
      // presented with.
      float samples[kMaxFramesPerLaunch];
      size_t j = i, j_present = i;
      int cnt = 0, cnt_present = 0;
      urt_handle present = 0;
      while (j < ops.size() && cnt <= kMaxFramesPerLaunch) {
        const urt_context::PostOp& q = ops[j];
        if (q.kind == 0 && q.dst == op.dst && q.frame == op.frame + cnt && cnt < kMaxFramesPerLaunch) { samples[cnt++] = q.sample; j++; }
        else if (q.kind == 2 && q.tex == op.dst && q.dst != op.dst && q.dst != B.tex && (present == 0 || q.dst == present)) {
          present = q.dst; j++; j_present = j; cnt_present = cnt;
        } else break;
      }
      if (present) { j = j_present; cnt = cnt_present; }
      float4* pdev = nullptr;
      if (present) {
        Texture* pt = find_texture(ctx, present);
        if (!pt) return fail(ctx, URT_ERR_INVALID_HANDLE, "deferred Blit: destination texture was released");
        pdev = pt->dev;
      }
      const float4* src = slots + (size_t)op.frame * ctx->slab_stride;
      hipError_t e = (cnt == 1 && !pdev) ? launch_blit_add(src, d->dev, (size_t)d->w * d->h, samples[0], ctx->stream)
                                         : launch_blit_add_multi(src, ctx->slab_stride, cnt, samples, d->dev, pdev, (size_t)d->w * d->h, ctx->stream);
      if (e != hipSuccess) return fail(ctx, URT_ERR_HIP, std::string("deferred Blit: ") + hipGetErrorString(e));
      i = j;
    } else if (op.kind == 2) {
      Texture* t = find_texture(ctx, op.tex);
      Texture* d = find_texture(ctx, op.dst);
      if (!t || !d) return fail(ctx, URT_ERR_INVALID_HANDLE, "deferred Blit: texture was released");
      const float4* img = op.tex == B.tex ? slots + (size_t)op.frame * ctx->slab_stride : t->dev;
      URT_HIP(ctx, hipMemcpyAsync(d->dev, img, (size_t)t->w * t->h * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
      i++;
    } else {
      Texture* t = find_texture(ctx, op.tex);
      if (!t) return fail(ctx, URT_ERR_INVALID_HANDLE, "deferred pack_rows: texture was released");
      const float4* img = op.tex == B.tex ? slots + (size_t)op.frame * ctx->slab_stride : t->dev;
      int group_rows = (t->h + 7) / 8;
      int n_strips = op.first_row < group_rows ? (group_rows - op.first_row + op.row_stride - 1) / op.row_stride : 0;
      hipError_t e = op.sample != 0.0f ? launch_pack_rows_rgb(const_cast<float4*>(img), (float*)op.dense, t->w, t->h, op.first_row, op.row_stride, n_strips, true, 0.0f, ctx->stream)
                                       : launch_pack_rows(const_cast<float4*>(img), (float4*)op.dense, t->w, t->h, op.first_row, op.row_stride, n_strips, true, ctx->stream);
      if (e != hipSuccess) return fail(ctx, URT_ERR_HIP, std::string("deferred pack_rows: ") + hipGetErrorString(e));
      i++;
    }
  }
  return URT_OK;
}

// frames one launch may hold for this dispatch
int batch_limit(urt_context* ctx, const FrameParams& P) {
  int lim = ctx->opt_frames_per_launch;
  if (lim == 0) {
    if (ctx->stream != ctx->own_stream) return 1;        // a caller that shares its stream expects the work ON the stream when dispatch returns
    // kAutoFrames frames per launch, within 8 GiB of Result slots: 2160p still gains from long launches (profiles/r02_logs/r2_fpl4k.log),
    // and 32 x 133 MB is nothing on a 288 GB part
    uint64_t frame_bytes = (uint64_t)P.width * (uint64_t)P.height * sizeof(float4);
    lim = (int)std::min<uint64_t>(kAutoFrames, std::max<uint64_t>(1, (8ull << 30) / std::max<uint64_t>(1, frame_bytes)));
  }
  // the work counter hands out 32-bit pixel slots: frames x tiles x 64 must stay below 2^32
  uint64_t slots = std::max<uint64_t>(1, ((uint64_t)P.tiles_x * (uint64_t)P.n_strips + (uint64_t)std::max(64, ctx->opt_xcd_run)) * 64u);   // (a frame's last run is padded when frames are interleaved)
  lim = (int)std::min<uint64_t>((uint64_t)lim, std::max<uint64_t>(1, 0xfffffffeull / slots / 2));
  return std::max(1, std::min(lim, (int)kMaxFramesPerLaunch));
}

int do_dispatch(urt_context* ctx, int kernel, int gx, int gy, int gz, int first_row, int row_stride) {
  if (kernel != 0) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "kernel index must be 0 (CSMain)");
  if (gx < 0 || gy < 0 || gz < 0 || first_row < 0 || row_stride < 1)
    return fail(ctx, URT_ERR_INVALID_ARGUMENT, "negative thread-group count or bad strip arguments");
  urt_handle res_h = ctx->t_result;
  Texture* res = find_texture(ctx, res_h);
  if (!res) return fail(ctx, URT_ERR_UNBOUND, "Dispatch: no texture bound to \"Result\" (RM:803)");
  if (res->w > 65535 || res->h > 65535) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "Result larger than 65535 pixels per side");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->scene_dirty) {
    int rc = flush_pending(ctx); if (rc) return rc;      // the deferred frames read the scene that is about to be replaced
    rc = prepare_scene(ctx); if (rc) return rc;
  }
  ctx->dispatches++;
  if (gx == 0 || gy == 0 || gz == 0) return URT_OK;

  DevScene S = ctx->ds;
  Texture* sky = find_texture(ctx, ctx->t_sky);
  if (sky) { S.sky = sky->dev; S.sky_w = sky->w; S.sky_h = sky->h; }
  else {     // an unbound SRV reads zeros
    if (!ctx->zero_sky) {
      URT_HIP(ctx, hipMalloc((void**)&ctx->zero_sky, sizeof(float4)));
      URT_HIP(ctx, hipMemsetAsync(ctx->zero_sky, 0, sizeof(float4), touch(ctx)));
    }
    S.sky = ctx->zero_sky; S.sky_w = 1; S.sky_h = 1;
  }

  FrameParams P{};
  std::memcpy(P.c2w, ctx->c2w, sizeof P.c2w);
  std::memcpy(P.invp, ctx->invp, sizeof P.invp);
  P.pixel_off_x = ctx->pixel_off[0]; P.pixel_off_y = ctx->pixel_off[1];
  P.seed = ctx->seed;
  P.num_bounces = ctx->num_bounces; P.num_rays = ctx->num_rays;
  P.width = res->w; P.height = res->h;
  long rw = std::min<long>((long)gx * 8, res->w), rh = std::min<long>((long)gy * 8, res->h);
  P.region_w = (int)rw; P.region_h = (int)rh;
  P.tiles_x = (P.region_w + 7) / 8;
  int group_rows = (P.region_h + 7) / 8;
  P.first_group_row = first_row; P.row_stride = row_stride;
  P.n_strips = first_row < group_rows ? (group_rows - first_row + row_stride - 1) / row_stride : 0;
  P.tlas_stack = ctx->tlas_stack; P.blas_stack = ctx->blas_stack + ctx->opt_stack_pad; P.watchdog_steps = ctx->watchdog_steps;
  P.block_threads = ctx->opt_block_threads; P.xcd_run = ctx->opt_xcd_run; P.tile_order = ctx->opt_tile_order >= 0 ? ctx->opt_tile_order : (S.n_meshes == 0 ? 1 : 0); P.refill_min = ctx->opt_refill_min;
  // lanes parked at a triangle BVH before the traversal phase runs: 16 with one mesh (C3 -2 %, C3D -6 % against 28), 24 when rays walk
  // several (C4, C5 -1 %) — re-measured after the work distribution became local (profiles/r02_logs/r2_blas_min.log)
  P.blas_min = ctx->opt_blas_min > 0 ? ctx->opt_blas_min : (S.n_meshes > 1 ? 24 : 16);
  // the traversal phase yields when fewer lanes than this are still traversing: measured best 14-18 with one mesh, 8-11 when rays
  // walk several triangle BVHs per Trace() (a yielding lane then continues its object-level walk sooner)
  P.blas_exit = ctx->opt_blas_exit > 0 ? ctx->opt_blas_exit : (S.n_meshes > 1 ? 9 : 14); P.shade_min = ctx->opt_shade_min; P.sky_min = ctx->opt_sky_min;
  P.n_frames = 1; P.frame_stride = 0;
  P.sched_trips = sched_trip_cap(ctx, P, 1); P.trip_flag = ctx->d_trip_flag;
  if (P.n_strips == 0 || P.tiles_x == 0) return URT_OK;
  P.n_shards = ctx->opt_work_shards; P.frame_group = ctx->opt_frame_group;
  if (ctx->opt_xcd_run <= 0) P.xcd_run = ctx->opt_kernel_mode >= 2 ? auto_run_length(P, 1) : 1;   // (batched launches: again at submission, with the launch's frame count)
  if ((uint64_t)P.tiles_x * (uint64_t)P.n_strips * 64u >= 0xffffffffull)
    return fail(ctx, URT_ERR_INVALID_ARGUMENT, "Dispatch: too many pixel slots in one dispatch");

  // region pixels this dispatch writes (threads outside Result write nothing, RS:468)
  {
    uint64_t px = 0;
    for (int j = 0; j < P.n_strips; j++) {
      int y0 = (first_row + j * row_stride) * 8;
      px += (uint64_t)std::max(0, std::min(P.region_h - y0, 8)) * (uint64_t)P.region_w;
    }
    ctx->pixels_dispatched += px;
  }
  bool degenerate = P.num_bounces <= 0 || P.num_rays <= 0;      // loops that never run: the megakernel handles them literally
  int mode = degenerate ? 0 : ctx->opt_kernel_mode;
  if ((mode == 3 || mode == 5) && P.num_bounces >= (1 << 24)) mode = 2;        // k_sched / k_serve keep the bounce index in 24 bits
  bool count = ctx->opt_count_stats != 0;
  const int region[4] = {P.region_w, P.region_h, first_row, row_stride};
  const bool full_cover = P.region_w == res->w && P.region_h == res->h && first_row == 0 && row_stride == 1;

  if (mode == 3 || mode == 5) {
    bool top_in_front = ctx->opt_top_front < 0 ? S.n_meshes > 1 : ctx->opt_top_front != 0;
    P.serve = mode == 5 && ctx->n_blas_nodes > 0;            // no triangle BVH, nothing to serve: mode 3's kernel
    P.pool_inloop = ctx->opt_serve_refill;
    int front_mode = configure_sched(ctx, S, P, top_in_front);
    P.shade_split = ctx->opt_shade_split != 0;
    FrameUniforms fu{};
    std::memcpy(fu.c2w, P.c2w, sizeof fu.c2w);
    std::memcpy(fu.invp, P.invp, sizeof fu.invp);
    fu.pixel_off_x = P.pixel_off_x; fu.pixel_off_y = P.pixel_off_y; fu.seed = P.seed;
    // May this dispatch be renamed to a fresh slab slot?  Its unwritten pixels must read as before: none (full cover), or
    // still the zeros of creation (only dispatches of this same region ever wrote the image).
    bool same_region = res->n_regions == 1 && std::memcmp(res->rg, region, sizeof region) == 0;
    bool renamable = !res->external && !res->ptr_exposed && (full_cover || (!res->other_writes && (res->n_regions == 0 || same_region)));
    int limit = renamable ? batch_limit(ctx, P) : 1;
    urt_context::Pending& B = ctx->pend;
    if (B.n > 0) {
      const FrameParams& Q = B.P;
      bool same = B.tex == res_h && B.scene_epoch == ctx->scene_epoch && B.S.sky == S.sky && B.S.sky_w == S.sky_w && B.S.sky_h == S.sky_h &&
                  B.count == count && B.front_mode == front_mode && Q.serve == P.serve && Q.num_bounces == P.num_bounces && Q.num_rays == P.num_rays &&
                  Q.width == P.width && Q.height == P.height && Q.region_w == P.region_w && Q.region_h == P.region_h &&
                  Q.first_group_row == P.first_group_row && Q.row_stride == P.row_stride && B.n < B.limit && limit > 1;
      if (!same) { int rc = flush_pending(ctx); if (rc) return rc; }
    }
    if (limit > 1 && B.n == 0) {                          // a new batch: its Result slots (fewer, or none, when memory is short)
      int rc = ensure_slab(ctx, res_h, *res, limit); if (rc) return rc;
      if (!ctx->slab || ctx->slab_tex != res_h || ctx->slab_frames < 2) limit = 1;
    }
    if (limit <= 1) {                                     // not batched: trace this frame now, straight into the texture
      FrameTable T{};
      T.f[0] = fu;
      P.frame_group = 1;
      int rc = launch_sched_frames(ctx, S, P, T, res->dev, front_mode, count);
      if (rc) return rc;
    } else {
      if (B.n == 0) {
        B.limit = std::min(limit, ctx->slab_frames);
        B.tex = res_h; B.scene_epoch = ctx->scene_epoch; B.S = S; B.P = P; B.front_mode = front_mode; B.count = count;
      }
      B.T.f[B.n] = fu;
      res->dev = ctx->slab + (size_t)B.n * ctx->slab_stride;   // Result now names this frame's slot
      B.n++;
    }
  } else {
    int rc = flush_pending(ctx); if (rc) return rc;
    if (mode == 1) {
      size_t n_paths = (size_t)P.tiles_x * 64 * (size_t)P.n_strips;
      rc = ensure_queues(ctx, n_paths, (size_t)P.num_rays * (size_t)(P.num_bounces + 1));
      if (rc) return rc;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->opt_time_dispatch) {
      rc = take_event(ctx, &e0); if (rc) return rc;
      rc = take_event(ctx, &e1); if (rc) return rc;
      URT_HIP(ctx, hipEventRecord(e0, touch(ctx)));
    }
    hipError_t le;
    if (mode == 1) le = launch_wavefront(S, P, ctx->q, res->dev, ctx->d_counters, count, touch(ctx));
    else if (mode == 2) {
      int waves_per_block = P.block_threads / 64;
      long want = ((long)P.tiles_x * P.n_strips + waves_per_block - 1) / waves_per_block;
      int wpc = ctx->opt_waves_per_cu;
      if (wpc <= 0) wpc = 20;
      long resident = (long)ctx->n_cus * wpc / waves_per_block;
      int nb = (int)std::max(1L, std::min(want, resident));
      le = launch_persist(S, P, res->dev, ctx->d_counters, ctx->d_next, nb, count, touch(ctx));
    } else if (mode == 4) {
      // one wave per workgroup; residency is bounded by the LDS one wave's path pool takes (kernels.hip k_pool)
      P.block_threads = 64;
      P.refill_min = ctx->opt_pool_refill; P.blas_min = ctx->opt_pool_blas_min; P.blas_exit = ctx->opt_pool_blas_exit;
      P.pool_inloop = ctx->opt_pool_inloop; P.pool_other_min = ctx->opt_pool_other_min;
      int k = ctx->opt_pool_k;
      size_t lds = pool_lds_bytes(P, k);
      while (k > 1 && lds > 160 * 1024) { k--; lds = pool_lds_bytes(P, k); }
      if (lds > 160 * 1024) return fail(ctx, URT_ERR_OUT_OF_MEMORY, "kernel_mode 4: the scene's traversal stacks do not fit the LDS of one CU; use kernel_mode 3");
      int fit = (int)std::max<size_t>(1, (160 * 1024) / lds);
      int wpc = ctx->opt_waves_per_cu > 0 ? ctx->opt_waves_per_cu : fit;
      long want = ((long)P.tiles_x * P.n_strips * 64 + 64L * k - 1) / (64L * k);
      int nb = (int)std::max(1L, std::min(want, (long)ctx->n_cus * wpc));
      le = launch_pool(S, P, res->dev, ctx->d_counters, ctx->d_next, nb, k, count, touch(ctx));
    } else le = launch_mega(S, P, res->dev, ctx->d_counters, count, touch(ctx));
    if (ctx->opt_time_dispatch) {
      (void)hipEventRecord(e1, touch(ctx));
      ctx->timing.emplace_back(e0, e1);
    }
    ctx->launches++;
    if (le != hipSuccess) return fail(ctx, URT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(le));
    record_launch(ctx, mode, 0, P, count, ctx->opt_waves_per_cu);
  }
  // remember what has written the image (see Texture)
  if (res->n_regions == 0) { res->n_regions = 1; std::memcpy(res->rg, region, sizeof region); }
  else if (std::memcmp(res->rg, region, sizeof region) != 0) res->n_regions = 2;
  return URT_OK;
}

}  // namespace

extern "C" {

int urt_abi_version(void) { return URT_ABI_SIGN * 4; }   // negative: an A/B / probe / diagnostic build (csrc/experiments.h), refused by loaders that did not opt in

int urt_device_count(int* out_count) {
  if (!out_count) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "out_count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *out_count = 0; return fail(nullptr, URT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  *out_count = n;
  return URT_OK;
}

int urt_context_create(int device, urt_context** out_ctx) {
  if (!out_ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
  *out_ctx = nullptr;
  URT_GUARD_BEGIN
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, URT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= n) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  URT_HIP(nullptr, hipSetDevice(device));
  urt_context* ctx = new urt_context();
  ctx->device = device;
  e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete ctx; return fail(nullptr, URT_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
  ctx->stream = ctx->own_stream;
  e = hipMalloc((void**)&ctx->d_counters, sizeof(DevCounters) * kCounterShards);
  if (e == hipSuccess) e = hipMemset(ctx->d_counters, 0, sizeof(DevCounters) * kCounterShards);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_next, kWorkShards * 128 + 65536 * 16 * sizeof(unsigned long long));   // work-counter shards; the rest: diagnostic stamps (URT_STAMPS builds)
  if (e == hipSuccess) e = hipMemset(ctx->d_next, 0, kWorkShards * 128 + 65536 * 16 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_next2, kWorkShards * 128 + 65536 * 16 * sizeof(unsigned long long));    // the launch on the second trace stream (same layout)
  if (e == hipSuccess) e = hipMemset(ctx->d_next2, 0, kWorkShards * 128 + 65536 * 16 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_trip_flag, 64, hipHostMallocMapped | hipHostMallocCoherent);   // the watchdog word the kernels raise (system-scope atomic)
  if (e == hipSuccess) { *ctx->h_trip_flag = 0; e = hipHostGetDevicePointer((void**)&ctx->d_trip_flag, ctx->h_trip_flag, 0); }
  if (e == hipSuccess) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) ctx->n_cus = n; }
  if (e != hipSuccess) {
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_next) (void)hipFree(ctx->d_next);
    if (ctx->d_next2) (void)hipFree(ctx->d_next2);
    if (ctx->h_trip_flag) (void)hipHostFree(ctx->h_trip_flag);
    (void)hipStreamDestroy(ctx->own_stream); delete ctx;
    return fail(nullptr, URT_ERR_HIP, std::string("counter allocation: ") + hipGetErrorString(e));
  }
  *out_ctx = ctx;
  return URT_OK;
  URT_GUARD_END(nullptr)
}

int urt_context_destroy(urt_context* ctx) {
  if (!ctx) return URT_OK;
  (void)hipSetDevice(ctx->device);
  (void)flush_pending(ctx);
  (void)hipStreamSynchronize(touch(ctx));
  resolve_timing(ctx);
  free_scene(ctx);
  for (auto& kv : ctx->textures) if (!kv.second.external && kv.second.own) (void)hipFree(kv.second.own);
  if (ctx->slab) (void)hipFree(ctx->slab);
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->ev_switch) (void)hipEventDestroy(ctx->ev_switch);
  for (int a = 0; a < 2; a++) for (int r = 0; r < 4; r++) if (ctx->q.s[a][r]) (void)hipFree(ctx->q.s[a][r]);
  if (ctx->q.counts) (void)hipFree(ctx->q.counts);
  if (ctx->zero_sky) (void)hipFree(ctx->zero_sky);
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  if (ctx->d_next) (void)hipFree(ctx->d_next);
  if (ctx->d_next2) (void)hipFree(ctx->d_next2);
  if (ctx->d_mail) (void)hipFree(ctx->d_mail);
  if (ctx->d_tables) (void)hipFree(ctx->d_tables);
  if (ctx->h_tables) (void)hipHostFree(ctx->h_tables);
  if (ctx->h_trip_flag) (void)hipHostFree(ctx->h_trip_flag);
  for (hipEvent_t e : ctx->table_ev) if (e) (void)hipEventDestroy(e);
  for (auto& r : ctx->rslot) {
    if (r.done) { (void)hipEventSynchronize(r.done); (void)hipEventDestroy(r.done); }
    if (r.snap) (void)hipEventDestroy(r.snap);
    if (r.dev) (void)hipFree(r.dev);
    if (r.host) (void)hipHostFree(r.host);
  }
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->srgb_first) (void)hipFree(ctx->srgb_first);
  for (int k = 0; k < 2; k++) {
    if (ctx->trace_q[k]) { (void)hipStreamSynchronize(ctx->trace_q[k]); (void)hipStreamDestroy(ctx->trace_q[k]); }
    if (ctx->trace_done[k]) (void)hipEventDestroy(ctx->trace_done[k]);
    if (ctx->pre_ev[k]) (void)hipEventDestroy(ctx->pre_ev[k]);
  }
  if (ctx->dep_ev) (void)hipEventDestroy(ctx->dep_ev);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return URT_OK;
}

const char* urt_last_error(urt_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int urt_context_set_stream(urt_context* ctx, void* hip_stream) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  { int rc = flush_pending(ctx); if (rc) return rc; }
  hipStream_t to = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  if (to == ctx->stream) return URT_OK;
  // no host synchronisation: everything issued so far on the old stream is ordered before whatever is issued on the new
  // one by an event (a caller that ping-pongs between a render and a communication stream must not stall on either)
  if (!ctx->ev_switch) URT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_switch, hipEventDisableTiming));
  URT_HIP(ctx, hipEventRecord(ctx->ev_switch, touch(ctx)));
  URT_HIP(ctx, hipStreamWaitEvent(to, ctx->ev_switch, 0));
  ctx->stream = to;
  return URT_OK;
}

int urt_flush(urt_context* ctx) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  URT_GUARD_BEGIN
  return flush_pending(ctx);
  URT_GUARD_END(ctx)
}

int urt_synchronize(urt_context* ctx) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  { int rc = flush_pending(ctx); if (rc) return rc; }
  URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));
  return check_watchdog(ctx);
}

/* ---- ComputeBuffer ---- */
int urt_buffer_create(urt_context* ctx, int count, int stride, urt_handle* out_buffer) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!out_buffer) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "out_buffer is NULL");
  // Unity: "ComputeBuffer count/stride must be greater than 0" and stride a multiple of 4
  if (count <= 0 || stride <= 0 || (stride & 3)) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "ComputeBuffer: count and stride must be > 0 and stride a multiple of 4");
  URT_GUARD_BEGIN
  Buffer b; b.count = count; b.stride = stride;
  b.host.assign((size_t)count * (size_t)stride, 0);
  urt_handle h = ctx->next_id++;
  ctx->buffers.emplace(h, std::move(b));
  *out_buffer = h;
  return URT_OK;
  URT_GUARD_END(ctx)
}

int urt_buffer_set_data(urt_context* ctx, urt_handle buffer, const void* data, int count) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  auto it = ctx->buffers.find(buffer);
  if (it == ctx->buffers.end()) return fail(ctx, URT_ERR_INVALID_HANDLE, "SetData: unknown buffer handle");
  Buffer& b = it->second;
  if (count < 0 || count > b.count) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "SetData: more elements than the buffer holds");
  if (count > 0 && !data) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "SetData: data is NULL");
  // The reference re-uploads EVERY list whenever anything changed (RM:738-745): data equal to what the buffer already holds changes
  // nothing and dirties nothing (a compare costs what the copy would)
  const size_t bytes = (size_t)count * (size_t)b.stride;
  if (b.has_data && (bytes == 0 || std::memcmp(b.host.data(), data, bytes) == 0)) return URT_OK;
  if (count > 0) std::memcpy(b.host.data(), data, bytes);
  b.has_data = true;
  for (int s = 0; s < B_COUNT; s++) if (ctx->bound[s] == buffer) { ctx->scene_dirty = true; ctx->dirty_slots |= 1u << s; }
  return URT_OK;
}

int urt_buffer_get_info(urt_context* ctx, urt_handle buffer, int* out_count, int* out_stride) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  auto it = ctx->buffers.find(buffer);
  if (it == ctx->buffers.end()) return fail(ctx, URT_ERR_INVALID_HANDLE, "unknown buffer handle");
  if (out_count) *out_count = it->second.count;
  if (out_stride) *out_stride = it->second.stride;
  return URT_OK;
}

int urt_buffer_release(urt_context* ctx, urt_handle buffer) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  auto it = ctx->buffers.find(buffer);
  if (it == ctx->buffers.end()) return fail(ctx, URT_ERR_INVALID_HANDLE, "Release: unknown buffer handle");
  for (int s = 0; s < B_COUNT; s++) if (ctx->bound[s] == buffer) { ctx->bound[s] = 0; ctx->scene_dirty = true; ctx->dirty_full = true; }
  ctx->buffers.erase(it);
  return URT_OK;
}

/* ---- textures ---- */
static int texture_create_impl(urt_context* ctx, int width, int height, void* ext, urt_handle* out_texture) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!out_texture) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "out_texture is NULL");
  if (width <= 0 || height <= 0) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "texture size must be positive");
  URT_GUARD_BEGIN
  URT_HIP(ctx, hipSetDevice(ctx->device));
  Texture t; t.w = width; t.h = height;
  size_t bytes = (size_t)width * (size_t)height * sizeof(float4);
  if (ext) { t.dev = (float4*)ext; t.external = true; t.other_writes = true; /* caller memory: contents unknown */ }
  else {
    URT_HIP(ctx, hipMalloc((void**)&t.dev, bytes));
    hipError_t e = hipMemsetAsync(t.dev, 0, bytes, touch(ctx));
    if (e != hipSuccess) { (void)hipFree(t.dev); return fail(ctx, URT_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e)); }
  }
  t.own = t.dev;
  urt_handle h = ctx->next_id++;
  ctx->textures.emplace(h, t);
  *out_texture = h;
  return URT_OK;
  URT_GUARD_END(ctx)
}

int urt_texture_create(urt_context* ctx, int width, int height, urt_handle* out_texture) {
  return texture_create_impl(ctx, width, height, nullptr, out_texture);
}

int urt_texture_create_external(urt_context* ctx, int width, int height, void* device_ptr, urt_handle* out_texture) {
  if (ctx && !device_ptr) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "device_ptr is NULL");
  return texture_create_impl(ctx, width, height, device_ptr, out_texture);
}

int urt_texture_set_pixels(urt_context* ctx, urt_handle texture, const float* rgba) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  Texture* t = find_texture(ctx, texture);
  if (!t) return fail(ctx, URT_ERR_INVALID_HANDLE, "unknown texture handle");
  if (!rgba) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "rgba is NULL");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  { int rc = flush_pending(ctx); if (rc) return rc; }
  t->other_writes = true;
  URT_HIP(ctx, hipMemcpyAsync(t->dev, rgba, (size_t)t->w * t->h * sizeof(float4), hipMemcpyHostToDevice, touch(ctx)));
  URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));
  return URT_OK;
}

int urt_texture_get_pixels(urt_context* ctx, urt_handle texture, float* rgba) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  Texture* t = find_texture(ctx, texture);
  if (!t) return fail(ctx, URT_ERR_INVALID_HANDLE, "unknown texture handle");
  if (!rgba) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "rgba is NULL");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  { int rc = flush_pending(ctx); if (rc) return rc; }
  URT_HIP(ctx, hipMemcpyAsync(rgba, t->dev, (size_t)t->w * t->h * sizeof(float4), hipMemcpyDeviceToHost, touch(ctx)));
  URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));
  return check_watchdog(ctx);                           // pixels of a launch that hit a cap are not handed out as good
}

// Pipelined readback.  begin: the image as it is at this point of the program order is snapshot on the render stream (a device-to-device
// copy: 33 MB at 1080p, ~20 us) and travels to a pinned host image on a stream of its own, so the frames dispatched AFTER the call render
// while it is on the PCIe bus; end: waits for that one copy and hands the pinned image out.
int urt_texture_read_begin(urt_context* ctx, urt_handle texture, uint64_t* out_ticket) { return urt_texture_read_begin_format(ctx, texture, URT_FORMAT_RGBA32F, out_ticket); }

// ... in the format of the host's `destination` (csrc/present.hip): the snapshot on the render stream IS the conversion kernel (16 B read,
// 4 or 8 B written per pixel), and only the converted image crosses the bus.
int urt_texture_read_begin_format(urt_context* ctx, urt_handle texture, int format, uint64_t* out_ticket) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!out_ticket) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "out_ticket is NULL");
  const size_t bpp = urtd::format_pixel_bytes(format);
  if (!bpp) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "urt_texture_read_begin_format: format must be URT_FORMAT_RGBA32F, _RGBA8_SRGB or _RGBA16F");
  Texture* t = find_texture(ctx, texture);
  if (!t) return fail(ctx, URT_ERR_INVALID_HANDLE, "unknown texture handle");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  { int rc = flush_pending(ctx); if (rc) return rc; }
  t = find_texture(ctx, texture);
  if (!ctx->copy_stream) URT_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  urt_context::ReadSlot& r = ctx->rslot[ctx->read_next % urt_context::kReadSlots];
  if (r.busy) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "urt_texture_read_begin: three readbacks are in flight — end the oldest first");
  const size_t px = (size_t)t->w * (size_t)t->h;
  if (r.pixels < px) {
    if (r.done) URT_HIP(ctx, hipEventSynchronize(r.done));
    if (r.dev) (void)hipFree(r.dev);
    if (r.host) (void)hipHostFree(r.host);
    r.dev = nullptr; r.host = nullptr; r.pixels = 0;
    URT_HIP(ctx, hipMalloc((void**)&r.dev, px * sizeof(float4)));
    URT_HIP(ctx, hipHostMalloc((void**)&r.host, px * sizeof(float4), hipHostMallocDefault));
    r.pixels = px;
  }
  if (!r.snap) { URT_HIP(ctx, hipEventCreateWithFlags(&r.snap, hipEventDisableTiming)); URT_HIP(ctx, hipEventCreateWithFlags(&r.done, hipEventDisableTiming)); }
  if (format == urtd::kFormatRGBA32F) {
    URT_HIP(ctx, hipMemcpyAsync(r.dev, t->dev, px * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    if (format == urtd::kFormatRGBA8sRGB && !ctx->srgb_first) {
      float first[urtd::kSrgbCodes];
      (void)urt_host_srgb8_first_floats(first);
      URT_HIP(ctx, hipMalloc((void**)&ctx->srgb_first, sizeof(first)));
      URT_HIP(ctx, hipMemcpy(ctx->srgb_first, first, sizeof(first), hipMemcpyHostToDevice));
    }
    URT_HIP(ctx, urtd::launch_encode(t->dev, r.dev, px, format, ctx->srgb_first, ctx->stream));
  }
  r.format = format;
  r.bytes = px * bpp;
  URT_HIP(ctx, hipEventRecord(r.snap, ctx->stream));
  URT_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, r.snap, 0));
  URT_HIP(ctx, hipMemcpyAsync(r.host, r.dev, r.bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
  URT_HIP(ctx, hipEventRecord(r.done, ctx->copy_stream));
  r.busy = true;
  r.ticket = ++ctx->read_next;                               // tickets start at 1; slot = (ticket - 1) % kReadSlots
  *out_ticket = r.ticket;
  return URT_OK;
}

int urt_texture_read_end(urt_context* ctx, uint64_t ticket, const float** out_rgba) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!out_rgba || ticket == 0) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "bad ticket / out_rgba is NULL");
  urt_context::ReadSlot& r = ctx->rslot[(ticket - 1) % urt_context::kReadSlots];
  if (r.busy && r.ticket == ticket && r.format != urtd::kFormatRGBA32F)
    return fail(ctx, URT_ERR_INVALID_ARGUMENT, "urt_texture_read_end: this ticket holds a converted image — end it with urt_texture_read_end_format");
  const void* p = nullptr;
  int rc = urt_texture_read_end_format(ctx, ticket, &p, nullptr);
  if (rc == URT_OK || rc == URT_ERR_WATCHDOG) *out_rgba = (const float*)p;
  return rc;
}

int urt_texture_read_end_format(urt_context* ctx, uint64_t ticket, const void** out_pixels, size_t* out_bytes) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!out_pixels || ticket == 0) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "bad ticket / out_pixels is NULL");
  urt_context::ReadSlot& r = ctx->rslot[(ticket - 1) % urt_context::kReadSlots];
  if (!r.busy || r.ticket != ticket) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "urt_texture_read_end: this ticket is not in flight");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  URT_HIP(ctx, hipEventSynchronize(r.done));
  r.busy = false;
  *out_pixels = r.host;                                      // valid until the third urt_texture_read_begin after this one
  if (out_bytes) *out_bytes = r.bytes;
  return check_watchdog(ctx);
}

int urt_texture_get_info(urt_context* ctx, urt_handle texture, int* out_width, int* out_height, void** out_device_ptr) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  Texture* t = find_texture(ctx, texture);
  if (!t) return fail(ctx, URT_ERR_INVALID_HANDLE, "unknown texture handle");
  if (out_width) *out_width = t->w;
  if (out_height) *out_height = t->h;
  if (out_device_ptr) {
    // the caller is going to touch the memory itself: submit what is deferred, give the image back its own (stable)
    // storage and never rename it again
    URT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = flush_pending(ctx); if (rc) return rc;
    rc = detach_from_slab(ctx, *t); if (rc) return rc;
    t->ptr_exposed = true;
    *out_device_ptr = t->dev;
  }
  return URT_OK;
}

int urt_texture_release(urt_context* ctx, urt_handle texture) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  auto it = ctx->textures.find(texture);
  if (it == ctx->textures.end()) return fail(ctx, URT_ERR_INVALID_HANDLE, "Release: unknown texture handle");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  { int rc = flush_pending(ctx); if (rc) return rc; }
  URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));
  if (!it->second.external && it->second.own) (void)hipFree(it->second.own);
  ctx->slab_oom_stride = 0;                                // device memory came back: the next batch may try the Result slots again
  if (ctx->slab_tex == texture) ctx->slab_tex = 0;
  if (ctx->t_sky == texture) ctx->t_sky = 0;
  if (ctx->t_result == texture) ctx->t_result = 0;
  ctx->textures.erase(it);
  return URT_OK;
}

/* ---- shader uniforms and bindings ---- */
int urt_shader_set_buffer(urt_context* ctx, int kernel, const char* name, urt_handle buffer) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (kernel != 0 || !name) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "SetBuffer: kernel must be 0 and name non-NULL");
  for (int s = 0; s < B_COUNT; s++) {
    if (std::strcmp(name, kBindNames[s]) != 0) continue;
    if (buffer) {
      auto it = ctx->buffers.find(buffer);
      if (it == ctx->buffers.end()) return fail(ctx, URT_ERR_INVALID_HANDLE, "SetBuffer: unknown buffer handle");
      if (it->second.stride != kBindStride[s])
        return fail(ctx, URT_ERR_LAYOUT, std::string("SetBuffer(") + name + "): stride " + std::to_string(it->second.stride) +
                                             " != " + std::to_string(kBindStride[s]) + " (RM:738-745)");
    }
    if (ctx->bound[s] != buffer) { ctx->bound[s] = buffer; ctx->scene_dirty = true; ctx->dirty_full = true; }   // re-binding the same buffer every frame (RM:787-794) is free
    return URT_OK;
  }
  return fail(ctx, URT_ERR_INVALID_ARGUMENT, std::string("SetBuffer: kernel CSMain has no buffer named ") + name);
}

int urt_shader_set_texture(urt_context* ctx, int kernel, const char* name, urt_handle texture) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (kernel != 0 || !name) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "SetTexture: kernel must be 0 and name non-NULL");
  if (texture && !find_texture(ctx, texture)) return fail(ctx, URT_ERR_INVALID_HANDLE, "SetTexture: unknown texture handle");
  if (std::strcmp(name, "_SkyboxTexture") == 0) { ctx->t_sky = texture; return URT_OK; }
  if (std::strcmp(name, "Result") == 0) { ctx->t_result = texture; return URT_OK; }
  return fail(ctx, URT_ERR_INVALID_ARGUMENT, std::string("SetTexture: kernel CSMain has no texture named ") + name);
}

int urt_shader_set_matrix(urt_context* ctx, const char* name, const float* m16) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!name || !m16) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "SetMatrix: NULL argument");
  if (std::strcmp(name, "_CameraToWorld") == 0) std::memcpy(ctx->c2w, m16, sizeof ctx->c2w);
  else if (std::strcmp(name, "_CameraInverseProjection") == 0) std::memcpy(ctx->invp, m16, sizeof ctx->invp);
  return URT_OK;   // undeclared names are ignored, as Unity does
}

int urt_shader_set_vector(urt_context* ctx, const char* name, const float* v4) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!name || !v4) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "SetVector: NULL argument");
  if (std::strcmp(name, "_PixelOffset") == 0) { ctx->pixel_off[0] = v4[0]; ctx->pixel_off[1] = v4[1]; }
  return URT_OK;
}

int urt_shader_set_float(urt_context* ctx, const char* name, float value) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!name) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "SetFloat: NULL name");
  if (std::strcmp(name, "_Seed") == 0) ctx->seed = value;
  return URT_OK;
}

int urt_shader_set_int(urt_context* ctx, const char* name, int value) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!name) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "SetInt: NULL name");
  if (std::strcmp(name, "_numBounces") == 0) ctx->num_bounces = value;
  else if (std::strcmp(name, "_numRays") == 0) ctx->num_rays = value;
  // "_MeshBVH_len" / "_SphereBVH_len": static const in the shader (RS:73-74) — accepted, no effect
  return URT_OK;
}

int urt_shader_dispatch(urt_context* ctx, int kernel, int groups_x, int groups_y, int groups_z) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  URT_GUARD_BEGIN
  return do_dispatch(ctx, kernel, groups_x, groups_y, groups_z, 0, 1);
  URT_GUARD_END(ctx)
}

int urt_shader_dispatch_rows(urt_context* ctx, int kernel, int groups_x, int groups_y, int groups_z, int first_group_row,
                             int row_stride) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  URT_GUARD_BEGIN
  return do_dispatch(ctx, kernel, groups_x, groups_y, groups_z, first_group_row, row_stride);
  URT_GUARD_END(ctx)
}

/* ---- blits ---- */
int urt_blit_add(urt_context* ctx, urt_handle src, urt_handle dst, float sample) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  Texture* s = find_texture(ctx, src);
  Texture* d = find_texture(ctx, dst);
  if (!s || !d) return fail(ctx, URT_ERR_INVALID_HANDLE, "Blit: unknown texture handle");
  if (s->w != d->w || s->h != d->h) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "Blit: source and destination sizes differ");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  URT_GUARD_BEGIN
  d->other_writes = true;
  urt_context::Pending& B = ctx->pend;
  if (B.n > 0 && src == B.tex && dst != src && (const float4*)d->dev != B.S.sky) {
    // the source is a frame that has not been traced yet: the blend is deferred with it (flush_pending runs it in order)
    // (a full batch is submitted by the next dispatch or observer — the present of this frame, RM:819, may still follow)
    B.ops.push_back(urt_context::PostOp{0, B.n - 1, src, dst, sample, 0, 1, nullptr});
    return URT_OK;
  }
  { int rc = flush_pending(ctx); if (rc) return rc; }
  URT_HIP(ctx, launch_blit_add(s->dev, d->dev, (size_t)s->w * s->h, sample, touch(ctx)));
  return URT_OK;
  URT_GUARD_END(ctx)
}

int urt_blit(urt_context* ctx, urt_handle src, urt_handle dst) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  Texture* s = find_texture(ctx, src);
  Texture* d = find_texture(ctx, dst);
  if (!s || !d) return fail(ctx, URT_ERR_INVALID_HANDLE, "Blit: unknown texture handle");
  if (s->w != d->w || s->h != d->h) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "Blit: source and destination sizes differ");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  URT_GUARD_BEGIN
  urt_context::Pending& B = ctx->pend;
  if (B.n > 0 && dst != B.tex && dst != src && (const float4*)d->dev != B.S.sky) {
    // frames are deferred: the copy is queued behind them, in program order (the present of RM:819 — its source is the image
    // the deferred blends accumulate into).  flush_pending fuses it into the blend pass.
    d->other_writes = true;
    B.ops.push_back(urt_context::PostOp{2, B.n - 1, src, dst, 0.0f, 0, 1, nullptr});
    if (B.n >= B.limit) return flush_pending(ctx);        // the batch is full and its last frame is presented: go
    return URT_OK;
  }
  { int rc = flush_pending(ctx); if (rc) return rc; }
  d->other_writes = true;
  if (dst != src)
    URT_HIP(ctx, hipMemcpyAsync(d->dev, s->dev, (size_t)s->w * s->h * sizeof(float4), hipMemcpyDeviceToDevice, touch(ctx)));
  return URT_OK;
  URT_GUARD_END(ctx)
}

static int pack_impl(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride, void* dense, bool to_dense,
                     uint64_t* out_bytes, bool rgb = false, float alpha = 0.0f) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  Texture* t = find_texture(ctx, texture);
  if (!t) return fail(ctx, URT_ERR_INVALID_HANDLE, "unknown texture handle");
  if (first_group_row < 0 || row_stride < 1) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "bad strip arguments");
  int group_rows = (t->h + 7) / 8;
  int n_strips = first_group_row < group_rows ? (group_rows - first_group_row + row_stride - 1) / row_stride : 0;
  if (out_bytes) *out_bytes = (uint64_t)n_strips * 8u * (uint64_t)t->w * (rgb ? 3 * sizeof(float) : sizeof(float4));
  if (!dense) return URT_OK;   // size query
  URT_HIP(ctx, hipSetDevice(ctx->device));
  URT_GUARD_BEGIN
  urt_context::Pending& B = ctx->pend;
  if (to_dense && B.n > 0) {     // reads an image that deferred work is still going to write: deferred with it, in order
    B.ops.push_back(urt_context::PostOp{1, B.n - 1, texture, 0, rgb ? 1.0f : 0.0f, first_group_row, row_stride, dense});   // (sample != 0: the RGB form)
    return URT_OK;
  }
  { int rc = flush_pending(ctx); if (rc) return rc; }
  if (!to_dense) t->other_writes = true;
  URT_GUARD_END(ctx)
  if (rgb) URT_HIP(ctx, launch_pack_rows_rgb(t->dev, (float*)dense, t->w, t->h, first_group_row, row_stride, n_strips, to_dense, alpha, touch(ctx)));
  else URT_HIP(ctx, launch_pack_rows(t->dev, (float4*)dense, t->w, t->h, first_group_row, row_stride, n_strips, to_dense, touch(ctx)));
  return URT_OK;
}

int urt_texture_pack_rows(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride, void* device_dst,
                          uint64_t* out_bytes) {
  return pack_impl(ctx, texture, first_group_row, row_stride, device_dst, true, out_bytes);
}

static int unpack_on_impl(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride, const void* device_src,
                          void* hip_stream, bool rgb, float alpha) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!device_src || !hip_stream) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "device_src / hip_stream is NULL");
  Texture* t = find_texture(ctx, texture);
  if (!t) return fail(ctx, URT_ERR_INVALID_HANDLE, "unknown texture handle");
  if (first_group_row < 0 || row_stride < 1) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "bad strip arguments");
  if (ctx->pend.n > 0) {                                  // never the case for a dedicated gather target
    bool touched = ctx->pend.tex == texture;
    for (const urt_context::PostOp& q : ctx->pend.ops) touched = touched || q.tex == texture || q.dst == texture;
    if (touched) { int rc = flush_pending(ctx); if (rc) return rc; }
  }
  if (in_slab(ctx, *t)) { int rc = detach_from_slab(ctx, *t); if (rc) return rc; }
  int group_rows = (t->h + 7) / 8;
  int n_strips = first_group_row < group_rows ? (group_rows - first_group_row + row_stride - 1) / row_stride : 0;
  t->other_writes = true;
  URT_HIP(ctx, hipSetDevice(ctx->device));
  if (rgb) URT_HIP(ctx, launch_pack_rows_rgb(t->dev, (float*)const_cast<void*>(device_src), t->w, t->h, first_group_row, row_stride, n_strips, false,
                                            alpha, (hipStream_t)hip_stream));
  else URT_HIP(ctx, launch_pack_rows(t->dev, (float4*)const_cast<void*>(device_src), t->w, t->h, first_group_row, row_stride, n_strips, false,
                                     (hipStream_t)hip_stream));
  return URT_OK;
}

int urt_texture_unpack_rows_on(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride, const void* device_src,
                               void* hip_stream) {
  return unpack_on_impl(ctx, texture, first_group_row, row_stride, device_src, hip_stream, false, 0.0f);
}

int urt_texture_pack_rows_rgb(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride, void* device_dst,
                              uint64_t* out_bytes) {
  return pack_impl(ctx, texture, first_group_row, row_stride, device_dst, true, out_bytes, true);
}

int urt_texture_unpack_rows_rgb(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride, const void* device_src,
                                float alpha, void* hip_stream) {
  if (ctx && !device_src) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "device_src is NULL");
  if (hip_stream) return unpack_on_impl(ctx, texture, first_group_row, row_stride, device_src, hip_stream, true, alpha);
  return pack_impl(ctx, texture, first_group_row, row_stride, const_cast<void*>(device_src), false, nullptr, true, alpha);
}

int urt_texture_unpack_rows(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride, const void* device_src) {
  if (ctx && !device_src) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "device_src is NULL");
  return pack_impl(ctx, texture, first_group_row, row_stride, const_cast<void*>(device_src), false, nullptr);
}

/* ---- measurement ---- */
int urt_set_option(urt_context* ctx, const char* name, int value) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!name) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "option name is NULL");
  { (void)hipSetDevice(ctx->device); int rc = flush_pending(ctx); if (rc) return rc; }   // deferred frames run with the options they were dispatched under
  if (std::strcmp(name, "blas_builder") == 0) {
    if (value < -1 || value > 3) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "blas_builder must be -1 (auto), 0 (host SAH), 1 (GPU LBVH), 2 (GPU LBVH built top-down within a depth budget) or 3 (binned SAH on the GPU)");
    if (value != ctx->opt_blas_builder) { ctx->scene_dirty = true; ctx->dirty_full = true; }
    ctx->opt_blas_builder = value;
  } else if (std::strcmp(name, "frames_per_launch") == 0) {
    if (value < 0 || value > kMaxFramesPerLaunch) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "frames_per_launch must be in [0, 64] (0 = auto)");
    ctx->opt_frames_per_launch = value;
  } else if (std::strcmp(name, "count_stats") == 0) ctx->opt_count_stats = value ? 1 : 0;
  else if (std::strcmp(name, "time_dispatch") == 0) ctx->opt_time_dispatch = value ? 1 : 0;
  else if (std::strcmp(name, "kernel_mode") == 0) {
    if (value < 0 || value > 5) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "kernel_mode must be 0..5");
    ctx->opt_kernel_mode = value;
  } else if (std::strcmp(name, "block_threads") == 0) {
    if (value != 64 && value != 128 && value != 256) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "block_threads must be 64, 128 or 256");
    ctx->opt_block_threads = value;
  } else if (std::strcmp(name, "blas_leaf_max") == 0) {
    if (value < 1 || value > 8) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "blas_leaf_max must be in [1, 8]");
    set_blas_leaf_max(value);
    ctx->scene_dirty = true; ctx->dirty_full = true;
  } else if (std::strcmp(name, "blas_min") == 0) {
    if (value < 0 || value > 256) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "blas_min must be in [0, 256] (0 = auto; kernel_mode 5 counts the waiting rays of a workgroup)");
    ctx->opt_blas_min = value;
  } else if (std::strcmp(name, "blas_exit") == 0) {
    if (value < 0 || value > 64) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "blas_exit must be in [0, 64] (0 = auto)");
    ctx->opt_blas_exit = value;
  } else if (std::strcmp(name, "refill_min") == 0) {
    if (value < 1 || value > 64) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "refill_min must be in [1, 64]");
    ctx->opt_refill_min = value;
  } else if (std::strcmp(name, "waves_per_cu") == 0) {
    if (value < 0 || value > 32) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "waves_per_cu must be in [0, 32] (0 = auto)");
    ctx->opt_waves_per_cu = value;
  } else if (std::strcmp(name, "sched_block") == 0) {
    if (value != 0 && value != 64 && value != 256) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "sched_block must be 0 (auto), 64 or 256");
    ctx->opt_sched_block = value;
  } else if (std::strcmp(name, "stack_pad") == 0) {
    if (value < 0 || value > 96) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "stack_pad must be in [0, 96]");
    ctx->opt_stack_pad = value;
  } else if (std::strcmp(name, "shade_min") == 0) {
    if (value < 1 || value > 64) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "shade_min must be in [1, 64]");
    ctx->opt_shade_min = value;
  } else if (std::strcmp(name, "front_list") == 0) {
    if (value < -1 || value > 2) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "front_list must be -1 (auto), 0, 1 or 2");
    ctx->opt_front_list = value;
  } else if (std::strcmp(name, "shade_split") == 0) {
    if (value < -1 || value > 1) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "shade_split must be -1 (auto), 0 or 1");
    ctx->opt_shade_split = value;
  } else if (std::strcmp(name, "serve_refill") == 0) {
    if (value < 1 || value > 64) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "serve_refill must be in [1, 64]");
    ctx->opt_serve_refill = value;
  } else if (std::strcmp(name, "sky_min") == 0) {
    if (value < 1 || value > 64) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "sky_min must be in [1, 64]");
    ctx->opt_sky_min = value;
  } else if (std::strcmp(name, "tile_order") == 0) {
    if (value < -1 || value > 1) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "tile_order must be -1 (auto), 0 or 1");
    ctx->opt_tile_order = value;
  } else if (std::strcmp(name, "lds_tlas") == 0) {
    ctx->opt_lds_tlas = value ? 1 : 0;
  } else if (std::strcmp(name, "top_front") == 0) {
    if (value < -1 || value > 1) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "top_front must be -1 (auto), 0 or 1");
    ctx->opt_top_front = value;
  } else if (std::strcmp(name, "top_nodes") == 0) {
    if (value < -1 || value > kTopOrderNodes) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "top_nodes must be in [0, 256], or -1 (auto)");
    ctx->opt_top_nodes = value;
  } else if (std::strcmp(name, "pool_k") == 0) {
    if (value < 1 || value > 4) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "pool_k must be in [1, 4]");
    ctx->opt_pool_k = value;
  } else if (std::strcmp(name, "pool_refill") == 0) {
    if (value < 1 || value > 256) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "pool_refill must be in [1, 256]");
    ctx->opt_pool_refill = value;
  } else if (std::strcmp(name, "pool_blas_min") == 0) {
    if (value < 1 || value > 256) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "pool_blas_min must be in [1, 256]");
    ctx->opt_pool_blas_min = value;
  } else if (std::strcmp(name, "pool_blas_exit") == 0) {
    if (value < 1 || value > 64) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "pool_blas_exit must be in [1, 64]");
    ctx->opt_pool_blas_exit = value;
  } else if (std::strcmp(name, "pool_other_min") == 0) {
    if (value < 1 || value > 64) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "pool_other_min must be in [1, 64]");
    ctx->opt_pool_other_min = value;
  } else if (std::strcmp(name, "pool_inloop") == 0) {
    if (value < 1 || value > 64) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "pool_inloop must be in [1, 64]");
    ctx->opt_pool_inloop = value;
  } else if (std::strcmp(name, "frame_group") == 0) {
    if (value < 1 || value > kMaxFramesPerLaunch) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "frame_group must be in [1, 64]");
    ctx->opt_frame_group = value;
  } else if (std::strcmp(name, "work_shards") == 0) {
    if (value < 1 || value > (int)kWorkShards || (value & (value - 1))) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "work_shards must be a power of two in [1, 64]");
    ctx->opt_work_shards = value;
  } else if (std::strcmp(name, "qnodes") == 0) {
    if (value < -1 || value > 1) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "qnodes must be -1 (auto), 0 or 1");
    ctx->opt_qnodes = value;
    ctx->scene_dirty = true; ctx->dirty_full = true;
  } else if (std::strcmp(name, "lbvh_slack") == 0) {
    if (value < 0 || value > 16) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "lbvh_slack must be 0..16");
    if (value != ctx->opt_lbvh_slack) { ctx->scene_dirty = true; ctx->dirty_full = true; }
    ctx->opt_lbvh_slack = value;
  } else if (std::strcmp(name, "overlap_launches") == 0) {
    if (value < 0 || value > 2) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "overlap_launches must be 0 (off), 1 (auto) or 2 (always)");
    { int rc = flush_pending(ctx); if (rc) return rc; }
    ctx->opt_overlap = value;
  } else if (std::strcmp(name, "front_cull") == 0) {
    if (value < 0 || value > 1) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "front_cull must be 0 or 1");
    if (ctx->opt_front_cull != value) { int rc = flush_pending(ctx); if (rc) return rc; ctx->opt_front_cull = value; ctx->scene_dirty = true; ctx->dirty_full = true; }
  } else if (std::strcmp(name, "refit") == 0) {
    ctx->opt_refit = value ? 1 : 0;
    ctx->scene_dirty = true; ctx->dirty_full = true;
  } else if (std::strcmp(name, "watchdog_cap") == 0) {
    if (value < 0) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "watchdog_cap must be >= 0 (0 = auto)");
    ctx->opt_watchdog_cap = value;
  } else if (std::strcmp(name, "xcd_run") == 0) {
    if (value < 0 || value > 4096) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "xcd_run must be in [0, 4096] (0 = auto)");
    ctx->opt_xcd_run = value;
  } else return fail(ctx, URT_ERR_INVALID_ARGUMENT, std::string("unknown option ") + name);
  return URT_OK;
}

int urt_get_counters(urt_context* ctx, urt_counters* out) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!out) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "out is NULL");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  { int rc = flush_pending(ctx); if (rc) return rc; }
  URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));
  resolve_timing(ctx);
  std::vector<DevCounters> shards(kCounterShards);
  URT_HIP(ctx, hipMemcpy(shards.data(), ctx->d_counters, sizeof(DevCounters) * kCounterShards, hipMemcpyDeviceToHost));
  std::memset(out, 0, sizeof *out);
  for (const DevCounters& dc : shards) {
    out->rays += dc.rays; out->tlas_nodes += dc.tlas_nodes; out->blas_nodes += dc.blas_nodes; out->tri_tests += dc.tri_tests;
    out->sphere_tests += dc.sphere_tests; out->hit_tri += dc.hit_tri; out->hit_sphere += dc.hit_sphere;
    out->hit_ground += dc.hit_ground; out->hit_sky += dc.hit_sky;
    out->watchdog_trips += (uint32_t)dc.watchdog;
  }
  out->pixels = ctx->pixels_dispatched;
  out->dispatches = ctx->dispatches;
  out->launches = ctx->launches;
  out->trace_ms = ctx->trace_ms;
  return URT_OK;
}

int urt_reset_counters(urt_context* ctx) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  { int rc = flush_pending(ctx); if (rc) return rc; }
  URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));
  resolve_timing(ctx);
  URT_HIP(ctx, hipMemset(ctx->d_counters, 0, sizeof(DevCounters) * kCounterShards));
  if (ctx->h_trip_flag) __atomic_store_n(ctx->h_trip_flag, 0u, __ATOMIC_RELEASE);
  ctx->dispatches = 0;
  ctx->launches = 0;
  ctx->pixels_dispatched = 0;
  ctx->trace_ms = 0;
  return URT_OK;
}

#ifdef URT_STAMPS
/* diagnostic builds only: per-wave (start, pool-exhausted, end, iters<<32|fetches) of the last persistent launch */
__attribute__((visibility("default"))) int urt_debug_read_stamps(urt_context* ctx, unsigned long long* out, int n_waves) {
  (void)hipStreamSynchronize(touch(ctx));
  hipError_t e = hipMemcpy(out, (char*)ctx->d_next + kWorkShards * 128, (size_t)n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);   // n_waves = number of u64 words
  (void)hipMemset((char*)ctx->d_next + kWorkShards * 128, 0, 65536 * 16 * sizeof(unsigned long long));
  return (int)e;
}
#endif

/* kernel_mode 5 with count_stats: visits of the traversal service, its trips, active lanes summed over the trips, claim rounds,
   rays claimed, rays suspended — summed since the last urt_reset_counters */
int urt_debug_serve_stats(urt_context* ctx, unsigned long long* out6) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!out6) return fail(ctx, URT_ERR_INVALID_ARGUMENT, "out6 is NULL");
  URT_HIP(ctx, hipSetDevice(ctx->device));
  { int rc = flush_pending(ctx); if (rc) return rc; }
  URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));
  std::vector<DevCounters> shards(kCounterShards);
  URT_HIP(ctx, hipMemcpy(shards.data(), ctx->d_counters, sizeof(DevCounters) * kCounterShards, hipMemcpyDeviceToHost));
  for (int q = 0; q < 6; q++) out6[q] = 0;
  for (const DevCounters& dc : shards) for (int q = 0; q < 6; q++) out6[q] += dc.serve[q];
  return URT_OK;
}

/* ---- introspection ---- */
int urt_debug_build_blas(const void* mesh_objects, int n_meshes, const float* vertices, int n_vertices, const int32_t* indices,
                         int n_indices, int* out_n_nodes, int* out_n_tris, int* out_max_depth) {
  URT_GUARD_BEGIN
  if (n_meshes < 0 || (n_meshes > 0 && !mesh_objects)) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "mesh_objects is NULL");
  std::string err;
  if (!build_blas((const uint8_t*)mesh_objects, n_meshes, vertices, n_vertices, indices, n_indices, nullptr, 0, g_debug_blas, err))
    return fail(nullptr, URT_ERR_SCENE, err);
  if (out_n_nodes) *out_n_nodes = (int)(g_debug_blas.nodes.size() / kBlasNodeFloats);
  if (out_n_tris) *out_n_tris = (int)g_debug_blas.tri_slot.size();
  if (out_max_depth) *out_max_depth = g_debug_blas.max_depth;
  return URT_OK;
  URT_GUARD_END(nullptr)
}

int urt_debug_blas_cache_stats(urt_context* ctx, uint64_t* out_reused, uint64_t* out_built) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  std::lock_guard<std::mutex> g(ctx->blas_cache.lock);
  if (out_reused) *out_reused = ctx->blas_cache.hits;
  if (out_built) *out_built = ctx->blas_cache.builds;
  return URT_OK;
}

/* The masked-walk table the default kernel derives from a mesh heap of <= 31 nodes (build_walk_table above), for host-side tests:
   out = (20 + 2 * n_eval) * 4 words; returns the number of words written through out_words, 0 when the heap does not qualify. */
int urt_debug_build_walk_table(const urt_BVHNode* heap, int n_nodes, int n_meshes, const int32_t* mesh_root, const int32_t* small_first,
                               float* out, int capacity_words, int* out_words) {
  if (out_words) *out_words = 0;
  if (n_nodes < 0 || (n_nodes > 0 && !heap) || n_meshes < 0 || (n_meshes > 0 && !mesh_root)) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "urt_debug_build_walk_table: bad arguments");
  URT_GUARD_BEGIN
  Buffer b; b.count = n_nodes; b.stride = URT_STRIDE_BVHNODE; b.has_data = true;
  b.host.resize((size_t)n_nodes * URT_STRIDE_BVHNODE);
  if (n_nodes > 0) std::memcpy(b.host.data(), heap, b.host.size());
  std::vector<int32_t> roots(mesh_root, mesh_root + n_meshes), sf;
  if (small_first) sf.assign(small_first, small_first + n_meshes);
  std::vector<float> t;
  if (!build_walk_table(n_nodes > 0 ? &b : nullptr, n_meshes, roots, sf, t)) return URT_OK;
  if ((int)t.size() > capacity_words || !out) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "urt_debug_build_walk_table: output buffer too small");
  std::memcpy(out, t.data(), t.size() * sizeof(float));
  if (out_words) *out_words = (int)t.size();
  return URT_OK;
  URT_GUARD_END(nullptr)
}

int urt_debug_refit_stats(urt_context* ctx, uint64_t* out_refitted_meshes, uint64_t* out_incremental_preparations) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (out_refitted_meshes) *out_refitted_meshes = ctx->refitted_meshes;
  if (out_incremental_preparations) *out_incremental_preparations = ctx->incremental_preps;
  return URT_OK;
}

int urt_debug_launch_info(urt_context* ctx, urt_launch_info* out) {
  if (!ctx || !out) return URT_ERR_INVALID_ARGUMENT;
  int rc = flush_pending(ctx); if (rc) return rc;          // "the last launch" includes the frames still deferred
  *out = ctx->last_launch;
  out->slab_frames = ctx->slab_frames; out->slab_frames_max = ctx->slab_frames_max; out->slab_out_of_memory = ctx->slab_oom_stride != 0 ? 1 : 0;
  out->blas_builder = ctx->last_builder;
  out->overlapped_launches = (int)std::min<uint64_t>(ctx->overlapped_launches, 0x7fffffff);
  return URT_OK;
}

int urt_debug_scene_info(urt_context* ctx, int* out_n_nodes, int* out_n_tris, int* out_max_depth, float* out_prepare_ms) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  URT_GUARD_BEGIN
  URT_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->scene_dirty) { int rc = flush_pending(ctx); if (rc) return rc; rc = prepare_scene(ctx); if (rc) return rc; }
  if (out_n_nodes) *out_n_nodes = ctx->n_blas_nodes;
  if (out_n_tris) *out_n_tris = ctx->n_scene_tris;
  if (out_max_depth) *out_max_depth = ctx->scene_max_depth;
  if (out_prepare_ms) *out_prepare_ms = ctx->last_prepare_ms;
  return URT_OK;
  URT_GUARD_END(ctx)
}

int urt_debug_read_scene_blas(urt_context* ctx, float* nodes, int32_t* tri_index, int32_t* mesh_root) {
  if (!ctx) return fail(nullptr, URT_ERR_INVALID_ARGUMENT, "ctx is NULL");
  URT_GUARD_BEGIN
  URT_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->scene_dirty) { int rc = flush_pending(ctx); if (rc) return rc; rc = prepare_scene(ctx); if (rc) return rc; }
  URT_HIP(ctx, hipStreamSynchronize(touch(ctx)));
  const DevScene& S = ctx->ds;
  if (nodes && ctx->n_blas_nodes > 0) URT_HIP(ctx, hipMemcpy(nodes, S.blas_nodes, (size_t)ctx->n_blas_nodes * kBlasNodeFloats * sizeof(float), hipMemcpyDeviceToHost));
  if (mesh_root && S.n_meshes > 0) URT_HIP(ctx, hipMemcpy(mesh_root, S.mesh_root, (size_t)S.n_meshes * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (tri_index && ctx->n_scene_tris > 0) {
    std::vector<float> tv((size_t)ctx->n_scene_tris * 12);
    URT_HIP(ctx, hipMemcpy(tv.data(), S.tri_verts, tv.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int k = 0; k < ctx->n_scene_tris; k++) std::memcpy(&tri_index[k], &tv[(size_t)k * 12 + 3], 4);     // index slot kept in v0.w
  }
  return URT_OK;
  URT_GUARD_END(ctx)
}

int urt_debug_get_blas(float* nodes, int32_t* tri_index, int32_t* mesh_root, int32_t* mesh_first_tri) {
  const BlasResult& b = g_debug_blas;
  if (nodes && !b.nodes.empty()) std::memcpy(nodes, b.nodes.data(), b.nodes.size() * sizeof(float));
  if (tri_index && !b.tri_slot.empty()) std::memcpy(tri_index, b.tri_slot.data(), b.tri_slot.size() * sizeof(int32_t));
  if (mesh_root && !b.mesh_root.empty()) std::memcpy(mesh_root, b.mesh_root.data(), b.mesh_root.size() * sizeof(int32_t));
  if (mesh_first_tri && !b.mesh_first_tri.empty()) std::memcpy(mesh_first_tri, b.mesh_first_tri.data(), b.mesh_first_tri.size() * sizeof(int32_t));
  return URT_OK;
}

}  // extern "C"

// ---- internal accessors for group.cpp (not part of the C ABI) ----------------------------------------------------------
namespace urtd {
hipStream_t context_stream(urt_context* ctx) { return touch(ctx); }   // (csrc/group.cpp enqueues gathers and blits on it)
int context_device(urt_context* ctx) { return ctx->device; }
int context_pending_frames(urt_context* ctx) { return ctx->pend.n; }
}  // namespace urtd
