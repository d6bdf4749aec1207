/* urt.h — C ABI of libunityraytracer_amd.so: the MI355X (gfx950) replacement for the GPU calls that
 * RemyMuj/UnityRayTracer's Assets/Scripts/RayTraceMaster.cs ("RM") makes through UnityEngine:
 * ComputeBuffer / ComputeShader.Set* / ComputeShader.Dispatch / RenderTexture / Graphics.Blit.
 * Each entry point cites the reference call site it stands in for; the C# P/Invoke shim that binds
 * them is shown in INTEGRATION.md.  The kernels behind urt_shader_dispatch()/urt_blit_add() are
 * hand-written HIP restatements of Assets/Shaders/RayTraceShader.compute ("RS", kernel CSMain) and
 * Assets/Shaders/AdditionShader.shader ("AS").
 *
 * Conventions:
 *  - plain C types only; every function returns an int status (URT_OK == 0) and never unwinds;
 *    urt_last_error() returns the message for the last non-zero status (Unity's API returns void
 *    and logs — a caller that wants that behaviour logs the string and carries on);
 *  - host memory passed in is copied before the call returns (ComputeBuffer.SetData semantics);
 *  - handles are opaque 64-bit ids owned by the library until *_release();
 *  - one caller thread per context (Unity main thread); work is issued in order on one HIP stream,
 *    so no synchronisation is needed between dispatch and blit; urt_synchronize() or a readback
 *    waits for completion;
 *  - buffer layouts are the byte layouts of urt_types.h (RM:42-45), matrices are 16 floats in
 *    Unity Matrix4x4 memory order (column-major), images are RGBA32F with row 0 = bottom row
 *    (RS:434,468: id.y = 0 is uv.y = -1);
 *  - there is no CPU fallback: every entry point that needs the GPU fails with
 *    URT_ERR_NO_DEVICE if no HIP device is usable.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "urt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define URT_API __attribute__((visibility("default")))

typedef struct urt_context urt_context;
typedef uint64_t urt_handle;

enum {
  URT_OK = 0,
  URT_ERR_INVALID_ARGUMENT = 1,
  URT_ERR_INVALID_HANDLE = 2,
  URT_ERR_NO_DEVICE = 3,
  URT_ERR_HIP = 4,
  URT_ERR_UNBOUND = 5,        /* dispatch without a Result texture / camera matrices */
  URT_ERR_LAYOUT = 6,         /* stride/count does not match the layout the name requires */
  URT_ERR_OUT_OF_MEMORY = 7,
  URT_ERR_SCENE = 8,          /* scene data fails validation (index out of range, stack too deep) */
  URT_ERR_WATCHDOG = 9        /* waves of a trace launch left through the kernel's iteration cap: pixels are missing.  Reported
                                 once, by the first urt_synchronize / urt_texture_get_pixels after the launch completed */
};

/* ---- library / context ------------------------------------------------------------------- */
URT_API int urt_abi_version(void);                       /* bumps when this header changes */
URT_API int urt_device_count(int* out_count);
/* One context per process and GPU (the one-process-per-GPU model).  device = HIP ordinal. */
URT_API int urt_context_create(int device, urt_context** out_ctx);
URT_API int urt_context_destroy(urt_context* ctx);
/* Message for the last failing call on ctx (ctx may be NULL for create failures). Never NULL. */
URT_API const char* urt_last_error(urt_context* ctx);
/* Issue all work on a caller-owned hipStream_t (e.g. torch's current stream); NULL = own stream. */
URT_API int urt_context_set_stream(urt_context* ctx, void* hip_stream);
/* Block until everything issued so far is complete. */
URT_API int urt_synchronize(urt_context* ctx);
/* Frame batching.  On its own stream the library DEFERS urt_shader_dispatch (default kernel) and the urt_blit_add /
 * urt_blit / urt_texture_pack_rows calls that follow it, and traces several consecutive frames with one persistent launch
 * (a 1080p frame is too small to fill 256 CUs through its 8-bounce tail; "frames_per_launch" below).  The whole frame of
 * RM:806-820 — Dispatch, Blit(_target, _converged, mat), Blit(_converged, destination) — is deferred; the blends of a batch
 * and the present run as ONE pass after the launch.  Every call that could observe an image (readback, synchronize,
 * counters, SetPixels, release, a device pointer handed out, option changes, ...) submits the deferred work first, so
 * results and ordering are exactly those of immediate execution (RM:806-820 is in-order).
 * As-if rule for the present: a deferred urt_blit(src, dst) that is followed by another deferred urt_blit to the same dst,
 * with nothing that could observe dst in between, is not materialised — dst receives the later image only.  No observer
 * can tell (observers submit first); a caller that looks at dst's memory by its own means must call urt_flush /
 * urt_synchronize first, as for any deferred work.
 * An error of deferred work (a failing launch) is reported by the call that submits it.
 * urt_flush submits the deferred work to the stream without waiting — for callers that synchronise by their own means
 * (their own stream + events); with a caller-owned stream deferral is off unless "frames_per_launch" is set explicitly. */
URT_API int urt_flush(urt_context* ctx);

/* ---- ComputeBuffer ------------------------------------------------------------------------ */
/* new ComputeBuffer(count, stride)                                   RM:247 */
URT_API int urt_buffer_create(urt_context* ctx, int count, int stride, urt_handle* out_buffer);
/* buffer.SetData(List<T>) — synchronous copy of count*stride bytes   RM:250 */
URT_API int urt_buffer_set_data(urt_context* ctx, urt_handle buffer, const void* data, int count);
/* buffer.count / buffer.stride                                       RM:237 */
URT_API int urt_buffer_get_info(urt_context* ctx, urt_handle buffer, int* out_count, int* out_stride);
/* buffer.Release()                                                   RM:195-210, 238 */
URT_API int urt_buffer_release(urt_context* ctx, urt_handle buffer);

/* ---- RenderTexture / Texture (RGBA32F = ARGBFloat, linear) -------------------------------- */
/* new RenderTexture(w, h, 0, ARGBFloat, Linear){enableRandomWrite}.Create()   RM:834-840 */
URT_API int urt_texture_create(urt_context* ctx, int width, int height, urt_handle* out_texture);
/* Same, over caller-owned device memory of width*height*16 bytes (e.g. a torch tensor that a
 * collective reads); the library never frees it. */
URT_API int urt_texture_create_external(urt_context* ctx, int width, int height, void* device_ptr,
                                        urt_handle* out_texture);
/* Upload / read back width*height*4 floats, row 0 = bottom.  (sky texture upload; readback stands
 * in for presenting _converged, RM:819.)  Both synchronise. */
URT_API int urt_texture_set_pixels(urt_context* ctx, urt_handle texture, const float* rgba);
URT_API int urt_texture_get_pixels(urt_context* ctx, urt_handle texture, float* rgba);
/* Pipelined readback for a host that LOOKS at every frame on the CPU (the reference presents on the GPU, RM:819; a host on another
 * device — or Unity's own `destination` behind the shim — needs the pixels in host memory).  urt_texture_get_pixels waits for the frame
 * and then for 33 MB over PCIe into pageable memory (2.2 ms at 1080p) before the next frame can start.  begin: submits the deferred
 * work, snapshots the image AS IT IS AT THIS POINT of the program order (a device copy on the render stream) and sends the snapshot to a
 * pinned host image on a copy stream of the library's own — it returns at once and later frames render while the snapshot travels;
 * end: waits for that one copy and hands out the pinned image (width x height x 4 floats, row 0 = bottom), valid until the third
 * urt_texture_read_begin after the one that made the ticket.  Up to three readbacks may be in flight.  A watchdog trip is reported by end. */
URT_API int urt_texture_read_begin(urt_context* ctx, urt_handle texture, uint64_t* out_ticket);
URT_API int urt_texture_read_end(urt_context* ctx, uint64_t ticket, const float** out_rgba);
/* The same readback in the FORMAT of the host's `destination` (RM:819 blits _converged into whatever the camera renders to): the image
 * is converted on the GPU (csrc/present.hip) and only the converted pixels cross the bus — 8.3 MB (RGBA8) or 16.6 MB (RGBA16F) per 1080p
 * frame instead of 33.2 MB.
 *   URT_FORMAT_RGBA32F     as urt_texture_read_begin
 *   URT_FORMAT_RGBA8_SRGB  bytes R, G, B, A per pixel: colour through the sRGB transfer function (the project renders in linear colour
 *                          space, ProjectSettings/ProjectSettings.asset:50, so a blit into an 8-bit back buffer encodes) — exactly the
 *                          bytes of urt_host_encode_srgb8 / urt_host_write_png; alpha as UNORM8
 *   URT_FORMAT_RGBA16F     four IEEE halfs per pixel, round to nearest even (a camera with allowHDR: ARGBHalf)
 * end_format: *out_pixels = the pinned host image (row 0 = bottom), *out_bytes (may be NULL) its size.  Tickets of converted images
 * must be ended with urt_texture_read_end_format. */
enum { URT_FORMAT_RGBA32F = 0, URT_FORMAT_RGBA8_SRGB = 1, URT_FORMAT_RGBA16F = 2 };
URT_API int urt_texture_read_begin_format(urt_context* ctx, urt_handle texture, int format, uint64_t* out_ticket);
URT_API int urt_texture_read_end_format(urt_context* ctx, uint64_t ticket, const void** out_pixels, size_t* out_bytes);
URT_API int urt_texture_get_info(urt_context* ctx, urt_handle texture, int* out_width, int* out_height,
                                 void** out_device_ptr);
/* texture.Release()                                                  RM:830-831 */
URT_API int urt_texture_release(urt_context* ctx, urt_handle texture);

/* ---- ComputeShader (RayTraceShader.compute; kernel index 0 = CSMain) ---------------------- */
/* shader.SetBuffer(0, name, buffer)                                  RM:255-259, names RM:787-794:
 *   _MeshObjects(112) _Vertices(12) _Indices(4) _Normals(12) _Spheres(56) _MeshBVH(28) _SphereBVH(28)
 * A name that was never bound counts as an empty buffer (RM:256 skips null; RS:375-379).
 * buffer == 0 unbinds. */
URT_API int urt_shader_set_buffer(urt_context* ctx, int kernel, const char* name, urt_handle buffer);
/* shader.SetTexture(0, "_SkyboxTexture" | "Result", tex)             RM:776, RM:803 */
URT_API int urt_shader_set_texture(urt_context* ctx, int kernel, const char* name, urt_handle texture);
/* shader.SetMatrix("_CameraToWorld" | "_CameraInverseProjection", m) RM:773-774 */
URT_API int urt_shader_set_matrix(urt_context* ctx, const char* name, const float* m16);
/* shader.SetVector("_PixelOffset", v)  (x,y used)                    RM:777 */
URT_API int urt_shader_set_vector(urt_context* ctx, const char* name, const float* v4);
/* shader.SetFloat("_Seed", v)                                        RM:778 */
URT_API int urt_shader_set_float(urt_context* ctx, const char* name, float value);
/* shader.SetInt("_numBounces" | "_numRays", v); "_MeshBVH_len" / "_SphereBVH_len" are accepted and
 * ignored (they are `static const` in the shader, RS:73-74).         RM:780-784
 * Names the shader does not declare are ignored, as Unity does. */
URT_API int urt_shader_set_int(urt_context* ctx, const char* name, int value);
/* shader.Dispatch(0, groupsX, groupsY, 1): one thread per pixel in 8x8 groups; threads outside the
 * Result texture write nothing.  Asynchronous, in order.            RM:806-810 -> RS:431-469 */
URT_API int urt_shader_dispatch(urt_context* ctx, int kernel, int groups_x, int groups_y, int groups_z);
/* Multi-GPU form: the same dispatch restricted to group rows first_group_row, +row_stride, ...
 * (8 pixel rows each).  Pixels keep their GLOBAL id.xy, so the union over ranks
 * r = 0..N-1 of dispatch_rows(r, N) is bit-identical to one full dispatch (RS:78,434). */
URT_API int urt_shader_dispatch_rows(urt_context* ctx, int kernel, int groups_x, int groups_y, int groups_z,
                                     int first_group_row, int row_stride);

/* ---- Graphics.Blit ------------------------------------------------------------------------ */
/* _additionMaterial.SetFloat("_Sample", sample); Graphics.Blit(src, dst, _additionMaterial):
 * dst = src * a + dst * (1 - a) with a = 1 / (sample + 1), all four channels   RM:817-818, AS:9,39-41 */
URT_API int urt_blit_add(urt_context* ctx, urt_handle src, urt_handle dst, float sample);
/* Graphics.Blit(src, dst): copy — the present of the accumulated frame      RM:819
 * While frames are deferred the copy is queued behind them (see "Frame batching" above). */
URT_API int urt_blit(urt_context* ctx, urt_handle src, urt_handle dst);
/* Strip helpers for the frame-end gather: copy the 8-row strips first_group_row, +row_stride, ... of
 * an image to/from a dense device buffer (strip-major).  out_bytes reports the packed size. */
URT_API int urt_texture_pack_rows(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride,
                                  void* device_dst, uint64_t* out_bytes);
URT_API int urt_texture_unpack_rows(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride,
                                    const void* device_src);
/* The same de-interleave issued on a CALLER-GIVEN stream (e.g. the communication stream the gather ran on), without touching
 * the context's own stream: the caller orders it (the image must not be in use by the context's queued work — a dedicated
 * gather target never is).  Lets rank 0 de-interleave frame i while its render stream is already on frame i+1. */
URT_API int urt_texture_unpack_rows_on(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride,
                                       const void* device_src, void* hip_stream);
/* The RGB forms (12 B per pixel instead of 16) for the gather of a RUNNING MEAN: AdditionShader blends the alpha channel like the
 * colours and its source alpha is a = 1 / (_Sample + 1) itself (AS:39-41), so after sample n the alpha of `_converged` is the same
 * value in every pixel — w_0 = a_0 a_0, w_n = a_n a_n + w_{n-1} (1 - a_n) in float32 — and need not travel: the ranks pack three
 * channels and the root passes that value as `alpha` when it de-interleaves.  hip_stream NULL = the context's own stream (deferred
 * and ordered like urt_texture_unpack_rows), else a caller-ordered stream like urt_texture_unpack_rows_on. */
URT_API int urt_texture_pack_rows_rgb(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride,
                                      void* device_dst, uint64_t* out_bytes);
URT_API int urt_texture_unpack_rows_rgb(urt_context* ctx, urt_handle texture, int first_group_row, int row_stride,
                                        const void* device_src, float alpha, void* hip_stream);

/* ---- measurement -------------------------------------------------------------------------- */
typedef struct urt_counters {
  uint64_t rays;          /* Trace() invocations (RS:454) — the unit of the Mrays/s metric */
  uint64_t tlas_nodes;    /* object-level BVHNode fetches, 28 B each (RS:306, RS:341) */
  uint64_t blas_nodes;    /* triangle-BVH node visits, 64 B each */
  uint64_t tri_tests;     /* Moller-Trumbore evaluations, 48 B each (RS:199-234) */
  uint64_t sphere_tests;  /* IntersectSphere evaluations, 16 B each (RS:175-196) */
  uint64_t hit_tri;       /* closest hits shaded on a triangle: 48 B normals + 40 B material */
  uint64_t hit_sphere;    /* closest hits shaded on a sphere: 40 B material */
  uint64_t hit_ground;    /* closest hits on the y = 0 plane */
  uint64_t hit_sky;       /* paths ended on the sky: 4 x 16 B texels */
  uint64_t pixels;        /* pixels written (16 B each) */
  uint64_t dispatches;    /* urt_shader_dispatch* calls since reset */
  float trace_ms;         /* GPU time of the trace kernels of those dispatches (HIP events; needs "time_dispatch") */
  uint32_t watchdog_trips; /* waves that hit a persistent kernel's iteration cap (scaled with frames x rays x bounces of the launch): 0 unless
                              there is a bug; when not, the next urt_synchronize / urt_texture_get_pixels fails with URT_ERR_WATCHDOG */
  uint64_t launches;      /* trace-kernel launches those dispatches became (< dispatches when frames were batched) */
} urt_counters;
/* Options: "blas_builder" (-1 = auto, the default: 0 for scenes of fewer than 200,000 triangles, 3 from there on;
 *                          0 = binned-SAH triangle BVH built on host threads: best trees; 1 = LBVH built on the GPU
 *                          from the uploaded buffers — Morton sort + Karras hierarchy, csrc/lbvh.hip: milliseconds instead of tens
 *                          of milliseconds for scenes whose objects move; 2 = the same radix tree built top-down within a depth
 *                          budget ("lbvh_slack", default 6 levels beyond a median tree): the traversal stacks live in LDS and a
 *                          30-level Karras tree costs workgroups per CU — frames on tree 2 cost +6 ... +10 % against the SAH trees
 *                          instead of +10 ... +26 %, the build 1 ms more on a million triangles; 3 = BINNED SAH ON THE GPU — the host builder's
 *                          algorithm level by level with atomics: frames as on the host's trees, 15 ms instead of 80 for a million
 *                          triangles; same pixels whichever builder),
 *          "front_cull" (0/1, default 1: object-level cull — a MeshObject whose heap-leaf box the ray passes, leaves behind or meets
 *                        beyond the ground-plane hit, by a margin, is not intersected although the reference would (RS:294-326 keeps
 *                        testing every popped leaf once `tests` is set; such an object cannot hold the closest hit).  Only leaves whose
 *                        box the library has verified to contain the object's triangles; 0 = the reference's literal work),
 *          "overlap_launches" (0 off / 1 auto (default) / 2 always: launches of up to 8 frames — a host that submits every frame on its own,
 *                              urt_flush or a present per frame — alternate between two trace streams of the library's own and take their
 *                              Result slots round-robin, so that the next launch fills the wave slots the draining waves of the previous one
 *                              give back (C3, one frame per submission: 0.61 -> 0.50 ms per frame); blends, presents and readbacks stay on
 *                              the context's stream in program order.  Auto: not while a pipelined readback is in flight — a host that
 *                              waits for finished frames gets them later when two launches share the chip (+6 % C3, +21 % C2 with two
 *                              tickets in flight).  Only on the library's own stream),
 *          "frames_per_launch" (0 = auto: own stream -> up to 64 frames per launch (fewer if their Result slots exceed 8 GiB), caller's stream -> 1;
 *                               1 = every dispatch is its own launch; 2..64 = batch that many, also on a caller's stream),
 *          "count_stats" (0/1: per-dispatch traversal counters, slower build of the kernel),
 *          "time_dispatch" (0/1: bracket each dispatch with HIP events, read by urt_get_counters),
 *          "kernel_mode" (0 = one thread per pixel; 1 = one launch per bounce over compacted path queues;
 *                         2 = persistent waves with in-wave path regeneration;
 *                         3 = 2 + lanes scheduled by phase inside the wave, the default;
 *                         4 = 3 with a pool of 64 x pool_k paths per wave kept in LDS, experimental;
 *                         5 = 3 with the triangle-BVH phase as a service shared by the four waves of a workgroup (rays are
 *                             posted to a mailbox, any wave claims and walks them; "serve_refill" 1..64, "blas_min" up to 256):
 *                             measured alternative, slower than 3),
 *          "block_threads" (64 | 128 | 256: modes 0-2), "xcd_run" (0 = auto: 1..8 by launch size | >= 1: consecutive tiles a work-counter shard hands out as one run), "work_shards" (1 | 2 | ... | 64: work counters in use), "frame_group" (1..64: frames of a batched launch whose tile runs are interleaved; default 64 = all),
 *          "tile_order" (0 bottom-up | 1 top-down: a launch ends with the rows nearest the ground plane | -1 auto, the default: top-down for scenes without triangle meshes), "waves_per_cu" (0 = auto, 1..32), "refill_min" (1..64),
 *          mode 3: "blas_min" / "blas_exit" (0 = auto by scene | 1..64) / "shade_min" / "sky_min" (1..64): vote thresholds, "shade_split" (-1 auto | 0 | 1: surface hits and
 *                  sky misses as separate phases), "sched_block" (0 auto | 64 | 256),
 *                  "top_nodes" (0..256 triangle-BVH nodes kept in LDS; -1 auto, the default: 64, or twice the number of big MeshObjects when the object-level phase is the masked one), "top_front" (-1 auto | 0 | 1: where that top is
 *                  walked), "front_list" (-1 auto | 0 | 1 | 2: multi-mesh scenes determine the objects a ray must test first and work them off in voted
 *                  trips — auto / 1: by mask arithmetic over the heap when it has <= 31 nodes, else as a list when <= 12 MeshObjects; 2: always the list), "lds_tlas" (0/1: small object-level tables in LDS),
 *          mode 4: "pool_k" (1..4), "pool_refill", "pool_blas_min" (1..256), "pool_blas_exit", "pool_inloop",
 *                  "pool_other_min" (1..64),
 *          "blas_leaf_max" (1..8: triangles per BVH leaf, default 2 or the environment variable URT_BLAS_LEAF_MAX read when the
 *          library is loaded — a process-wide builder setting; rebuilds the BVH),
 *          "stack_pad" (0..96: test hook, unused extra entries per traversal stack -> the > 64 KiB LDS launch path),
 *          "qnodes" (0 off, the default | 1 on | -1 on unless some MeshObject spans fewer than 1024 grid cells: the traversal loop of the default
 *                    kernel reads 32-byte quantized copies of the triangle-BVH nodes — two vector loads per node step instead of four; conservative
 *                    boxes on one 16-bit grid over the whole forest, csrc/qnodes.hip.  A measured alternative: same pixels, -1.3 % frame time on
 *                    single-mesh scenes, +1.3 % on C4 / C5: the loop waits on the latency of one dependent fetch per step, not on its width),
 *          "refit" (0/1, default 1: moved MeshObjects are refitted on the GPU instead of rebuilt — see urt_debug_refit_stats),
 *          "watchdog_cap" (test hook: scheduler trips a wave may make before it gives up; 0 = auto = 2^24 x frames of the launch x
 *                          max(1, numRays x numBounces / 8))
 *          — tuning knobs; they change speed only, never pixels. */
URT_API int urt_set_option(urt_context* ctx, const char* name, int value);
URT_API int urt_get_counters(urt_context* ctx, urt_counters* out);   /* synchronises */
URT_API int urt_reset_counters(urt_context* ctx);

/* ---- device groups: one host thread, N GPUs (SURVEY.md 8b "device list for 1/2/4/8 GPUs", 8e) ------------------------------
 * The reference issues everything from Unity's main thread to one GPU (RM:806-810).  A host that wants the frame
 * tile-partitioned over the GPUs of a node keeps its call sequence and swaps urt_* for urt_group_*:
 *  - scene buffers, uniforms, textures, options and Graphics.Blit calls are REPLICATED on every rank's context (one handle
 *    value names the same object on every rank);
 *  - urt_group_shader_dispatch gives rank r the 8-row strips r, r+N, ... with global pixel ids (= urt_shader_dispatch_rows):
 *    the union over the ranks is bit-identical to one full dispatch;
 *  - accumulation stays local to the ranks' strips; urt_group_gather is the ONE exchange per frame: strips of `src_texture`
 *    from every rank -> the full image `dst_texture` on rank 0 (pack kernels + hipMemcpyPeerAsync over xGMI, point-to-point
 *    to the root, ordered by events, no host synchronisation; queued behind the ranks' batched frames and submitted in
 *    bursts of up to 16);
 *  - readback, synchronize and counters act on the whole group (pixels come from rank 0).
 * The device list may repeat an ordinal (several ranks on one card).  urt_group_context exposes a rank's context for
 * per-rank inspection; objects must be created through the group. */
typedef struct urt_group urt_group;
URT_API int urt_group_create(const int* devices, int n_devices, urt_group** out_group);
URT_API int urt_group_destroy(urt_group* group);
URT_API int urt_group_size(urt_group* group);
URT_API urt_context* urt_group_context(urt_group* group, int rank);
URT_API const char* urt_group_last_error(urt_group* group);          /* group may be NULL for create failures */
URT_API int urt_group_buffer_create(urt_group* group, int count, int stride, urt_handle* out_buffer);            /* RM:247 */
URT_API int urt_group_buffer_set_data(urt_group* group, urt_handle buffer, const void* data, int count);          /* RM:250 */
URT_API int urt_group_buffer_release(urt_group* group, urt_handle buffer);
URT_API int urt_group_texture_create(urt_group* group, int width, int height, urt_handle* out_texture);          /* RM:834-840 */
URT_API int urt_group_texture_set_pixels(urt_group* group, urt_handle texture, const float* rgba);
URT_API int urt_group_texture_get_pixels(urt_group* group, urt_handle texture, float* rgba);                     /* rank 0's image */
URT_API int urt_group_texture_release(urt_group* group, urt_handle texture);
URT_API int urt_group_shader_set_buffer(urt_group* group, int kernel, const char* name, urt_handle buffer);      /* RM:787-794 */
URT_API int urt_group_shader_set_texture(urt_group* group, int kernel, const char* name, urt_handle texture);    /* RM:776, 803 */
URT_API int urt_group_shader_set_matrix(urt_group* group, const char* name, const float* m16);                   /* RM:773-774 */
URT_API int urt_group_shader_set_vector(urt_group* group, const char* name, const float* v4);                    /* RM:777 */
URT_API int urt_group_shader_set_float(urt_group* group, const char* name, float value);                         /* RM:778 */
URT_API int urt_group_shader_set_int(urt_group* group, const char* name, int value);                             /* RM:780-784 */
URT_API int urt_group_set_option(urt_group* group, const char* name, int value);
URT_API int urt_group_shader_dispatch(urt_group* group, int kernel, int groups_x, int groups_y, int groups_z);   /* RM:806-810, partitioned */
URT_API int urt_group_blit_add(urt_group* group, urt_handle src, urt_handle dst, float sample);                  /* RM:817-818 */
URT_API int urt_group_blit(urt_group* group, urt_handle src, urt_handle dst);                                    /* RM:819 */
URT_API int urt_group_gather(urt_group* group, urt_handle src_texture, urt_handle dst_texture);
URT_API int urt_group_flush(urt_group* group);
URT_API int urt_group_synchronize(urt_group* group);
URT_API int urt_group_get_counters(urt_group* group, urt_counters* out);   /* summed over the ranks; dispatches/launches/trace_ms = the largest rank's */
URT_API int urt_group_reset_counters(urt_group* group);

/* ---- host-side scene preparation (no GPU needed) ---------------------- */
/* RayTraceMaster.ComputeNormals (RM:340-368): per-vertex sum of the un-normalised face normals of every index slot whose
 * vertex POSITION equals this vertex's (weld across all meshes), in ascending slot order, then Vector3.Normalize —
 * O(V + I) by position hashing instead of the reference's O(V * I).  out_normals = 3 * n_vertices floats. */
URT_API int urt_host_compute_normals(const float* vertices, int n_vertices, const int32_t* indices, int n_indices,
                                     float* out_normals);
/* SetupBVHLeaves(List<MeshObject>) (RM:405-433): one world-space box per MeshObject (112-B records).  literal != 0 keeps
 * the reference's quirks (box seeded from _vertices[_indices[0]], the mesh's own first index slot skipped, A.7);
 * literal == 0 gives the tight box. */
URT_API int urt_host_mesh_leaf_bounds(const void* mesh_objects, int n_meshes, const float* vertices, int n_vertices,
                                      const int32_t* indices, int n_indices, int literal, urt_BVHNode* out_leaves);
/* SetupBVHLeaves(List<Sphere>) (RM:436-455): literal != 0 keeps the inverted boxes (vmin = pos + r, vmax = pos - r). */
URT_API int urt_host_sphere_leaf_bounds(const void* spheres, int n_spheres, int literal, urt_BVHNode* out_leaves);
/* CreateBVH's output contract (RM:681-722): implicit heap of 2^D - 1 nodes, D = ceil(log2 n) + 1, children 2i+1 / 2i+2,
 * interior and filler nodes index -1 (fillers all-zero, RM:490-494).  This one builds the tree by a deterministic median split
 * (O(n log n)); urt_host_build_object_bvh_pairing below is the reference's own O(n^3) pairing heuristic.  Pixels do not depend on
 * which of the two made the heap. */
URT_API int urt_host_object_bvh_length(int n_objects);
URT_API int urt_host_build_object_bvh(const urt_BVHNode* leaves, int n_objects, urt_BVHNode* out_nodes, int capacity);
/* The reference's OWN builder restated literally — SetupBVHRankList / PairBVHBounds / JoinBVH / CreateBVH (RM:459-722): nearest-
 * neighbour ranking with "forbidden" pairs last, greedy pairing from start index n-1 (the volume comparison is dead code: the
 * candidate list is aliased, RM:670), lone trees joined under a copy of their own root, layers woven into the implicit heap.
 * Only the order of exact ties in the ranking sort (an unstable List.Sort in the reference) is this library's choice (stable). */
URT_API int urt_host_build_object_bvh_pairing(const urt_BVHNode* leaves, int n_objects, urt_BVHNode* out_nodes, int capacity);
URT_API const char* urt_host_last_error(void);

/* ---- host-side image I/O (no GPU needed) ---------------------------------- */
/* Radiance RGBE (.hdr, FORMAT=32-bit_rle_rgbe, flat or new-style RLE scanlines) -> RGBA32F, row 0 = bottom: the pixels a
 * caller hands to urt_texture_set_pixels for "_SkyboxTexture" (RM:776).  out_rgba == NULL queries the size. */
URT_API int urt_host_load_hdr(const char* path, int* out_width, int* out_height, float* out_rgba, size_t capacity_floats);
/* RGBA32F image -> RGBA32F image of another size with a separable Mitchell-Netravali filter: the importer step the reference's sky
 * assets go through (`maxTextureSize: 2048`, `resizeAlgorithm: 0`, Assets/Skyboxes/CloudedSunGlow4k.hdr.meta:36,73).  An approximation
 * of Unity's closed-source resampler (its BC6H compression is not reproduced). */
URT_API int urt_host_resize_rgba(const float* src, int width, int height, float* dst, int new_width, int new_height);
/* RGBA32F image (row 0 = bottom) -> .pfm (float RGB, bottom row first). */
URT_API int urt_host_write_pfm(const char* path, const float* rgba, int width, int height);
/* RGBA32F linear image -> 8-bit sRGB .png (what ScreenCapture.CaptureScreenshot produces for RM:762). */
URT_API int urt_host_write_png(const char* path, const float* rgba, int width, int height);
/* The PNG writer's pixel encoding on its own: RGBA32F linear -> RGBA8 (colour: sRGB transfer function, clamped, NaN -> 0; alpha: UNORM8),
 * and the encoder as the step function it is: out256[k] = the smallest float whose code is >= k (out256[0] = -inf) — the table the GPU
 * encoder of urt_texture_read_begin_format searches. */
URT_API int urt_host_encode_srgb8(const float* rgba, size_t n_pixels, unsigned char* out_rgba8);
URT_API int urt_host_srgb8_first_floats(float* out256);
URT_API const char* urt_host_io_last_error(void);

/* ---- host-side debug log and BVH inspection (no GPU needed) ------------------ */
/* RayTraceDebug.Log (RD:25-36): appends `text` + newline to the file when level <= debug_level.  Returns URT_OK when
 * written, 1 when filtered by the level (the reference's Log returns 1 then as well). */
URT_API int urt_host_log(const char* path, int debug_level, int level, const char* text);
/* The five count lines of RebuildObjectLists (RM:331-335): "# of Spheres: n", "# of Mesh Objects: n", ... at level 2. */
URT_API int urt_host_log_scene_counts(const char* path, int debug_level, int n_spheres, int n_mesh_objects, int n_vertices,
                                      int n_indices, int n_normals);
/* The two reports of RebuildTrees (RM:731-735): amount, depth, complete length 2^depth - 1, real length — at level 2. */
URT_API int urt_host_log_tree_report(const char* path, int debug_level, int n_mesh_objects, int mesh_depth, int mesh_real_length,
                                     int n_spheres, int sphere_depth, int sphere_real_length);
/* Text stand-in for the editor gizmos of RayTraceDebug.DrawBVH (RD:92-117): walks `depth` levels of the implicit heap from the
 * root (children 2i+1 / 2i+2, pre-order) and writes one line per node: "(position in list, object index)" as the gizmo labels
 * it (RD:108), the box, its centre, and " [ray]" when the test segment start -> end passes RD's CPU slab test (RD:70-89; the
 * gizmo paints those boxes black).  ray_start3 / ray_end3 may both be NULL.  out_lines = nodes written. */
URT_API int urt_host_dump_bvh(const char* path, const urt_BVHNode* nodes, int n_nodes, int depth, const float* ray_start3,
                              const float* ray_end3, int* out_lines);
/* Text stand-in for RayTraceDebug.DrawNormals (RD:165-183): per index slot of every MeshObject the gizmo's base point
 * MultiplyPoint3x4(_vertices[_indices[i]]) and the tip of its line MultiplyPoint3x4(_vertices[_indices[i]] + _normals[_indices[i]] * 0.1f),
 * one line "mesh slot (base) -> (tip)".  out_lines = slots written. */
URT_API int urt_host_dump_normals(const char* path, const void* mesh_objects, int n_meshes, const float* vertices, int n_vertices,
                                  const int32_t* indices, int n_indices, const float* normals, int n_normals, int* out_lines);
URT_API const char* urt_host_debug_last_error(void);

/* ---- introspection for tests (host only, no GPU needed) ------------------------------------ */
/* Run the library's triangle-BVH builder — the one urt_shader_dispatch uses — over host copies of
 * _MeshObjects (112 B records), _Vertices, _Indices, and keep the result in a process-wide cache.
 * Reports the sizes.  tests/ use it to validate the BVH and to hand it to the oracle's culled mode. */
URT_API int urt_debug_build_blas(const void* mesh_objects, int n_meshes, const float* vertices, int n_vertices,
                                 const int32_t* indices, int n_indices, int* out_n_nodes, int* out_n_tris,
                                 int* out_max_depth);
/* Copy the cached BVH out: nodes = n_nodes * 16 floats (layout in DESIGN.md "BLAS"); tri_index = n_tris
 * index-slot numbers (i of RS:243) in leaf order; mesh_root / mesh_first_tri = one entry per MeshObject.
 * Any pointer may be NULL. */
URT_API int urt_debug_get_blas(float* nodes, int32_t* tri_index, int32_t* mesh_root, int32_t* mesh_first_tri);
/* The triangle BVH of the context's CURRENT device scene, whichever builder made it ("blas_builder" option): prepares the
 * scene if it is stale (as the next dispatch would).  scene_info reports sizes, the depth of the deepest leaf and the host
 * wall time the last preparation took; read_scene_blas copies the nodes (n_nodes x 16 floats), the index slot of every
 * leaf-order triangle and the MeshObject roots back from the GPU.  Any pointer may be NULL. */
URT_API int urt_debug_scene_info(urt_context* ctx, int* out_n_nodes, int* out_n_tris, int* out_max_depth, float* out_prepare_ms);
/* The last trace launch of this context (deferred frames are submitted first): which kernel instantiation ran — by the name rocprofv3
 * prints for it, so that bench.py and the profile reducers name the kernel that really ran instead of re-deriving the dispatch logic —,
 * its grid, its dynamic LDS and what the frame batching did (slab_frames_max < the requested batch, or slab_out_of_memory: the Result
 * slots did not fit and the batch was halved / switched off, context.cpp ensure_slab).  No counterpart in the reference (one Dispatch
 * per frame, RM:806-810); measurement only. */
typedef struct urt_launch_info {
  char kernel[96];            /* e.g. "k_sched<false, 256, 0, false, false>" (COUNT, BLOCK, FMODE, MULTI, QN: kernels.hip) */
  int kernel_mode;            /* option "kernel_mode" the launch ran under (a degenerate dispatch runs mode 0) */
  int front_mode;             /* kernel_mode 3 / 5: 0 one mesh, 1 BVH top walked in the object-level phase, 2 listed, 3 masked */
  int count_stats;            /* the counting instantiation */
  int n_blocks, block_threads, lds_bytes, waves_per_cu;
  int n_frames, frame_group, xcd_run, tile_order, top_nodes, tlas_stack, blas_stack;
  int lds_tables;             /* bit 0 mesh heap, 1 sphere heap + spheres, 2 single-leaf triangle records, 3 walk table */
  int slab_frames, slab_frames_max, slab_out_of_memory;
  int experiment;             /* 1: the library is an A/B / probe / diagnostic build (negative urt_abi_version) */
  int blas_builder;           /* the triangle-BVH builder the scene's last full preparation used (0..3; what "blas_builder" -1 = auto resolved to) */
  int trace_stream;           /* 0: launched on the context's stream; 1 / 2: on one of the library's two trace streams (option "overlap_launches") */
  int slab_base;              /* first Result slot of the launch */
  int overlapped;             /* 1: the launch did not wait for the previous launch (it waited for the main stream as of the previous submission) */
  int overlapped_launches;    /* such launches since the context was created */
} urt_launch_info;
URT_API int urt_debug_launch_info(urt_context* ctx, urt_launch_info* out_info);
URT_API int urt_debug_read_scene_blas(urt_context* ctx, float* nodes, int32_t* tri_index, int32_t* mesh_root);
/* Scene preparation of a context keeps the triangle BVH of every MeshObject and reuses it at the next preparation when the
 * MeshObject's matrix and the positions behind its index slots are unchanged (the reference re-uploads every buffer when
 * any object moves, RM:262-336).  Reports how many MeshObject BVHs were reused / built since the context was created. */
URT_API int urt_debug_blas_cache_stats(urt_context* ctx, uint64_t* out_reused, uint64_t* out_built);
/* The walk table the default kernel's masked object-level phase uses for a mesh heap of <= 31 nodes (DESIGN.md §5 "masked FRONT"): header
 * words (n_eval, levels, interior mask, exist mask | leaf_any, leaf_valid | 4 depth masks | 4 left-child shifts), 32 x {root, small_first}
 * per heap position in pop order, then n_eval x {vmin.xyz, parent bit | vmax.xyz, position bit}.  out_words = 0 when the heap does not
 * qualify (empty or > 31 nodes).  Host only: tests/test_walk_table.py checks the mask arithmetic against the literal walk RS:294-326. */
URT_API int urt_debug_build_walk_table(const urt_BVHNode* heap, int n_nodes, int n_meshes, const int32_t* mesh_root, const int32_t* small_first,
                                       float* out, int capacity_words, int* out_words);
/* Dynamic scenes.  When the only buffer CONTENTS that changed since the scene was prepared are those of _MeshObjects, _MeshBVH, _Spheres
 * and _SphereBVH (same counts, same index range per MeshObject; data equal to what a buffer already holds counts as unchanged — the
 * reference re-uploads every list when one object moves, RM:215-230 -> 262-336), the device scene is updated in place: materials and
 * object-level tables are re-packed, and MeshObjects whose localToWorldMatrix changed keep the topology of their triangle BVH — their
 * triangle records and boxes are recomputed on the GPU (csrc/refit.hip; option "refit" = 0 turns this off).  Reports MeshObjects
 * refitted and in-place preparations since the context was created. */
URT_API int urt_debug_refit_stats(urt_context* ctx, uint64_t* out_refitted_meshes, uint64_t* out_incremental_preparations);
/* kernel_mode 5 with "count_stats" = 1: what the shared traversal service did since the last urt_reset_counters —
 * out6 = visits of the service, its trips, active lanes summed over the trips, claim rounds, rays claimed, rays suspended. */
URT_API int urt_debug_serve_stats(urt_context* ctx, unsigned long long* out6);

#ifdef __cplusplus
}
#endif
