// group.cpp — urt_group_*: ONE host thread drives N GPUs through the C ABI (SURVEY.md §8b "context create/destroy (device
// list for 1/2/4/8 GPUs)", §8e).  The reference is single-GPU (one Dispatch per frame from Unity's main thread, RM:806-810);
// a P/Invoke host that wants the frame tile-partitioned over the GPUs of a node keeps ITS call sequence and swaps urt_* for
// urt_group_*:
//   * scene, uniforms, textures and the AdditionShader blit are REPLICATED: every call is forwarded to every rank's context.
//     Contexts hand out handles from the same deterministic counter, so one handle value names the same object on every
//     rank (checked);
//   * urt_group_shader_dispatch gives rank r the 8-row strips r, r+N, ... with GLOBAL pixel ids (urt_shader_dispatch_rows):
//     the union of the ranks' strips is bit-identical to one full dispatch (RS:78,434);
//   * accumulation stays local; urt_group_gather is the ONE exchange per frame: every rank packs its strips of an image
//     (k_pack_rows), the peers' strips travel to rank 0 with hipMemcpyPeerAsync — point-to-point over xGMI, each peer on its own
//     link to the root, no ring — and rank 0 de-interleaves them into a full image.  No host synchronisation: streams are
//     ordered by events.  Gathers are queued behind the ranks' deferred (batched) frames and submitted in bursts, so the
//     per-rank frame batching of context.cpp keeps working under a gather-every-frame protocol.
// The device list may name one ordinal several times (N contexts on one card): that is how the path is tested on a 1-GPU box.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/urt.h"
#include "context_internal.h"

struct urt_group {
  std::vector<urt_context*> ctx;
  std::string err;
  // gather staging: slot s of rank r = that rank's packed strips of pending gather s
  static constexpr int kSlots = 16;
  size_t stage_bytes = 0;                              // per rank and slot (packed strips of the largest rank)
  std::vector<void*> stage;                            // [rank] on that rank's device: kSlots x stage_bytes
  std::vector<void*> recv;                             // [rank] on rank 0's device:   kSlots x stage_bytes
  std::vector<std::vector<hipEvent_t>> ev_copy;        // [rank][slot] the peer copy of that slot has landed on rank 0
  std::vector<std::vector<hipEvent_t>> ev_free;        // [rank][slot] rank 0 has unpacked it (the slot may be reused)
  // gathers whose packs are queued but whose copies are not issued yet (kind 0, staging slot `slot`), and — in program order
  // between them — the plain blits that read or write an image such a gather writes (kind 1: the present of the gathered image,
  // RM:819): they must run after the unpack that is still to be issued
  struct PendingGather { int kind; urt_handle src, dst; int slot; };
  std::vector<PendingGather> pending;
  int pending_gathers = 0;
  uint64_t gathers = 0;
};

namespace {

std::string g_group_create_error;

int gfail(urt_group* g, int code, const std::string& msg) {
  if (g) g->err = msg; else g_group_create_error = msg;
  return code;
}

int rank_fail(urt_group* g, int rank, int code) {
  return gfail(g, code, "rank " + std::to_string(rank) + ": " + urt_last_error(g->ctx[(size_t)rank]));
}

#define GROUP_HIP(g, expr)                                                                              \
  do {                                                                                                  \
    hipError_t e__ = (expr);                                                                            \
    if (e__ != hipSuccess) return gfail(g, e__ == hipErrorOutOfMemory ? URT_ERR_OUT_OF_MEMORY : URT_ERR_HIP, \
                                        std::string(#expr) + ": " + hipGetErrorString(e__));            \
  } while (0)

// forward one call to every rank; the first failure is reported with its rank
#define FORWARD(g, call)                                                                                \
  do {                                                                                                  \
    if (!(g)) return gfail(nullptr, URT_ERR_INVALID_ARGUMENT, "group is NULL");                         \
    for (size_t r__ = 0; r__ < (g)->ctx.size(); r__++) {                                                \
      urt_context* c = (g)->ctx[r__];                                                                   \
      int rc__ = (call);                                                                                \
      if (rc__ != URT_OK) return rank_fail((g), (int)r__, rc__);                                        \
    }                                                                                                   \
    return URT_OK;                                                                                      \
  } while (0)

void free_staging(urt_group* g) {
  for (size_t r = 0; r < g->ctx.size(); r++) {
    if (r < g->stage.size() && g->stage[r]) { (void)hipSetDevice(urtd::context_device(g->ctx[r])); (void)hipFree(g->stage[r]); }
    if (r < g->recv.size() && g->recv[r]) { (void)hipSetDevice(urtd::context_device(g->ctx[0])); (void)hipFree(g->recv[r]); }
  }
  g->stage.clear(); g->recv.clear(); g->stage_bytes = 0;
}

int flush_gathers(urt_group* g);

int ensure_staging(urt_group* g, size_t bytes) {
  if (bytes <= g->stage_bytes) return URT_OK;
  { int rc = flush_gathers(g); if (rc) return rc; }                      // queued gathers have packed into the old buffers: copy + unpack them first
  for (urt_context* c : g->ctx) { int rc = urt_synchronize(c); if (rc) return gfail(g, rc, urt_last_error(c)); }
  free_staging(g);
  size_t n = g->ctx.size();
  g->stage.assign(n, nullptr); g->recv.assign(n, nullptr);
  for (size_t r = 0; r < n; r++) {
    GROUP_HIP(g, hipSetDevice(urtd::context_device(g->ctx[r])));
    GROUP_HIP(g, hipMalloc(&g->stage[r], bytes * urt_group::kSlots));
    GROUP_HIP(g, hipSetDevice(urtd::context_device(g->ctx[0])));
    GROUP_HIP(g, hipMalloc(&g->recv[r], bytes * urt_group::kSlots));
  }
  g->stage_bytes = bytes;
  return URT_OK;
}

// Submit the queued gathers: every rank's deferred frames + packs go to its stream, then the strips travel to rank 0
// (one peer copy per rank and gather) and rank 0 de-interleaves them, all ordered by events.
int flush_gathers(urt_group* g) {
  if (g->pending.empty()) return URT_OK;
  size_t n = g->ctx.size();
  urt_context* root = g->ctx[0];
  int dev0 = urtd::context_device(root);
  std::vector<urt_group::PendingGather> todo;
  todo.swap(g->pending);
  g->pending_gathers = 0;
  for (size_t r = 0; r < n; r++) {
    urt_context* c = g->ctx[r];
    int rc = urt_flush(c);                                              // the packs were queued behind the frames they read
    if (rc) return rank_fail(g, (int)r, rc);
    int dev = urtd::context_device(c);
    hipStream_t st = urtd::context_stream(c);
    GROUP_HIP(g, hipSetDevice(dev));
    for (const urt_group::PendingGather& op : todo) {
      if (op.kind == 0) {
        char* src = (char*)g->stage[r] + (size_t)op.slot * g->stage_bytes;
        char* dst = (char*)g->recv[r] + (size_t)op.slot * g->stage_bytes;
        GROUP_HIP(g, hipMemcpyPeerAsync(dst, dev0, src, dev, g->stage_bytes, st));   // xGMI point-to-point (or on-device when the ordinals coincide)
        GROUP_HIP(g, hipEventRecord(g->ev_copy[r][(size_t)op.slot], st));
      } else if (r != 0) {
        rc = urt_blit(c, op.src, op.dst);                               // replicated like every blit (only rank 0's copy shows the gathered image)
        if (rc) return rank_fail(g, (int)r, rc);
      }
    }
  }
  GROUP_HIP(g, hipSetDevice(dev0));
  hipStream_t st0 = urtd::context_stream(root);
  for (const urt_group::PendingGather& op : todo) {
    if (op.kind == 1) {                                                 // in program order after the unpacks before it
      int rc = urt_blit(root, op.src, op.dst);
      if (rc) return rank_fail(g, 0, rc);
      continue;
    }
    for (size_t r = 0; r < n; r++) {
      GROUP_HIP(g, hipStreamWaitEvent(st0, g->ev_copy[r][(size_t)op.slot], 0));
      int rc = urt_texture_unpack_rows(root, op.dst, (int)r, (int)n, (char*)g->recv[r] + (size_t)op.slot * g->stage_bytes);
      if (rc) return rank_fail(g, 0, rc);
      GROUP_HIP(g, hipEventRecord(g->ev_free[r][(size_t)op.slot], st0));
    }
  }
  return URT_OK;
}

// does a queued gather (or a blit queued with them) read or write one of these images?
bool touches_pending(const urt_group* g, urt_handle a, urt_handle b) {
  for (const urt_group::PendingGather& op : g->pending)
    if (op.src == a || op.dst == a || (b && (op.src == b || op.dst == b))) return true;
  return false;
}

// is one of these images WRITTEN by work that is still queued here — a gather's late unpack (its dst) or a blit queued behind one?
// Only such images force a blit into the queue: what reads or writes a gather's SRC is ordered by the rank contexts' own deferred
// operations (the pack was queued there in program order), and queuing it would run it after frames dispatched later.
bool written_by_pending(const urt_group* g, urt_handle a, urt_handle b) {
  for (const urt_group::PendingGather& op : g->pending)
    if (op.dst == a || (b && op.dst == b)) return true;
  return false;
}

}  // namespace

extern "C" {

const char* urt_group_last_error(urt_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int urt_group_create(const int* devices, int n_devices, urt_group** out_group) {
  if (!out_group) return gfail(nullptr, URT_ERR_INVALID_ARGUMENT, "out_group is NULL");
  *out_group = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return gfail(nullptr, URT_ERR_INVALID_ARGUMENT, "device list must name 1..64 devices");
  urt_group* g = new (std::nothrow) urt_group();
  if (!g) return gfail(nullptr, URT_ERR_OUT_OF_MEMORY, "host allocation failed");
  for (int r = 0; r < n_devices; r++) {
    urt_context* c = nullptr;
    int rc = urt_context_create(devices[r], &c);
    if (rc) { std::string m = urt_last_error(nullptr); urt_group_destroy(g); return gfail(nullptr, rc, "rank " + std::to_string(r) + ": " + m); }
    g->ctx.push_back(c);
  }
  // peer access to the root for the frame-end gather (already-enabled and same-device cases are fine)
  int dev0 = devices[0];
  for (int r = 1; r < n_devices; r++) {
    if (devices[r] == dev0) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, devices[r], dev0) == hipSuccess && can) {
      (void)hipSetDevice(devices[r]);
      hipError_t e = hipDeviceEnablePeerAccess(dev0, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // the copy engine falls back to staging through the host
    }
  }
  g->ev_copy.resize((size_t)n_devices); g->ev_free.resize((size_t)n_devices);
  for (int r = 0; r < n_devices; r++) {
    for (int s = 0; s < urt_group::kSlots; s++) {
      hipEvent_t a = nullptr, b = nullptr;
      (void)hipSetDevice(devices[r]);
      if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess) { urt_group_destroy(g); return gfail(nullptr, URT_ERR_HIP, "hipEventCreate failed"); }
      (void)hipSetDevice(dev0);
      if (hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(a); urt_group_destroy(g); return gfail(nullptr, URT_ERR_HIP, "hipEventCreate failed"); }
      g->ev_copy[(size_t)r].push_back(a); g->ev_free[(size_t)r].push_back(b);
    }
  }
  *out_group = g;
  return URT_OK;
}

int urt_group_destroy(urt_group* g) {
  if (!g) return URT_OK;
  for (urt_context* c : g->ctx) (void)urt_synchronize(c);
  for (auto& v : g->ev_copy) for (hipEvent_t e : v) (void)hipEventDestroy(e);
  for (auto& v : g->ev_free) for (hipEvent_t e : v) (void)hipEventDestroy(e);
  free_staging(g);
  for (urt_context* c : g->ctx) (void)urt_context_destroy(c);
  delete g;
  return URT_OK;
}

int urt_group_size(urt_group* g) { return g ? (int)g->ctx.size() : 0; }

urt_context* urt_group_context(urt_group* g, int rank) {
  if (!g || rank < 0 || rank >= (int)g->ctx.size()) return nullptr;
  return g->ctx[(size_t)rank];
}

/* ---- replicated calls ---- */
static int create_replicated(urt_group* g, urt_handle* out, const char* what, int (*make)(urt_context*, void*, urt_handle*), void* arg) {
  if (!g) return gfail(nullptr, URT_ERR_INVALID_ARGUMENT, "group is NULL");
  if (!out) return gfail(g, URT_ERR_INVALID_ARGUMENT, "out handle is NULL");
  urt_handle first = 0;
  for (size_t r = 0; r < g->ctx.size(); r++) {
    urt_handle h = 0;
    int rc = make(g->ctx[r], arg, &h);
    if (rc) return rank_fail(g, (int)r, rc);
    if (r == 0) first = h;
    else if (h != first)
      return gfail(g, URT_ERR_INVALID_HANDLE, std::string(what) + ": the ranks' contexts handed out different handles — objects were created on a "
                                              "single context behind the group's back");
  }
  *out = first;
  return URT_OK;
}

int urt_group_buffer_create(urt_group* g, int count, int stride, urt_handle* out_buffer) {
  int args[2] = {count, stride};
  return create_replicated(g, out_buffer, "buffer_create", [](urt_context* c, void* a, urt_handle* h) { return urt_buffer_create(c, ((int*)a)[0], ((int*)a)[1], h); }, args);
}
int urt_group_buffer_set_data(urt_group* g, urt_handle buffer, const void* data, int count) { FORWARD(g, urt_buffer_set_data(c, buffer, data, count)); }
int urt_group_buffer_release(urt_group* g, urt_handle buffer) { FORWARD(g, urt_buffer_release(c, buffer)); }
int urt_group_texture_create(urt_group* g, int width, int height, urt_handle* out_texture) {
  int args[2] = {width, height};
  return create_replicated(g, out_texture, "texture_create", [](urt_context* c, void* a, urt_handle* h) { return urt_texture_create(c, ((int*)a)[0], ((int*)a)[1], h); }, args);
}
int urt_group_texture_set_pixels(urt_group* g, urt_handle texture, const float* rgba) {
  if (g && touches_pending(g, texture, 0)) { int rc = flush_gathers(g); if (rc) return rc; }   // a queued unpack must not land on top of these pixels
  FORWARD(g, urt_texture_set_pixels(c, texture, rgba));
}
int urt_group_texture_release(urt_group* g, urt_handle texture) {
  if (g) { int rc = flush_gathers(g); if (rc) return rc; }
  FORWARD(g, urt_texture_release(c, texture));
}
int urt_group_shader_set_buffer(urt_group* g, int kernel, const char* name, urt_handle buffer) { FORWARD(g, urt_shader_set_buffer(c, kernel, name, buffer)); }
int urt_group_shader_set_texture(urt_group* g, int kernel, const char* name, urt_handle texture) { FORWARD(g, urt_shader_set_texture(c, kernel, name, texture)); }
int urt_group_shader_set_matrix(urt_group* g, const char* name, const float* m16) { FORWARD(g, urt_shader_set_matrix(c, name, m16)); }
int urt_group_shader_set_vector(urt_group* g, const char* name, const float* v4) { FORWARD(g, urt_shader_set_vector(c, name, v4)); }
int urt_group_shader_set_float(urt_group* g, const char* name, float value) { FORWARD(g, urt_shader_set_float(c, name, value)); }
int urt_group_shader_set_int(urt_group* g, const char* name, int value) { FORWARD(g, urt_shader_set_int(c, name, value)); }
int urt_group_set_option(urt_group* g, const char* name, int value) {
  if (g) { int rc = flush_gathers(g); if (rc) return rc; }
  FORWARD(g, urt_set_option(c, name, value));
}
int urt_group_blit_add(urt_group* g, urt_handle src, urt_handle dst, float sample) {
  if (g && touches_pending(g, src, dst)) { int rc = flush_gathers(g); if (rc) return rc; }     // in order with the unpack that is still to be issued
  FORWARD(g, urt_blit_add(c, src, dst, sample));
}
int urt_group_blit(urt_group* g, urt_handle src, urt_handle dst) {
  if (g && written_by_pending(g, src, dst)) {
    // the present of a gathered image (gather(_converged -> full); Blit(full, destination), RM:819): queued with the gathers, it
    // runs right after the unpack it depends on — the ranks keep batching their frames
    g->pending.push_back(urt_group::PendingGather{1, src, dst, 0});
    return URT_OK;
  }
  FORWARD(g, urt_blit(c, src, dst));
}

/* ---- the partitioned dispatch and the frame-end gather ---- */
int urt_group_shader_dispatch(urt_group* g, int kernel, int groups_x, int groups_y, int groups_z) {
  if (!g) return gfail(nullptr, URT_ERR_INVALID_ARGUMENT, "group is NULL");
  int n = (int)g->ctx.size();
  for (int r = 0; r < n; r++) {
    int rc = urt_shader_dispatch_rows(g->ctx[(size_t)r], kernel, groups_x, groups_y, groups_z, r, n);
    if (rc) return rank_fail(g, r, rc);
  }
  return URT_OK;
}

int urt_group_gather(urt_group* g, urt_handle src_texture, urt_handle dst_texture) {
  if (!g) return gfail(nullptr, URT_ERR_INVALID_ARGUMENT, "group is NULL");
  size_t n = g->ctx.size();
  int w = 0, h = 0, w2 = 0, h2 = 0;
  int rc = urt_texture_get_info(g->ctx[0], src_texture, &w, &h, nullptr);
  if (rc) return rank_fail(g, 0, rc);
  rc = urt_texture_get_info(g->ctx[0], dst_texture, &w2, &h2, nullptr);
  if (rc) return rank_fail(g, 0, rc);
  if (w != w2 || h != h2 || src_texture == dst_texture) return gfail(g, URT_ERR_INVALID_ARGUMENT, "gather: source and destination must be two images of the same size");
  uint64_t bytes = 0;
  rc = urt_texture_pack_rows(g->ctx[0], src_texture, 0, (int)n, nullptr, &bytes);     // rank 0 has the most strips
  if (rc) return rank_fail(g, 0, rc);
  if ((rc = ensure_staging(g, (size_t)bytes))) return rc;
  size_t slot = (size_t)g->pending_gathers;
  for (size_t r = 0; r < n; r++) {
    urt_context* c = g->ctx[r];
    // the slot's previous use must have been unpacked on rank 0 before this rank's next peer copy overwrites the receive
    // buffer (a never-recorded event counts as complete)
    GROUP_HIP(g, hipSetDevice(urtd::context_device(c)));
    GROUP_HIP(g, hipStreamWaitEvent(urtd::context_stream(c), g->ev_free[r][slot], 0));
    rc = urt_texture_pack_rows(c, src_texture, (int)r, (int)n, (char*)g->stage[r] + slot * g->stage_bytes, nullptr);   // deferred behind the rank's batched frames
    if (rc) return rank_fail(g, (int)r, rc);
  }
  g->pending.push_back(urt_group::PendingGather{0, src_texture, dst_texture, (int)slot});
  g->pending_gathers++;
  g->gathers++;
  // submit when the burst is full, or at once when the ranks are not deferring frames (then there is nothing to wait for)
  bool deferring = false;
  for (urt_context* c : g->ctx) deferring = deferring || urtd::context_pending_frames(c) > 0;
  if (!deferring || g->pending_gathers >= urt_group::kSlots) return flush_gathers(g);
  return URT_OK;
}

int urt_group_flush(urt_group* g) {
  if (!g) return gfail(nullptr, URT_ERR_INVALID_ARGUMENT, "group is NULL");
  int rc = flush_gathers(g);
  if (rc) return rc;
  FORWARD(g, urt_flush(c));
}

int urt_group_synchronize(urt_group* g) {
  if (!g) return gfail(nullptr, URT_ERR_INVALID_ARGUMENT, "group is NULL");
  int rc = flush_gathers(g);
  if (rc) return rc;
  // peers first (their copies feed rank 0's stream), the root last
  for (size_t r = g->ctx.size(); r-- > 0;) { rc = urt_synchronize(g->ctx[r]); if (rc) return rank_fail(g, (int)r, rc); }
  return URT_OK;
}

int urt_group_texture_get_pixels(urt_group* g, urt_handle texture, float* rgba) {
  if (!g) return gfail(nullptr, URT_ERR_INVALID_ARGUMENT, "group is NULL");
  int rc = flush_gathers(g);
  if (rc) return rc;
  for (size_t r = g->ctx.size(); r-- > 1;) { rc = urt_synchronize(g->ctx[r]); if (rc) return rank_fail(g, (int)r, rc); }
  rc = urt_texture_get_pixels(g->ctx[0], texture, rgba);
  return rc ? rank_fail(g, 0, rc) : URT_OK;
}

int urt_group_get_counters(urt_group* g, urt_counters* out) {
  if (!g) return gfail(nullptr, URT_ERR_INVALID_ARGUMENT, "group is NULL");
  if (!out) return gfail(g, URT_ERR_INVALID_ARGUMENT, "out is NULL");
  int rc = flush_gathers(g);
  if (rc) return rc;
  std::memset(out, 0, sizeof *out);
  for (size_t r = 0; r < g->ctx.size(); r++) {
    urt_counters c;
    rc = urt_get_counters(g->ctx[r], &c);
    if (rc) return rank_fail(g, (int)r, rc);
    out->rays += c.rays; out->tlas_nodes += c.tlas_nodes; out->blas_nodes += c.blas_nodes; out->tri_tests += c.tri_tests;
    out->sphere_tests += c.sphere_tests; out->hit_tri += c.hit_tri; out->hit_sphere += c.hit_sphere; out->hit_ground += c.hit_ground;
    out->hit_sky += c.hit_sky; out->pixels += c.pixels; out->watchdog_trips += c.watchdog_trips;
    out->dispatches = c.dispatches > out->dispatches ? c.dispatches : out->dispatches;     // frames, not frames x ranks
    out->launches = c.launches > out->launches ? c.launches : out->launches;
    out->trace_ms = c.trace_ms > out->trace_ms ? c.trace_ms : out->trace_ms;                // the ranks run concurrently: the slowest one
  }
  return URT_OK;
}

int urt_group_reset_counters(urt_group* g) {
  if (g) { int rc = flush_gathers(g); if (rc) return rc; }
  FORWARD(g, urt_reset_counters(c));
}

}  // extern "C"
