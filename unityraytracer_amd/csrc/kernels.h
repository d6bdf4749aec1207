// kernels.h — host-callable launchers of the HIP kernels in kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include "urt_device.h"

namespace urtd {

// the last trace launch made through the launchers below (process-wide, written on the launching host thread): the kernel instantiation by
// the name rocprofv3 prints for it, its grid and its dynamic LDS — what urt_debug_launch_info reports, so that tools name the kernel that
// really ran instead of re-deriving the dispatch logic
struct TraceLaunchRecord { char kernel[96]; int n_blocks, block_threads, lds_bytes; };
const TraceLaunchRecord& last_trace_launch();

// mode 0: whole CSMain per thread (RS:431-469)
hipError_t launch_mega(const DevScene& S, const FrameParams& P, float4* result, DevCounters* ctr, bool count, hipStream_t st);
// mode 1: generate + one launch per bounce over compacted path queues
hipError_t launch_wavefront(const DevScene& S, const FrameParams& P, const PathQueues& Q, float4* result, DevCounters* ctr,
                            bool count, hipStream_t st);
// mode 2: persistent waves with path regeneration (one launch per dispatch)
hipError_t launch_persist(const DevScene& S, const FrameParams& P, float4* result, DevCounters* ctr, unsigned int* next,
                          int n_blocks, bool count, hipStream_t st);
// mode 3: persistent waves whose lanes are scheduled by phase (FRONT / BLAS / SHADE) inside the wave
// One launch traces P.n_frames consecutive frames (uniforms: T[0 .. n_frames) in DEVICE memory, Result images P.frame_stride apart from `result`).
// front_mode: 0 = rays enter a triangle BVH through the BLAS phase only, 1 = they first walk its LDS-resident top inside FRONT,
// 2 = listed form of 1 (needs P.lds_mesh and n_meshes <= 12), 3 = masked form of 2 (mesh heap of <= 31 nodes: P.walk_f4 float4s of walk
// table behind S.mesh_tlas, P.lds_mesh = 0; kernels.hip front_masked)
hipError_t launch_sched(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                        unsigned int* next, int n_blocks, int front_mode, bool count, hipStream_t st);
// mode 5: mode 3 with the triangle-BVH phase as a service shared by the 4 waves of a workgroup (k_serve); `mail` = 2 float4 per
// thread of the grid (ray origin / direction of the posted rays); needs P.serve = 1 and P.block_threads = 256
hipError_t launch_serve(const DevScene& S, const FrameParams& P, const FrameUniforms* T, float4* result, DevCounters* ctr,
                        unsigned int* next, float4* mail, int n_blocks, int front_mode, bool count, hipStream_t st);
// fused AdditionShader blends of n consecutive frames into one image: dst = blend(... blend(blend(dst, src_0), src_1) ..., src_{n-1})
// in that order per pixel — the same operations as n launch_blit_add calls; src_f = src + f * frame_stride.  `present` (or null)
// also receives the final value: the copy Graphics.Blit(dst, present) would make after the last blend (RM:819)
hipError_t launch_blit_add_multi(const float4* src, size_t frame_stride, int n, const float* samples, float4* dst, float4* present,
                                 size_t n_pixels, hipStream_t st);
size_t sched_lds_bytes(const DevScene& S, const FrameParams& P);          // dynamic LDS of one workgroup (4 waves) of mode 3
// mode 4: persistent waves over a pool of 64*k paths per wave kept in LDS, phases run on compacted lanes
hipError_t launch_pool(const DevScene& S, const FrameParams& P, float4* result, DevCounters* ctr, unsigned int* next,
                       int n_blocks, int k, bool count, hipStream_t st);
size_t pool_lds_bytes(const FrameParams& P, int k);   // dynamic LDS of one wave (= one workgroup) of mode 4
// AdditionShader blend (AS:9,39-41)
hipError_t launch_blit_add(const float4* src, float4* dst, size_t n_pixels, float sample, hipStream_t st);
// strips <-> dense buffer
hipError_t launch_pack_rows(float4* img, float4* dense, int width, int height, int first_group_row, int row_stride,
                            int n_strips, bool to_dense, hipStream_t st);

// the same with three channels per pixel (12 B, RGB); unpacking writes `alpha` into the fourth
hipError_t launch_pack_rows_rgb(float4* img, float* dense, int width, int height, int first_group_row, int row_stride,
                                int n_strips, bool to_dense, float alpha, hipStream_t st);

}  // namespace urtd
