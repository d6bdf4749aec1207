// urt_device.h — device-side scene/frame descriptors shared by the host library (context.cpp) and the
// HIP kernels (kernels.hip).  Layouts are the library's own (SoA / 16-byte records for coalesced
// dwordx4 loads); they are derived on upload from the reference layouts of urt_types.h.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "blas_builder.h"   // kBlasNodeFloats, kEmptyMeshRoot

namespace urtd {

// Triangle BVH ("BLAS") node: 64 bytes = 4 x float4
//   q0 = c0.min.xyz, c0.max.x     q1 = c0.max.yz, c1.min.xy
//   q2 = c1.min.z, c1.max.xyz     q3 = child0, child1 (int bits), 0, 0
// child >= 0: interior node index.  child < 0: leaf, code = ~child, first leaf-order triangle slot =
// code >> 3, count = (code & 7) + 1.  Host-side/introspection layout is the flat 16-float form
// [c0.min c0.max c1.min c1.max child0 child1 0 0]; it is the same bytes.
struct DevScene {
  // object-level BVHs in the reference's implicit-heap order (RS:57-61,70-71), repacked to 32 B:
  //   [2i] = vmin.xyz, index (int bits)    [2i+1] = vmax.xyz, 0
  const float4* mesh_tlas;   int n_mesh_tlas;
  const float4* sphere_tlas; int n_sphere_tlas;
  // spheres (RS:51-55): position.xyz + radius; materials as 3 x float4
  const float4* sphere_pr;   int n_spheres;
  // materials: spheres [0, n_spheres), then the mesh objects, then the ground plane (RS:164-170); 4 x float4 each, holding what
  // Shade derives from the material alone, precomputed on the host (context.cpp pack_material):
  //   [4i] (1/diffChance) * albedo', specChance   [4i+1] (1/specChance) * specular, specChance + diffChance
  //   [4i+2] emission, diffChance                  [4i+3] alpha, 1/(alpha+1), (alpha+2)/(alpha+1), 0
  const float4* materials;
  // mesh objects (RS:43-49)
  const int32_t* mesh_root;  int n_meshes;
  // MeshObjects that are a single BVH leaf (<= 8 triangles: quads, planes): position of their triangle records in the LDS copy
  // the phase scheduler keeps of them (in triangles), or -1; n_small = triangles in that copy (0 = none)
  const int32_t* mesh_small_first; int n_small;
  // triangle BVH over world-space triangles, all meshes in one pool
  const float4* blas_nodes;  // 4 x float4 per node: child boxes as [lo, hi] — what the builders and the refit write
  const float4* blas_cnodes; // the same nodes with child boxes as (centre, half extent): what the trace kernels read (csrc/qnodes.hip k_center)
  const float4* blas_qnodes; // or null: [0] grid origin.xyz, quality  [1] cell.xyz, 0  [2 + 2n ..] the 32-byte quantized form of node n (csrc/qnodes.hip)
  const float4* tri_verts;   // [3k] v0.xyz, index slot i (int bits)  [3k+1] e1.xyz, mesh id (int bits)  [3k+2] e2.xyz, 0
  const float4* tri_norms;   // [3k..3k+2] n0, n1, n2 (object space, RS:259-261)
  // sky (RS:9-10): RGBA32F, row 0 = bottom, bilinear + repeat
  const float4* sky;         int sky_w, sky_h;
  // object-level cull (urt_math.h tlas_cull): 1 = some leaf of the mesh heap carries a cull word (its box was verified to contain its
  // MeshObject, csrc/cullflags.hip); 0 = the walk skips the cull arithmetic altogether
  int cull_any;
};

struct FrameParams {
  float c2w[16];            // _CameraToWorld            RS:5
  float invp[16];           // _CameraInverseProjection  RS:6
  float pixel_off_x, pixel_off_y;   // _PixelOffset      RS:7
  float seed;               // _Seed                     RS:16
  int num_bounces;          // _numBounces               RS:18
  int num_rays;             // _numRays                  RS:19
  int width, height;        // Result.GetDimensions      RS:438
  int region_w, region_h;   // min(8*groupsX, W), min(8*groupsY, H): pixels the dispatch covers
  int tiles_x;              // ceil(region_w / 8)
  int first_group_row;      // multi-GPU strips: global group row of local strip 0
  int row_stride;           //                   and the stride between this rank's strips
  int n_strips;             // local strips (8 pixel rows each)
  int tlas_stack;           // LDS entries per lane reserved for the object-level stacks (the listed FRONT keeps its object list in the first two)
  int blas_stack;           // LDS entries per lane reserved for the triangle-BVH stack
  int block_threads;        // workgroup size (64, 128 or 256)
  int tile_order;           // persistent modes: 0 = strips bottom to top (natural), 1 = top to bottom
  int xcd_run;              // blocks per XCD run in the tile order (kernels.hip tile_pixel); persistent modes: tiles per run of a work-counter shard
  int n_shards;             // persistent modes: work-counter shards in use (power of two, 1..kWorkShards)
  int frame_group;          // batched launches: frames whose tile runs are interleaved (1 = frame after frame)
  int refill_min;           // persistent modes: dead lanes per wave that trigger a refill (1..64)
  int blas_min;             // mode 3: lanes parked in BLAS before the traversal phase is scheduled (1..64)
  int shade_min;            // mode 3: SHADE (surface-hit) lanes that make the phase run ahead of FRONT (1..64)
  int sky_min;              // mode 3: SKY (miss) lanes that make the sky-lookup phase run ahead of FRONT (1..64)
  int shade_split;          // mode 3: surface hits and misses are scheduled as separate phases (0/1)
  int blas_exit;            // mode 3: the traversal phase yields when fewer lanes than this are still traversing (1..64)
  int top_nodes;            // mode 3: triangle-BVH nodes [0, top_nodes) are copied to LDS (breadth-first top of the forest); 0 = none
  int lds_small;            // mode 3: triangle records of the single-leaf MeshObjects in LDS (needs lds_mesh) (0/1)
  int lds_mesh, lds_sphere; // mode 3: keep the object-level mesh heap + roots / sphere heap + spheres in LDS (0/1)
  int walk_f4;              // mode 3, masked FRONT (front mode 3): float4s of the walk table behind the mesh heap's device copy, kept in LDS
                            // instead of the heap itself (context.cpp build_walk_table); 0 = not in use
  int serve;                // mode 5: the traversal phase is a service shared by the waves of a workgroup (kernels.hip k_serve) (0/1)
  int pool_inloop;          // modes 4, 5: idle lanes that trigger a re-feed of the traversal phase from the waiting rays (1..64)
  int pool_other_min;       // mode 4: lanes of FRONT / SHADE work that make those phases worth a trip while rays queue for the BVH
  unsigned int watchdog_steps;  // cap on traversal trips per scheduled BLAS phase: a few times (nodes + leaves) of the scene
  unsigned int sched_trips;     // persistent modes: cap on scheduler trips per wave, scaled with the launch (frames x rays x bounces; context.cpp)
  unsigned int* trip_flag;      // host-mapped word: a wave that leaves through a cap adds 1 (the next synchronising call reports URT_ERR_WATCHDOG)
  // mode 3, frame batching: ONE launch traces n_frames consecutive frames of the same scene/resolution (the library defers
  // dispatches, context.cpp).  Frame f's uniforms are entry f of the launch's table in device memory; its Result image starts
  // at result + f * frame_stride.  The c2w/invp/pixel_off/seed above are those of frame 0 (all the other kernel modes read).
  int n_frames;                 // >= 1
  unsigned int frame_stride;    // float4 elements between the Result images of consecutive frames of the launch
};

// per-frame uniforms of a batched launch (RM:773-778): 144 bytes each, read with scalar loads through a wave-uniform pointer
// (the frame index is made wave-uniform first)
struct FrameUniforms {
  float c2w[16];            // _CameraToWorld
  float invp[16];           // _CameraInverseProjection
  float pixel_off_x, pixel_off_y;   // _PixelOffset
  float seed;               // _Seed
  float pad;
};
static constexpr int kMaxFramesPerLaunch = 64;      // the table lives in device memory (context.cpp stages it through pinned host slots): 64 x 144 B
struct FrameTable { FrameUniforms f[kMaxFramesPerLaunch]; };   // host-side image of one launch's table

static constexpr unsigned int kWorkShards = 64;   // work counters of the persistent kernels (power of two), 128 B apart
static constexpr int kCounterShards = 256;   // power of two; a block adds to shard blockIdx & (N-1)
struct alignas(128) DevCounters {            // one shard = one 128-byte line of 64-bit counters
  unsigned long long rays, tlas_nodes, blas_nodes, tri_tests, sphere_tests;
  unsigned long long hit_tri, hit_sphere, hit_ground, hit_sky;
  unsigned long long watchdog;   // waves that left a persistent kernel through its iteration cap (must stay 0)
  // mode 5 with count_stats: visits of the traversal service, its trips, active lanes summed over the trips, claim rounds,
  // rays claimed, rays suspended (urt_debug_serve_stats)
  unsigned long long serve[6];
};
static_assert(sizeof(DevCounters) == 128, "one counter shard = one 128-byte line");

// path state of the wavefront pipeline: 4 x float4 per path, SoA by record
//   s0 = origin.xyz, seed   s1 = direction.xyz, pixel (int bits: y << 16 | x)
//   s2 = energy.xyz, 0      s3 = result.xyz, 0
struct PathQueues {
  float4* s[2][4];          // ping-pong
  unsigned int* counts;     // [num_rays * (num_bounces + 1)] live-path counts per (ray, bounce)
};

}  // namespace urtd
