// Compile-only reproducer for DESIGN.md §8, first bullet (round 1: a divergent select between two kernel-argument pointers in
// the long megakernel read the wrong material table and faulted on a scene without spheres, gpurun_out/diag2.log).
// This is the construct in isolation: `kind` is per-lane, both tables are by-value kernel arguments.
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only profiles/isa/kernarg_select_repro.hip -o -
#include <hip/hip_runtime.h>
struct Scene { const float4* sphere_mat; int n_spheres; const float4* mesh_mat; int n_meshes; const int* kind; const int* id; };
__global__ void k_select(Scene S, float4* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int kind = S.kind[i], id = S.id[i];
  const float4* m = kind == 2 ? S.sphere_mat + 3 * (size_t)id : S.mesh_mat + 3 * (size_t)id;   // the removed construct
  float4 a = m[0], b = m[1], c = m[2];
  out[i] = make_float4(a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z, a.w);
}
