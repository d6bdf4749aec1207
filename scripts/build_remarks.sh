#!/bin/bash
# Compile-only build of the product with -Rpass-analysis=kernel-resource-usage; prints one line per kernel
# (VGPRs / AGPRs / SGPRs / scratch / occupancy / LDS).  Usage: scripts/build_remarks.sh [out.so] [extra hipcc flags...]
cd "$(dirname "$0")/../unityraytracer_amd" || exit 1
OUT=${1:-/tmp/urt_remarks.so}; shift
# KERNELS_ONLY=1: compile csrc/kernels.hip alone (-c): the trace kernels' numbers in half a minute
if [ -n "$KERNELS_ONLY" ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -pthread -Xarch_host -march=x86-64-v3 -Xarch_device -fno-slp-vectorize \
    -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" -c csrc/kernels.hip -o "$OUT" 2> /tmp/urt_remarks.log
else
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden -pthread -Xarch_host -march=x86-64-v3 -Xarch_device -fno-slp-vectorize \
  -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" -o "$OUT" \
  csrc/kernels.hip csrc/lbvh.hip csrc/refit.hip csrc/qnodes.hip csrc/cullflags.hip csrc/present.hip csrc/context.cpp csrc/blas_builder.cpp csrc/host_scene.cpp csrc/host_io.cpp csrc/host_debug.cpp csrc/group.cpp 2> /tmp/urt_remarks.log
fi
rc=$?
grep -E "error" -A6 /tmp/urt_remarks.log | head -40
python3 - <<'PY'
import re, subprocess
rows, cur = [], None
for line in open('/tmp/urt_remarks.log'):
    m = re.search(r'remark: (?:\s*)(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (.*?) \[-Rpass', line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name':
        cur = {'name': v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    if 'rocprim' in name: continue                       # library sort / scan kernels of the GPU BVH build
    name = re.sub(r'^(void )?\(anonymous namespace\)::', '', name).split('(')[0]
    print(f"{name:45s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('TotalSGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} B/lane  "
          f"spill v{r.get('VGPRs Spill','?')} s{r.get('SGPRs Spill','?')}  occupancy {r.get('Occupancy [waves/SIMD]','?')}  LDS {r.get('LDS Size [bytes/block]','?')}")
PY
exit $rc
