#!/bin/bash
# Run on the GPU box from the repo root: rocprofv3 kernel trace + separate PMC passes of the bench command.
#   scripts/collect_profiles.sh CONFIG [full|lite] [TAG [bench args...]]      (CONFIG = C2 | C3 | C3D | C4 | C5)
# full = kernel trace + ten PMC passes (the headline config); lite = kernel trace + FETCH_SIZE + WRITE_SIZE passes.
# TAG + bench args profile another command of the same config, e.g. `C3 full _s20 --gpus 1 --steps 20 --warmup 5` = the driver's.
# Output: gpurun_out/prof_<CONFIG><TAG>/<pass>/... (scratch); reduce with scripts/reduce_profiles.py and copy the result to profiles/.
# PMC passes are never combined with a trace domain; every pass has its own timeout; the program after `--` is python3 itself.
set -u
CFG=${1:-C3}; MODE=${2:-full}; TAG=${3:-}
if [ $# -ge 3 ]; then shift 3; else shift $#; fi
EXTRA="$*"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_$CFG$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="bench.py --config $CFG $EXTRA"   # no extra args = the default command: 8 warm-up frames, 256 timed frames (four launches of 64)
echo "python3 $BENCH" > "$OUT/command.txt"
timeout -k 10 300 python3 $BENCH > "$OUT/bench.json.log" 2> "$OUT/bench.err" || exit 1
echo "bench done" >> "$OUT/progress.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $BENCH --no-cpu-baseline > "$OUT/trace.log" 2>&1 || exit 1
echo "trace done" >> "$OUT/progress.txt"
if [ "$MODE" = full ]; then
  SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE")
else
  SETS=("FETCH_SIZE" "WRITE_SIZE")
fi
for set in "${SETS[@]}"; do
  name=$(echo "$set" | tr ' ' '+')
  timeout -k 5 240 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$name" -- python3 $BENCH --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1 || { echo "pass $name failed" >> "$OUT/failed.txt"; }
  echo "pass $name done" >> "$OUT/progress.txt"
done
