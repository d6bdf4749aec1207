#!/bin/bash
# A/B of library options with bench.py's own metric on the driver's command: scripts/ab_opts_bench.sh "xcd_run=1" "xcd_run=2 frame_group=10" ...
# every argument is one variant (space-separated NAME=VALUE pairs; "" = defaults); REPS (default 2) passes, alternating.  Output: gpurun_out/ab_opts_bench.log
mkdir -p gpurun_out; : > gpurun_out/ab_opts_bench.log
CMD=${BENCH_CMD:---gpus 1 --steps 20 --warmup 5}
for rep in $(seq ${REPS:-2}); do for v in "$@"; do
  opts=""; for kv in $v; do opts="$opts --opt $kv"; done
  python bench.py $CMD --no-cpu-baseline $opts | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])" >> gpurun_out/ab_opts_bench.log || exit 1
done; done
sort gpurun_out/ab_opts_bench.log
