mkdir -p gpurun_out/r3; rm -f gpurun_out/r3/ab_order_bench.log
for rep in 1 2 3; do for o in 0 1; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --opt tile_order=$o | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s20 order $o', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])" >> gpurun_out/r3/ab_order_bench.log || exit 1
  python bench.py --no-cpu-baseline --opt tile_order=$o | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default order $o', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])" >> gpurun_out/r3/ab_order_bench.log || exit 1
done; done
