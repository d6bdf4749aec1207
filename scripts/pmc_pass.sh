#!/bin/bash
# One or more separate rocprofv3 PMC passes of a bench command (run on the GPU box from the repo root):
#   scripts/pmc_pass.sh OUTNAME "bench args" "COUNTERS OF PASS 1" ["COUNTERS OF PASS 2" ...]
# Every pass is its own rocprofv3 run (never combined with a trace domain), with its own timeout; the program after `--` is python3.
# Output: gpurun_out/pmc_<OUTNAME>/<pass>/ (scratch) and a per-kernel average table gpurun_out/pmc_<OUTNAME>.txt
set -u
NAME=$1; ARGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
for set in "$@"; do
  name=$(echo "$set" | tr ' ' '+')
  timeout -k 5 240 rocprofv3 --pmc $set --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS --no-cpu-baseline > "$OUT/$name.log" 2>&1 || echo "pass $name failed (see $OUT/$name.log)"
  echo "pass $name done" >> "$OUT/progress.txt"
done
python3 - "$OUT" > "$OUT.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rocprim" in k: continue
        k = k.replace("void (anonymous namespace)::", "")
        acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} launches {len(v):4d}  mean {sum(v) / len(v):16.1f}  sum {sum(v):18.1f}")
PY
cat "$OUT.txt"
