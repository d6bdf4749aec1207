"""Reduce gpurun_out/prof_<CONFIG> (scripts/collect_profiles.sh) to the small files kept under profiles/:
   <prefix>_kernel_stats.csv, <prefix>_pmc_k_sched.json, <prefix>_bench.json.log, and the config's entry of pmc_traffic.json.
   usage: python scripts/reduce_profiles.py CONFIG [round-prefix, default r03] [TAG]      (TAG as given to collect_profiles.sh)
   pmc_traffic.json is keyed by config AND frames per launch (configs[cfg]["by_frames_per_launch"]["20"]): bench.py looks the
   entry of its own launch shape up for `roofline.traffic` and the `issue` object."""
import csv, glob, json, os, shutil, sys
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
tag = sys.argv[3] if len(sys.argv) > 3 else ""
src = f"gpurun_out/prof_{cfg}{tag}"
prefix = f"{rnd}_bench_{cfg.lower()}{tag}"
command = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else f"python3 bench.py --config {cfg}"
dst = "profiles"
KERNEL = "k_sched<false"   # the timed trace kernel (the counting replay is k_sched<true ...)

def newest(pattern):
    """one file per pass directory: gpurun merges new results next to older ones"""
    by_dir = {}
    for f in glob.glob(pattern, recursive=True):
        d = f[len(src):].strip(os.sep).split(os.sep)[0]
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]): by_dir[d] = f
    return sorted(by_dir.values())

bench = {}
b = os.path.join(src, "bench.json.log")
if os.path.exists(b):
    shutil.copy(b, os.path.join(dst, f"{prefix}_n1.json.log"))
    try: bench = json.loads(open(b).read().strip().splitlines()[-1])
    except Exception: bench = {}
stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
kname, avg_ns, calls = None, None, None
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{prefix}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        if KERNEL in r["Name"]:
            kname, avg_ns, calls = r["Name"], float(r["AverageNs"]), int(r["Calls"])
            print("kernel stats:", r["Name"][:70], "calls", r["Calls"], "avg ns", r["AverageNs"])
    # duration of the timed launch itself (the last one of the kernel) from the kernel trace next to the stats
    last_ns = None
    for tf in glob.glob(stats[0].replace("kernel_stats.csv", "kernel_trace.csv")):      # the trace of the same run (same pid prefix)
        rows = [r for r in csv.DictReader(open(tf)) if KERNEL in r["Kernel_Name"]]
        if rows:
            r = max(rows, key=lambda r: int(r["Start_Timestamp"]))
            last_ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            print("timed launch (last of", len(rows), "):", last_ns, "ns")
            # every launch of the trace kernels in the run, in order (the stats file only has their average): clock warm-up launches,
            # the W warm-up frames, the timed launch(es), then the counting replay (k_sched<true ...>)
            allk = sorted((r for r in csv.DictReader(open(tf)) if "k_sched<" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
            with open(os.path.join(dst, f"{prefix}_kernel_launches.csv"), "w") as f:
                f.write("launch,kernel,duration_ns\n")
                for k, r in enumerate(allk):
                    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
                    f.write(f'{k},"{name}",{int(r["End_Timestamp"]) - int(r["Start_Timestamp"])}\n')
pmc = {}
for f in newest(os.path.join(src, "pmc_*", "**", "*counter_collection.csv")):
    acc = {}
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]: continue
        acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    # the TIMED launch = the last launch of the kernel in the run (the earlier one traces the warm-up frames); sums over the
    # counter's instances (XCDs / SEs) of that one dispatch
    for k, per in acc.items(): pmc[k] = per[max(per, key=int)]
json.dump(pmc, open(os.path.join(dst, f"{prefix}_pmc_k_sched.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(pmc, indent=1, sort_keys=True))
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    hbm = int(2 * pmc["FETCH_SIZE"] * 1024 + pmc["WRITE_SIZE"] * 1024)
    path = os.path.join(dst, "pmc_traffic.json")
    try: allcfg = json.load(open(path))
    except Exception: allcfg = {}
    if "configs" not in allcfg: allcfg = {"configs": {}}
    rl = bench.get("roofline") or {}
    fpl = rl.get("frames_per_launch")
    entry = {
        "kernel": kname, "round": rnd, "frames_per_launch": fpl, "command": command,
        "FETCH_SIZE_KB_per_launch": pmc["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": pmc["WRITE_SIZE"],
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md 'HBM'); WRITE_SIZE exact",
        "hbm_bytes_per_launch": hbm, "rocprof_timed_launch_ms": None if not stats or last_ns is None else last_ns / 1e6,
        "rocprof_avg_launch_ms": None if avg_ns is None else avg_ns / 1e6, "rocprof_launches": calls,
        "bench_launch_ms": rl.get("launch_ms"), "algorithmic_bytes_per_launch": (rl.get("algorithmic") or {}).get("bytes_per_launch", rl.get("algorithmic_bytes_per_launch")),
        "algorithmic_frac_of_hbm_peak": (rl.get("algorithmic") or {}).get("frac_of_hbm_peak", rl.get("frac")),
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- {command} --no-cpu-baseline; the LAST launch of the timed kernel in the run"}
    # what bounds the kernel when the byte fraction saturates: issue-side figures of the same launch (separate SQ_* passes)
    need = ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")
    if all(k in pmc for k in need):
        cycles = pmc["GRBM_GUI_ACTIVE"] / 8.0                     # per XCD: the launch's active cycles
        entry["issue"] = {
            # VALU wave-instructions issued, against the chip's best rate: one wave64 fma per 2 cycles per SIMD (1024 SIMDs).  The same chip
            # issues add/min/max mixes at one per 2.5-2.65 cycles and compare + select pairs at one per 4.1-4.9 (scripts/micro/valu_rate.hip,
            # profiles/r03_logs/r3_valu_rate_microbench.log): `simd_cycles_per_valu_inst` is what this kernel's mix gets
            "valu_frac": round(2.0 * pmc["SQ_INSTS_VALU"] / 1024.0 / cycles, 4),
            "simd_cycles_per_valu_inst": round(1024.0 * cycles / pmc["SQ_INSTS_VALU"], 3),
            "lane_util": round(pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_INSTS_VALU"] * 64.0), 4),
            "wait_frac": round(pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"], 4),
            "valu_insts_per_frame": None if not fpl else round(pmc["SQ_INSTS_VALU"] / fpl),
            "valu_insts_per_launch": round(pmc["SQ_INSTS_VALU"]),          # bench.py's roofline: x 128 flop / its live launch time / 157.3 TFLOP/s
            "source": f"profiles/{prefix}_pmc_k_sched.json: 2 x SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE/8 cycles), SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU x 64), SQ_WAIT_ANY / SQ_WAVE_CYCLES of the last timed launch"}
    c = allcfg["configs"].setdefault(cfg, {})
    if "by_frames_per_launch" not in c:                           # (round-2 layout: one flat entry per config)
        old = dict(c); c.clear(); c["by_frames_per_launch"] = {}
        if old: c["by_frames_per_launch"][str(int(round(old.get("frames_per_launch") or 1)))] = old
    c["by_frames_per_launch"][str(int(round(fpl or 1)))] = entry
    json.dump(allcfg, open(path, "w"), indent=1)
    print("hbm bytes per launch", hbm, entry.get("issue"))
