#!/usr/bin/env python3
"""One parametrised knob sweep / A-B tool (run on the GPU box from the repo root).

    python scripts/sweep.py [--configs C3 C3@4K C4] [--set kernel_mode=3 ...] [--grid waves_per_cu=10,12,16 refill_min=8,32 ...]
                            [--frames 8] [--warmup 3] [--reps 2] [--bounces B] [--libs default build_a.so build_b.so]

Every point of the cartesian product of --grid (on top of the fixed --set options) renders `frames` frames of each
configuration through RayTraceMaster.OnRenderImage and prints
    kernel ms  = sum of the trace kernels' own HIP-event durations / frames   (urt_get_counters().trace_ms)
    wall ms    = host clock around the frame loop incl. the final synchronize / frames  (what bench.py reports;
                 with overlapped frames this is smaller than kernel ms)
Options a build does not know are reported and the point is skipped.  --libs runs every point in a child process per library
build (URT_LIB_PATH), interleaved `reps` times, for A/B of BUILDS on one box.  `name=a,b` values are integers
(urt_set_option).  This replaces the one-off sweep scripts of round 1 (sweep4..9, top_sweep, band_sweep, leaf_sweep,
pool_sweep, onoff, order_test, scale_test, top_occ_test).
"""
from __future__ import annotations

import argparse
import itertools
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_kv(items, many):
    out = []
    for it in items or []:
        k, v = it.split("=", 1)
        out.append((k, [int(x) for x in v.split(",")] if many else int(v)))
    return out


def make_scene(name, bounces):
    from unityraytracer_amd import scenes
    if "@" in name:
        base, res = name.split("@")
        w, h = {"4K": (3840, 2160), "1080": (1920, 1080), "540": (960, 540)}[res]
        sc = scenes.CONFIGS[base](w, h)
    else:
        sc = scenes.CONFIGS[name]()
    if bounces:
        sc.num_bounces = bounces
    return sc


def run_points(args):
    from unityraytracer_amd import Context, RayTraceMaster
    fixed = parse_kv(args.set, many=False)
    grid = parse_kv(args.grid, many=True)
    ctx = Context(0)
    for rep in range(args.reps):
        for name in args.configs:
            sc = make_scene(name, args.bounces)
            for combo in itertools.product(*[vals for _, vals in grid]) if grid else [()]:
                opts = dict(fixed)
                opts.update({k: v for (k, _), v in zip(grid, combo)})
                label = " ".join(f"{k}={v}" for k, v in opts.items()) or "(defaults)"
                try:
                    for k, v in opts.items():
                        ctx.set_option(k, v)
                except Exception as e:                       # an option this build does not have
                    print(f"{name:7s} {label}: skipped ({e})", flush=True)
                    continue
                ctx.set_option("count_stats", 0)
                ctx.set_option("time_dispatch", 1)
                m = RayTraceMaster(ctx, sc)
                for _ in range(args.warmup):
                    m.OnRenderImage()
                ctx.synchronize()
                ctx.reset_counters()
                t0 = time.perf_counter()
                for _ in range(args.frames):
                    m.OnRenderImage()
                ctx.synchronize()
                wall = (time.perf_counter() - t0) * 1e3 / args.frames
                c = ctx.counters()
                m.OnDisable()
                print(f"{name:7s} {label}: kernel {c['trace_ms'] / args.frames:8.3f} ms  wall {wall:8.3f} ms  "
                      f"{c['rays'] / args.frames / wall / 1e3:8.1f} Mrays/s  wd {c['watchdog_trips']}", flush=True)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="*", default=["C3"])
    ap.add_argument("--set", nargs="*", default=[])
    ap.add_argument("--grid", nargs="*", default=[])
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--bounces", type=int, default=0)
    ap.add_argument("--libs", nargs="*", default=None, help="library builds to A/B ('default' = the in-tree one)")
    args = ap.parse_args()
    if not args.libs:
        run_points(args)
        return
    child = [sys.executable, os.path.abspath(__file__), "--configs", *args.configs, "--frames", str(args.frames), "--warmup", str(args.warmup),
             "--bounces", str(args.bounces)]
    if args.set:
        child += ["--set", *args.set]
    if args.grid:
        child += ["--grid", *args.grid]
    for rep in range(args.reps):
        for lib in args.libs:
            env = dict(os.environ)
            if lib != "default":
                env["URT_LIB_PATH"] = os.path.abspath(lib)
                env["URT_ALLOW_EXPERIMENT"] = "1"              # A/B builds report a negative ABI version and are refused otherwise (csrc/experiments.h)
            out = subprocess.run(child, env=env, capture_output=True, text=True, cwd=ROOT)
            for line in out.stdout.splitlines():
                print(f"{os.path.basename(lib):24s} {line}", flush=True)
            if out.returncode:
                print(out.stderr[-600:], flush=True)


if __name__ == "__main__":
    main()
