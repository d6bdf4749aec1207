# like ab.py, but every build is also run under several option sets: python scripts/ab_opts.py libs... -- CFG...
import os, subprocess, sys
args = sys.argv[1:]
cfgs = ["C3"]
if "--" in args:
    i = args.index("--"); cfgs = args[i + 1:]; args = args[:i]
code = r'''
import sys
sys.path.insert(0, '.')
from unityraytracer_amd import Context, RayTraceMaster, scenes
ctx = Context(0)
def run(sc, opts, frames=8):
    for k, v in opts.items():
        try: ctx.set_option(k, v)
        except Exception: return None
    ctx.set_option("count_stats", 0); ctx.set_option("time_dispatch", 1)
    m = RayTraceMaster(ctx, sc)
    for _ in range(3): m.OnRenderImage()
    ctx.synchronize(); ctx.reset_counters()
    for _ in range(frames): m.OnRenderImage()
    c = ctx.counters(); m.OnDisable()
    return c['trace_ms'] / frames
for name in sys.argv[1:]:
    sc = scenes.config3(3840, 2160) if name == "C3@4K" else scenes.CONFIGS[name]()
    for label, o in (("b64 top0", {"sched_block": 64, "top_nodes": 0}), ("b64 top16", {"sched_block": 64, "top_nodes": 16}), ("b256 top32", {"sched_block": 256, "top_nodes": 32}), ("b256 top64", {"sched_block": 256, "top_nodes": 64})):
        ms = run(sc, o, frames=4 if name in ("C4", "C5") else 8)
        if ms is not None: print(f"{name:6s} {label:10s} {ms:8.3f} ms", flush=True)
        elif label == "b64 top0": print(f"{name:6s} {'(no opts)':10s} {run(sc, {}, frames=4 if name in ('C4', 'C5') else 8):8.3f} ms", flush=True)
'''
for rep in range(2):
    for lib in args:
        env = dict(os.environ)
        if lib != "default": env["URT_LIB_PATH"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", code] + cfgs, env=env, capture_output=True, text=True)
        for line in out.stdout.splitlines(): print(f"{os.path.basename(lib):20s} {line}", flush=True)
        if out.returncode: print(out.stderr[-400:])
