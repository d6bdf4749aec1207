# A/B of library builds on the same box: python scripts/ab.py libA.so libB.so ... [-- CFG ...]; each build runs in its own process, interleaved 3x
import os, subprocess, sys
args = sys.argv[1:]
cfgs = ["C3"]
if "--" in args:
    i = args.index("--"); cfgs = args[i + 1:]; args = args[:i]
for rep in range(3):
    for lib in args:
        env = dict(os.environ)
        if lib != "default": env["URT_LIB_PATH"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "scripts/configs_run.py"] + cfgs, env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if "trace" in line: print(f"{os.path.basename(lib):24s} {line.split(':')[0][:28]:28s} {line.split('trace')[1].split(',')[0].strip()}", flush=True)
