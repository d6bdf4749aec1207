"""Build recipe of the product (in-tree, explicit hipcc — no JIT cache): the HIP library behind include/urt.h."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libunityraytracer_amd.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17",
    "-ffp-contract=off",            # normative arithmetic: no implicit fma (include/urt_math.h)
    "-fPIC", "-shared", "-fvisibility=hidden", "-pthread",
    "-Xarch_host", "-march=x86-64-v3",   # inline hardware fma for the host-side vertex pre-transform
    "-Xarch_device", "-fno-slp-vectorize",   # v_pk_fma_f32 issues at half rate on gfx950 and its operand pairs cost moves: -0.5 .. -1.4 % frame time (r3_ab_cnodes_noslp.log)
    "-Wall", "-Wno-unused-function",
]
SOURCES = ["kernels.hip", "lbvh.hip", "refit.hip", "qnodes.hip", "cullflags.hip", "present.hip", "context.cpp", "blas_builder.cpp", "host_scene.cpp", "host_io.cpp", "host_debug.cpp", "group.cpp"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_variant(out: str, defines, verbose: bool = False) -> str:
    """An A/B or diagnostic build of the same sources with extra -D flags (e.g. URT_STAMPS, URT_SCHED_OCC=4) -> `out`
    (objects under unityraytracer_amd/build/<name of out>/).  Used through URT_LIB_PATH by scripts/sweep.py --libs and scripts/stamps*.py."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(_HERE, "build", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    # every variant is an EXPERIMENT build: negative urt_abi_version, refused by _lib.load() unless URT_ALLOW_EXPERIMENT=1 (csrc/experiments.h)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-DURT_EXPERIMENT"] + [d for d in defines if d != "-DURT_EXPERIMENT"]
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    base = os.path.join(_HERE, "build")
    build_library()                                          # the objects shared with the product must be current (they are reused below)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=_HERE)

    objs, jobs = [], []
    for src in srcs:
        name = os.path.basename(src)
        if name in ("kernels.hip", "context.cpp", "blas_builder.cpp") or not os.path.exists(os.path.join(base, name + ".o")):   # the rest does not depend on the defines
            obj = os.path.join(objdir, name + ".o")
            jobs.append([hipcc] + cflags + ["-c", src, "-o", obj])
        else:
            obj = os.path.join(base, name + ".o")
        objs.append(obj)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-pthread", "-o", out] + objs)
    return out


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 ... -> unityraytracer_amd/libunityraytracer_amd.so (cross-compiles without a GPU).
    One object per source under unityraytracer_amd/build/ (compiled in parallel, reused while newer than its source and every
    header), then one link: an edit of context.cpp does not recompile the kernels."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + \
        [os.path.join(_ROOT, "include", h) for h in os.listdir(os.path.join(_ROOT, "include"))]
    if not force and _newer(LIB, srcs + hdrs + [os.path.abspath(__file__)]):     # (the flags live in this file)
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or not _newer(obj, [src] + hdrs + [os.path.abspath(__file__)]):
            jobs.append([hipcc] + cflags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=_HERE)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, os.path.basename(src) + ".o") for src in srcs]
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-pthread", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2:                                   # python -m unityraytracer_amd.build out.so -DURT_STAMPS ...
        print(build_variant(os.path.abspath(sys.argv[1]), sys.argv[2:], verbose=True))
    else:
        print(build_library(force=len(sys.argv) > 1 and sys.argv[1] == "--force", verbose=True))
