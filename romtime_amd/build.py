"""Build libromtime_hip.so (gfx950) in-tree with hipcc.

``python -m romtime_amd.build`` or ``romtime_amd.build.build()``.  hipcc cross-compiles
without a GPU; the .so lands in ``romtime_amd/lib/`` (git-ignored, shipped by gpurun).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libromtime_hip.so")
SOURCES = ["api.hip", "gemm_mfma.hip", "tallskinny.hip", "rank_update.hip", "gram_mfma.hip", "deim.hip", "sparse.hip", "project_fused.hip", "solve.hip", "sweep.hip", "symeig.hip", "host_dense.cpp", "p1_assembly.hip", "pod_orth.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# measurement builds only, e.g. ROMTIME_EXTRA_HIPFLAGS=-DROMTIME_PF_ABLATE (timing switches of csrc/project_fused.hip)
FLAGS += os.environ.get("ROMTIME_EXTRA_HIPFLAGS", "").split()


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_panel.h"), os.path.join(CSRC, "wave_ops.h"), os.path.join(CSRC, "host_dense.h"), os.path.join(HERE, "..", "include", "romtime_hip.h")]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
