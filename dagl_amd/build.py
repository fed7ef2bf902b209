"""Build the HIP device library in-tree: dagl_amd/csrc/*.hip -> dagl_amd/csrc/libdagl_ce.so (gfx950 only)."""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdagl_ce.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("DAGL_EXTRA_FLAGS", "").split()       # e.g. -DDAGL_ABLATION for the debug variants (tools/ablate.sh)


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(os.path.dirname(CSRC), "..", "include", "dagl_ce.h")]
    if not force and not _stale(LIB, srcs + hdrs) and not os.environ.get("DAGL_EXTRA_FLAGS"):
        return LIB                # the library is newer than every source: nothing to do (the objects do not travel to the GPU box)
    objs = []
    jobs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with cf.ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
