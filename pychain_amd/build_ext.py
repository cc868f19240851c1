"""In-tree build of libpychain_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m pychain_amd.build_ext [--force]

Every translation unit is compiled to an object of its own (in parallel, only the stale ones: an object
depends on its source and on every header), then linked.  Objects live under build/obj (git-ignored).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB = os.path.join(_HERE, "libpychain_hip.so")
OBJ = os.path.join(os.path.dirname(_HERE), "build", "obj")
SOURCES = ["den_rec.hip", "den_lazy.hip", "den_kernels.hip", "plan.cpp", "fst.cpp", "pack.cpp", "cpu.cpp", "den_general.hip", "num_kernels.hip", "num_general.hip", "num_compat.hip",
           "api.hip"]          # (the slowest translation units first: they are compiled side by side)
HEADERS = ["common.h", "plan_format.h", "den_kernels.h", "num_kernels.h", "device_utils.h", "den_common.inc.h", "den_lazy.inc.h",
           "den_pair.inc.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-fno-slp-vectorize"]   # v_pk_*_f32 pairs cost v_movs and lengthen the dependent chains here


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _header_time():
    deps = [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(INCLUDE, "pychain_hip.h"), os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps if os.path.exists(d))


def _stale(src, obj, htime):
    return not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), htime)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in _sources()) or _header_time() > t


def build(force=False, verbose=False, extra_flags=(), lib=None):
    """`extra_flags` / `lib`: ablation and timing builds (-DPYCHAIN_EXP_*, -DPYCHAIN_PROFILE_PHASES) into a library of
    their own (objects are then not cached)."""
    out = lib or LIB
    variant = bool(extra_flags) or lib is not None
    if not force and not variant and not needs_build():
        return out
    # (a variant's objects live under a name derived from its flags and output - stable from process to process, so a
    # rebuilt ablation reuses its directory instead of leaving a new one behind every time)
    objdir = OBJ if not variant else OBJ + "_" + hashlib.sha1(repr((tuple(extra_flags), os.path.abspath(out))).encode()).hexdigest()[:10]
    os.makedirs(objdir, exist_ok=True)
    htime = _header_time()
    jobs = []
    for s in _sources():
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s + ".o")
        if force or variant or _stale(src, obj, htime):
            jobs.append([_hipcc()] + FLAGS + list(extra_flags) + ["-I", INCLUDE, "-c", src, "-o", obj])
    if verbose:
        for j in jobs:
            print(" ".join(j))

    def run(cmd):
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + \
        [os.path.join(objdir, s + ".o") for s in _sources()] + ["-o", out]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
