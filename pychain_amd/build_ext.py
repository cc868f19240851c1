"""In-tree build of libpychain_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m pychain_amd.build_ext [--force]
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB = os.path.join(_HERE, "libpychain_hip.so")
SOURCES = ["plan.cpp", "fst.cpp", "den_kernels.hip", "num_kernels.hip", "api.hip"]
HEADERS = ["common.h", "plan_format.h", "den_kernels.h", "num_kernels.h", "device_utils.h"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(INCLUDE, "pychain_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-fno-slp-vectorize",   # v_pk_*_f32 pairs cost v_movs and lengthen the dependent chains here
           "-I", INCLUDE] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
