#!/usr/bin/env python3
"""Build the reference's own CPU path, UNMODIFIED, as a checker binary.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.

Compiles, from where they lie under /root/reference (never copied into this
repo), the four host translation units of the reference's native extension

    pytorch_binding/src/pychain.cc                       (pybind boundary, :26-135)
    pytorch_binding/src/base.cc                          (verbose level, ApproxEqual)
    pytorch_binding/src/chain-computation.cc             (denominator, CPU branch :136-175, :272-310)
    pytorch_binding/src/chain-log-domain-computation.cc  (numerator,   CPU branch :123-159, :231-271)

into oracle/_ref/pychain_C.so (git-ignored, but it travels to the GPU box).

Notes on what is and is not done here:
  * chain-kernels-ansi.h:17-18 includes <cuda.h>/<cuda_runtime.h>.  This image
    carries NVIDIA's real headers inside the triton wheel
    (triton/backends/nvidia/include); they are used as-is.  No stand-in header
    is written.
  * The two .cu files cannot be compiled (no nvcc) and are not needed: the four
    cuda_chain_hmm_* symbols are only referenced from the `if (cuda_)` branches.
    They stay UNDEFINED in the .so; loaders must dlopen it with RTLD_LAZY
    (oracle/ref_loader.py does) so they are never resolved on the CPU path.
  * -DNDEBUG is mandatory: it is what the reference gets from Python's
    sysconfig CFLAGS under `setup.py install`, and without it
    chain-log-domain-computation.cc:144 asserts on the first -inf alpha.
"""
import os
import subprocess
import sys
import sysconfig

REF_SRC = "/root/reference/pytorch_binding/src"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "pychain_C.so")
SOURCES = ["pychain.cc", "base.cc", "chain-computation.cc", "chain-log-domain-computation.cc"]


def cuda_header_dir():
    import triton  # only for its bundled, genuine CUDA headers
    d = os.path.join(os.path.dirname(triton.__file__), "backends", "nvidia", "include")
    if not os.path.exists(os.path.join(d, "cuda_runtime.h")):
        raise RuntimeError("no CUDA headers on this image: reference is unbuildable here")
    return d


def build(force=False):
    if not os.path.isdir(REF_SRC):
        raise RuntimeError("reference sources are not present (expected on the GPU box)")
    srcs = [os.path.join(REF_SRC, s) for s in SOURCES]
    if (not force and os.path.exists(OUT)
            and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs)):
        return OUT
    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    inc = cpp_extension.include_paths() + [sysconfig.get_paths()["include"], cuda_header_dir(), REF_SRC]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    objs = []
    for s in srcs:
        o = os.path.join(OUT_DIR, os.path.basename(s) + ".o")
        cmd = (["g++", "-c", s, "-o", o, "-O2", "-DNDEBUG", "-fPIC", "-std=c++17", "-w",
                "-DTORCH_EXTENSION_NAME=pychain_C", "-DTORCH_API_INCLUDE_EXTENSION_H",
                "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
               + ["-I" + i for i in inc])
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = (["g++", "-shared", "-o", OUT] + objs
           + ["-L" + torch_lib, "-Wl,-rpath," + torch_lib,
              "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10"])
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
