"""Load oracle/_ref/pychain_C.so (the reference's own CPU path; see build_ref.py).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, bench.py's cpu_baseline
leg, __graft_entry__.smoke().  The product package never imports this.

The .so keeps the four cuda_chain_hmm_* symbols undefined (their .cu files need
nvcc); RTLD_LAZY leaves them unresolved, which is fine on the CPU path.
"""
import importlib.machinery
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "pychain_C.so")
_mod = None


def available():
    return os.path.exists(REF_SO)


def load():
    """Return the reference `pychain_C` module (forward_backward,
    forward_backward_log_domain, set_verbose_level: pychain.cc:131-135)."""
    global _mod
    if _mod is not None:
        return _mod
    if not available():
        raise FileNotFoundError(REF_SO + " missing: run oracle/build_ref.py where /root/reference exists")
    import torch  # noqa: F401  (libtorch must be loaded first)
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        loader = importlib.machinery.ExtensionFileLoader("pychain_C", REF_SO)
        spec = importlib.util.spec_from_loader("pychain_C", loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    _mod = mod
    return mod
