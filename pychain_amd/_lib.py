"""ctypes binding of libpychain_hip.so (include/pychain_hip.h).

There is NO fallback: if the shared library is missing, or a call is made with
tensors that are not on a HIP device, this raises.  The product path never
touches the CPU checker.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYCHAIN_HIP_LIB") or os.path.join(_HERE, "libpychain_hip.so")  # env: kernel experiments only
ABI_VERSION = 8

GRAD_LOG, GRAD_LINEAR, GRAD_ACCUM = 0, 1, 2

_lib = None

_vp, _i, _f, _sz, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_int64

_SIGNATURES = {
    "pychain_hip_abi_version": (_i, []),
    "pychain_hip_last_error": (ctypes.c_char_p, []),
    "pychain_hip_set_verbose_level": (None, [_i]),
    "pychain_hip_get_verbose_level": (_i, []),
    "pychain_hip_set_den_phase_mask": (None, [_i]),
    "pychain_hip_set_den_lazy": (None, [_i]),
    "pychain_hip_den_recursion_is_lazy": (_i, [_i, _i, _i]),
    "pychain_hip_set_option": (_i, [ctypes.c_char_p, ctypes.c_char_p]),
    "pychain_hip_debug_launch_map": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i]),
    "pychain_hip_den_plan_build": (_i64, [_vp] * 9 + [_i, _i, _i, _vp, _sz]),
    "pychain_hip_den_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pychain_hip_den_plan_info": (_i, [_vp, _sz, _vp]),
    "pychain_hip_den_forward_backward": (_i, [_vp, _i64, _i, _i, _i, _vp, _i, _vp, _i, _i, _f, _f,
                                              _vp, _vp, _vp, _vp, _sz, _vp]),
    "pychain_hip_num_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "pychain_hip_num_forward_backward": (_i, [_vp] * 8 + [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f,
                                              _vp, _vp, _vp, _vp, _sz, _vp]),
    "pychain_hip_chain_loss_forward_backward": (_i, [_vp, _i64, _i, _i, _f] + [_vp] * 8 + [_i, _i, _i]
                                                + [_vp, _vp, _i, _i, _i, _f] + [_vp] * 4
                                                + [_vp, _sz, _vp, _sz, _vp]),
    "pychain_hip_chain_loss_forward": (_i, [_vp, _i64, _i, _i, _f] + [_vp] * 8 + [_i, _i, _i]
                                       + [_vp, _vp, _i, _i, _i] + [_vp, _vp, _vp, _f, _vp]
                                       + [_vp, _sz, _vp, _sz, _vp]),
    "pychain_hip_rescale": (_i, [_vp, _sz, _vp, _vp]),
    "pychain_hip_chain_loss_backward": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i,
                                             _f, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "pychain_hip_fst_read": (_vp, [ctypes.c_char_p, _i64]),
    "pychain_hip_fst_from_arcs": (_vp, [ctypes.c_int32, ctypes.c_int32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "pychain_hip_fst_free": (None, [_vp]),
    "pychain_hip_fst_write": (_i, [_vp, ctypes.c_char_p]),
    "pychain_hip_fst_num_states": (ctypes.c_int32, [_vp]),
    "pychain_hip_fst_start": (ctypes.c_int32, [_vp]),
    "pychain_hip_fst_num_arcs": (_i64, [_vp]),
    "pychain_hip_fst_to_tensors": (_i, [_vp, _i] + [_vp] * 7),
    "pychain_hip_fst_leaky_probs": (_i, [_vp, _vp]),
}
EXPORTS = tuple(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "pychain_amd: %s is missing. Build it with `python -m pychain_amd.build_ext` "
                "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback."
                % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        v = l.pychain_hip_abi_version()
        if v != ABI_VERSION:
            raise ImportError("pychain_amd: libpychain_hip.so has ABI %d, expected %d" % (v, ABI_VERSION))
        _lib = l
        # tuning scripts set PYCHAIN_<OPTION> in the environment: read ONCE here, never on the call path
        for name in OPTIONS:
            v = os.environ.get("PYCHAIN_" + name.upper())
            if v:
                l.pychain_hip_set_option(name.encode(), v.encode())
    return _lib


OPTIONS = ("den_segments", "den_relaunch", "den_bounds", "no_fold", "gamma16", "num_no_staging_waves", "den_pair")


class option(object):
    """`with _lib.option("den_segments", 3): ...` - a test / tuning option of the library for the duration
    of a block (include/pychain_hip.h: pychain_hip_set_option)."""

    def __init__(self, name, value=1):
        self.name, self.value = name, value

    def __enter__(self):
        check(lib().pychain_hip_set_option(self.name.encode(), str(self.value).encode()), "pychain_hip_set_option")
        return self

    def __exit__(self, *exc):
        lib().pychain_hip_set_option(self.name.encode(), None)
        return False


class PychainHipError(RuntimeError):
    pass


def check(rc, what):
    if rc < 0:
        raise PychainHipError("%s failed (%d): %s" % (what, rc, lib().pychain_hip_last_error().decode()))
    return rc
