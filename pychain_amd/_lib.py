"""ctypes binding of libpychain_hip.so (include/pychain_hip.h).

There is NO fallback: if the shared library is missing, or a call is made with
tensors that are not on a HIP device, this raises.  The product path never
touches the CPU checker.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYCHAIN_HIP_LIB") or os.path.join(_HERE, "libpychain_hip.so")  # env: kernel experiments only
ABI_VERSION = 17
TOTALS = 8            # floats of a `totals` buffer (include/pychain_hip.h: PYCHAIN_HIP_TOTALS)

GRAD_LOG, GRAD_LINEAR, GRAD_ACCUM = 0, 1, 2
CPU_NO_CLAMP = 0x100     # (host twin of the numerator only: include/pychain_hip.h)
F32, BF16, F16 = 0, 1, 2        # include/pychain_hip.h: PYCHAIN_HIP_F32 / _BF16 / _F16

_lib = None

_vp, _i, _f, _sz, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_int64

_SIGNATURES = {
    "pychain_hip_abi_version": (_i, []),
    "pychain_hip_last_error": (ctypes.c_char_p, []),
    "pychain_hip_set_verbose_level": (None, [_i]),
    "pychain_hip_get_verbose_level": (_i, []),
    "pychain_hip_set_den_phase_mask": (None, [_i]),
    "pychain_hip_set_den_lazy": (None, [_i]),
    "pychain_hip_den_kernel_names": (_i, [_i, _i, _i, _i, _i, ctypes.c_char_p, _sz]),
    "pychain_hip_set_option": (_i, [ctypes.c_char_p, ctypes.c_char_p]),
    "pychain_hip_set_thread_option": (_i, [ctypes.c_char_p, ctypes.c_char_p]),
    "pychain_hip_get_option": (_i, [ctypes.c_char_p, ctypes.c_char_p, _sz]),
    "pychain_hip_debug_launch_map": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i]),
    "pychain_hip_debug_stream_rings": (_i, [_i, _vp, _i, _vp, _i]),
    "pychain_hip_debug_occupy": (_i, [_i, _i, _vp]),
    "pychain_hip_den_plan_build": (_i64, [_vp] * 9 + [_i, _i, _i, _vp, _sz]),
    "pychain_hip_den_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pychain_hip_den_workspace_min_bytes": (_sz, [_i, _i, _i, _i]),
    "pychain_hip_den_plan_info": (_i, [_vp, _sz, _vp]),
    "pychain_hip_den_uses_row_buffer": (_i, [_i64, _i, _i, _i, _i, _i, _i]),
    "pychain_hip_den_time_segments": (_i, [_i64, _i, _i, _i, _i, _i, _i]),
    "pychain_hip_den_forward_backward": (_i, [_vp, _i64, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _f, _f,
                                              _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pychain_hip_den_half_native": (_i, [_i64, _i, _i, _i, _i, _i]),
    "pychain_hip_num_half_native": (_i, [_i, _i, _i]),
    "pychain_hip_chain_loss_half_native": (_i, [_i64, _i, _i, _i, _i, _i, _i, _i]),
    "pychain_hip_chain_loss_slices": (_i, [_i64, _i, _i]),
    "pychain_hip_num_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "pychain_hip_num_forward_backward": (_i, [_vp] * 8 + [_i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f,
                                              _vp, _vp, _vp, _vp, _sz, _vp]),
    "pychain_hip_chain_loss_forward_backward": (_i, [_vp, _i64, _i, _i, _f] + [_vp] * 8 + [_i, _i, _i]
                                                + [_vp, _i, _vp, _i, _i, _i, _f] + [_vp] * 4 + [_f, _vp, _vp]
                                                + [_vp, _sz, _vp, _sz, _vp]),
    "pychain_hip_chain_loss_forward": (_i, [_vp, _i64, _i, _i, _f] + [_vp] * 8 + [_i, _i, _i]
                                       + [_vp, _i, _vp, _i, _i, _i] + [_vp, _vp, _vp, _f, _vp] + [_f, _vp, _vp]
                                       + [_vp, _sz, _vp, _sz, _vp]),
    "pychain_hip_cpu_calls": (ctypes.c_long, []),
    "pychain_hip_cpu_den_forward_backward": (_i, [_vp] * 9 + [_i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _i]),
    "pychain_hip_cpu_num_forward_backward": (_i, [_vp] * 8 + [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i]),
    "pychain_hip_den_tseg_state": (_i, [_vp, _vp]),
    "pychain_hip_den_tseg_state_bytes": (_sz, []),
    "pychain_hip_rescale": (_i, [_vp, _i, _sz, _vp, _vp]),
    "pychain_hip_loss_total": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp]),
    "pychain_hip_chain_loss_backward": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _i,
                                             _f, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "pychain_hip_batch_layout": (_i64, [_i, _i, _i, _i, _vp, _vp]),
    "pychain_hip_batch_pack": (_i, [_i, _i, _i, _i, _vp, _vp, _sz]),
    "pychain_hip_batch_reorder": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pychain_hip_batch_reorder_dev": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
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


OPTIONS = ("verbose", "den_phase_mask", "den_lazy", "den_dma", "den_segments", "den_pair", "gamma16", "debug_corrupt_row", "num_compat", "den_tseg", "den_tburn", "plan_split", "chain_slices", "den_sg", "den_cross", "den_q")

_thread_values = threading.local()       # what this thread's overrides currently are (for restoring)


def get_option(name):
    """The value of a library option in effect for the calling thread (None = unset)."""
    buf = ctypes.create_string_buffer(256)
    n = check(lib().pychain_hip_get_option(name.encode(), buf, 256), "pychain_hip_get_option")
    return buf.value.decode() if n > 0 else None


class option(object):
    """`with _lib.option("den_segments", 3): ...` - a test / tuning option of the library for the duration of a
    block, FOR THE CALLING THREAD ONLY (include/pychain_hip.h: pychain_hip_set_thread_option); the previous
    override - or the process-wide default, e.g. one taken from PYCHAIN_<NAME> at load time - is back afterwards."""

    def __init__(self, name, value=1):
        self.name, self.value = name, value

    def __enter__(self):
        d = _thread_values.__dict__.setdefault("v", {})
        self.prev = d.get(self.name)
        d[self.name] = str(self.value)
        check(lib().pychain_hip_set_thread_option(self.name.encode(), str(self.value).encode()),
              "pychain_hip_set_thread_option")
        return self

    def __exit__(self, *exc):
        d = _thread_values.__dict__.setdefault("v", {})
        if self.prev is None:
            d.pop(self.name, None)
            lib().pychain_hip_set_thread_option(self.name.encode(), None)
        else:
            d[self.name] = self.prev
            lib().pychain_hip_set_thread_option(self.name.encode(), self.prev.encode())
        return False


def den_kernel_names(slot_rows, num_states, num_pdfs, batch, plans_shared=True, fused=False):
    """("recursion kernel", "occupancy kernel") a denominator call of this shape would launch (`fused`: as part of a fused loss)."""
    buf = ctypes.create_string_buffer(128)
    check(lib().pychain_hip_den_kernel_names(int(slot_rows), int(num_states), int(num_pdfs), int(batch),
                                             int(bool(plans_shared)) | (2 if fused else 0), buf, 128), "pychain_hip_den_kernel_names")
    rec, occ = buf.value.decode().split(",")
    return rec, occ


class PychainHipError(RuntimeError):
    pass


def check(rc, what):
    if rc < 0:
        raise PychainHipError("%s failed (%d): %s" % (what, rc, lib().pychain_hip_last_error().decode()))
    return rc
