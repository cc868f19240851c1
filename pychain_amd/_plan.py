"""Compiled denominator plans: host build (C++) + device residency (cached).

Replaces the reference's per-call `.cuda()` of eleven graph tensors
(chain-computation.cc:77-89): a graph is compiled once and its plan stays on
the device for as long as the ChainGraph object lives.
"""
import ctypes
import hashlib
import os
import tempfile

import numpy as np
import torch

from . import _lib

HINT_GENERAL = -(1 << 31)      # PYCHAIN_HIP_HINT_GENERAL: a plan in the general format (any H, K, D; den_general.hip)

_NAMES = ["forward_transitions", "forward_transition_indices", "forward_transition_probs",
          "backward_transitions", "backward_transition_indices", "backward_transition_probs",
          "leaky_probs", "initial_probs", "final_probs"]


class DevicePlan(object):
    """Device-resident plan(s): `blob` uint8 tensor, byte `stride` between per-sequence
    plans (0 = shared), `slot_rows` = arcs per wave the kernels keep in registers."""

    def __init__(self, blob, stride, slot_rows, num_states):
        self.blob, self.stride, self.slot_rows, self.num_states = blob, int(stride), int(slot_rows), int(num_states)
        # the plan's burn-in controller (include/pychain_hip.h: pychain_hip_den_tseg_state): 64 bytes on the device, zeroed once,
        # attached by plan address - the kernels of a time-segmented call read and update it in stream order, nothing on the
        # host ever reads it.  Detached when the plan goes.
        self.tseg_state = None
        if blob.is_cuda and self.stride == 0:
            self.tseg_state = torch.zeros(int(_lib.lib().pychain_hip_den_tseg_state_bytes()) // 4, dtype=torch.int32, device=blob.device)
            _lib.check(_lib.lib().pychain_hip_den_tseg_state(blob.data_ptr(), self.tseg_state.data_ptr()), "pychain_hip_den_tseg_state")
            self._attached = blob.data_ptr()

    def __del__(self):
        try:
            if getattr(self, "_attached", None):
                _lib.lib().pychain_hip_den_tseg_state(self._attached, None)
        except Exception:          # (interpreter shutdown)
            pass


def _np(t, dtype):
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=dtype)


def build_plan_blob(forward_transitions, forward_transition_indices, forward_transition_probs,
                    backward_transitions, backward_transition_indices, backward_transition_probs,
                    leaky_probs, initial_probs, final_probs, num_pdfs, use_cache=True):
    """Host tensors of ONE graph -> numpy uint8 plan blob (pychain_hip_den_plan_build), through the
    on-disk cache."""
    arrs = [_np(forward_transitions, np.int32), _np(forward_transition_indices, np.int32),
            _np(forward_transition_probs, np.float32),
            _np(backward_transitions, np.int32), _np(backward_transition_indices, np.int32),
            _np(backward_transition_probs, np.float32),
            _np(leaky_probs, np.float32), _np(initial_probs, np.float32), _np(final_probs, np.float32)]
    H = arrs[1].shape[0]
    K = arrs[0].shape[0]
    ptrs = [a.ctypes.data_as(ctypes.c_void_p) for a in arrs]
    L = _lib.lib()
    cdir = _cache_dir() if use_cache and K >= 256 else None       # (tiny graphs compile in microseconds)
    path = None
    if cdir is not None:
        path = os.path.join(cdir, _plan_key(arrs, num_pdfs) + ".plan")
        try:
            blob = np.fromfile(path, dtype=np.uint8)
            if blob.nbytes >= 32:
                info = np.zeros(8, dtype=np.int32)
                rc = L.pychain_hip_den_plan_info(blob.ctypes.data_as(ctypes.c_void_p), blob.nbytes,
                                                 info.ctypes.data_as(ctypes.c_void_p))
                # (plan_info also verifies the checksum of the payload behind the header)
                # (info[6]: the graph's states where the plan has put some on several lanes - info[0] then counts positions)
                if rc == 0 and (int(info[6]) or int(info[0]), int(info[1]), int(info[2]), int(info[3]) + (int(info[5]) << 31)) == (H, K, int(num_pdfs), blob.nbytes):
                    return blob
        except (OSError, ValueError):
            pass
    need = _lib.check(L.pychain_hip_den_plan_build(*ptrs, H, K, int(num_pdfs), None, 0), "den_plan_build")
    blob = np.zeros(int(need), dtype=np.uint8)
    _lib.check(L.pychain_hip_den_plan_build(*ptrs, H, K, int(num_pdfs),
                                            blob.ctypes.data_as(ctypes.c_void_p), blob.nbytes),
               "den_plan_build")
    if path is not None:
        try:
            os.makedirs(cdir, mode=0o700, exist_ok=True)
            fd, tmp = tempfile.mkstemp(dir=cdir, suffix=".tmp")
            with os.fdopen(fd, "wb") as f:
                f.write(blob.tobytes())
            os.replace(tmp, path)
        except OSError:
            pass                      # a read-only home directory only costs the compile time
    return blob


def plan_info(blob):
    info = np.zeros(8, dtype=np.int32)
    _lib.check(_lib.lib().pychain_hip_den_plan_info(blob.ctypes.data_as(ctypes.c_void_p), blob.nbytes,
                                                    info.ctypes.data_as(ctypes.c_void_p)), "den_plan_info")
    # num_states: what the calls are given as num_states - the plan's POSITIONS (a state with many arcs may sit on several:
    # csrc/plan.cpp, "states on several lanes"); graph_states: the graph's own count
    return dict(num_states=int(info[0]), num_transitions=int(info[1]), num_pdfs=int(info[2]),
                bytes=int(info[3]) + (int(info[5]) << 31), slot_rows=int(info[4]),
                graph_states=int(info[6]) or int(info[0]), split_positions=int(info[7]))


# ---- on-disk cache of compiled plans -------------------------------------------------------------
# Compiling the C3 graph takes ~5 s (slot-order annealing); N ranks of a data-parallel job and every
# restart would each pay it.  A plan is a pure function of the nine graph tensors, the pdf count, the plan
# format version, the compiler knobs and the library build, so it is stored under a hash of exactly those
# ($PYCHAIN_PLAN_CACHE_DIR, default ~/.cache/pychain_amd/plans, created 0700; "0" / "off" disables).  Writes are
# atomic (temp file + rename): ranks racing on one graph all end up with the same bytes.  A file is only
# believed if its header matches the request AND its payload matches the checksum in the header.
_KNOBS = ("PYCHAIN_PLAN_GENERAL", "PYCHAIN_PLAN_SLACK", "PYCHAIN_PLAN_BALANCE", "PYCHAIN_PLAN_ANNEAL", "PYCHAIN_PLAN_FIT", "PYCHAIN_PLAN_LINEAR", "PYCHAIN_PLAN_SPLIT", "PYCHAIN_PLAN_GAMMA_BOUND", "PYCHAIN_PLAN_SG")


def _cache_dir():
    d = os.environ.get("PYCHAIN_PLAN_CACHE_DIR")
    if d is not None and d.strip().lower() in ("", "0", "off", "none"):
        return None
    return d or os.path.join(os.path.expanduser("~"), ".cache", "pychain_amd", "plans")


_BUILD_ID = None


def _build_id():
    """sha256 of the library file: a rebuilt plan compiler (other annealing, other slack rules) must not be served
    plans its predecessor wrote, whether or not somebody remembered to bump PLAN_VERSION."""
    global _BUILD_ID
    if _BUILD_ID is None:
        h = hashlib.sha256()
        with open(_lib.LIB_PATH, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        _BUILD_ID = h.hexdigest()[:24]
    return _BUILD_ID


def _plan_key(arrays, num_pdfs):
    h = hashlib.sha256()
    h.update(b"pychain_amd plan v%d abi %d pdfs %d build %s" % (_plan_format_version(), _lib.ABI_VERSION, int(num_pdfs),
                                                                 _build_id().encode()))
    for k in _KNOBS:
        h.update(("%s=%s;" % (k, os.environ.get(k, ""))).encode())
    h.update(("plan_split=%s;" % (_lib.get_option("plan_split") or "")).encode())
    for a in arrays:
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _plan_format_version():
    # PLAN_VERSION of csrc/plan_format.h as the library reports it in a plan header (bytes 4..8)
    global _PLAN_VERSION
    if _PLAN_VERSION is None:
        t = torch.tensor
        blob = build_plan_blob(t([[0, 0, 0]], dtype=torch.int32), t([[0, 1]], dtype=torch.int32), t([1.0]),
                               t([[0, 0, 0]], dtype=torch.int32), t([[0, 1]], dtype=torch.int32), t([1.0]),
                               t([1.0]), t([1.0]), t([1.0]), 1, use_cache=False)
        _PLAN_VERSION = int(np.frombuffer(blob[4:8].tobytes(), dtype=np.int32)[0])
    return _PLAN_VERSION


_PLAN_VERSION = None


def _tensor_versions(graph):
    """(data_ptr, _version) of the nine graph tensors: an in-place edit (final_probs.fill_, new leaky
    probs) or a replaced tensor makes the cached plan stale."""
    return tuple((getattr(graph, n).data_ptr(), getattr(graph, n)._version) for n in _NAMES)


def graph_plan(graph, num_pdfs, device):
    """DevicePlan of a ChainGraph (shared denominator), cached on the graph until one of its tensors
    is modified or replaced."""
    key = (str(device), int(num_pdfs), _lib.get_option("plan_split"))
    if not hasattr(graph, "_plan_cache"):
        graph._plan_cache = {}
    ver = _tensor_versions(graph)
    hit = graph._plan_cache.get(key)
    if hit is None or hit[0] != ver:
        blob = build_plan_blob(*[getattr(graph, n) for n in _NAMES], num_pdfs)
        info = plan_info(blob)
        plan = DevicePlan(torch.from_numpy(blob).to(device), 0, info["slot_rows"], info["num_states"])
        graph._plan_cache[key] = hit = (ver, plan)
    return hit[1]


def batch_plans(tensors, num_pdfs, device):
    """Per-sequence probability-domain graphs ([B,...] tensors) -> DevicePlan.
    Rows that are all identical collapse to one shared plan (stride 0)."""
    ts = [tensors[n].detach().cpu() for n in _NAMES]
    B = ts[0].shape[0]
    H = ts[1].shape[1]
    same = all(bool((t[1:] == t[:1]).all()) for t in ts) if B > 1 else True
    rows = [0] if same else range(B)
    # (the compiler is native code behind ctypes, which releases the GIL: the plans of a list of graphs compile side by side)
    if len(rows) > 1:
        # (plans passed with a stride share one position count: no state on several lanes - option plan_split is per thread)
        def build(b):
            with _lib.option("plan_split", "0"):
                return build_plan_blob(*[t[b] for t in ts], num_pdfs)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(rows), os.cpu_count() or 1, 16)) as ex:
            blobs = list(ex.map(build, rows))
    else:
        blobs = [build_plan_blob(*[t[b] for t in ts], num_pdfs) for b in rows]
    infos = [plan_info(b) for b in blobs]
    hints = [i["slot_rows"] for i in infos]
    H = max(i["num_states"] for i in infos)
    assert all(i["num_states"] == H for i in infos), "plans of one batch must have the same number of positions"
    if any(h == HINT_GENERAL for h in hints):
        # the format follows from the sizes (and PYCHAIN_PLAN_GENERAL): all plans of one batch are alike
        assert all(h == HINT_GENERAL for h in hints)
        slot_rows = HINT_GENERAL
    else:
        slot_rows = sum(max((h >> sh) & mask for h in hints) << sh for sh, mask in ((0, 1023), (10, 511), (20, 255)))
        if any((h >> 28) & 1 for h in hints):      # a state on several beta positions: not for the pair kernel
            slot_rows |= 1 << 28
        for bit in (19, 29, 30):                   # every plan takes one-word state vectors / holds four-wave tiles / fits the lazy recursion (<= 4 groups per wave)
            if all((h >> bit) & 1 for h in hints):
                slot_rows |= 1 << bit
    if same:
        return DevicePlan(torch.from_numpy(blobs[0]).to(device), 0, slot_rows, H)
    stride = (max(b.nbytes for b in blobs) + 255) // 256 * 256
    allb = np.zeros(stride * B, dtype=np.uint8)
    for i, b in enumerate(blobs):
        allb[i * stride:i * stride + b.nbytes] = b
    return DevicePlan(torch.from_numpy(allb).to(device), stride, slot_rows, H)
