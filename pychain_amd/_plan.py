"""Compiled denominator plans: host build (C++) + device residency (cached).

Replaces the reference's per-call `.cuda()` of eleven graph tensors
(chain-computation.cc:77-89): a graph is compiled once and its plan stays on
the device for as long as the ChainGraph object lives.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_NAMES = ["forward_transitions", "forward_transition_indices", "forward_transition_probs",
          "backward_transitions", "backward_transition_indices", "backward_transition_probs",
          "leaky_probs", "initial_probs", "final_probs"]


class DevicePlan(object):
    """Device-resident plan(s): `blob` uint8 tensor, byte `stride` between per-sequence
    plans (0 = shared), `slot_rows` = arcs per wave the kernels keep in registers."""

    def __init__(self, blob, stride, slot_rows, num_states):
        self.blob, self.stride, self.slot_rows, self.num_states = blob, int(stride), int(slot_rows), int(num_states)


def _np(t, dtype):
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=dtype)


def build_plan_blob(forward_transitions, forward_transition_indices, forward_transition_probs,
                    backward_transitions, backward_transition_indices, backward_transition_probs,
                    leaky_probs, initial_probs, final_probs, num_pdfs):
    """Host tensors of ONE graph -> numpy uint8 plan blob (pychain_hip_den_plan_build)."""
    arrs = [_np(forward_transitions, np.int32), _np(forward_transition_indices, np.int32),
            _np(forward_transition_probs, np.float32),
            _np(backward_transitions, np.int32), _np(backward_transition_indices, np.int32),
            _np(backward_transition_probs, np.float32),
            _np(leaky_probs, np.float32), _np(initial_probs, np.float32), _np(final_probs, np.float32)]
    H = arrs[1].shape[0]
    K = arrs[0].shape[0]
    ptrs = [a.ctypes.data_as(ctypes.c_void_p) for a in arrs]
    L = _lib.lib()
    need = _lib.check(L.pychain_hip_den_plan_build(*ptrs, H, K, int(num_pdfs), None, 0), "den_plan_build")
    blob = np.zeros(int(need), dtype=np.uint8)
    _lib.check(L.pychain_hip_den_plan_build(*ptrs, H, K, int(num_pdfs),
                                            blob.ctypes.data_as(ctypes.c_void_p), blob.nbytes),
               "den_plan_build")
    return blob


def plan_info(blob):
    info = np.zeros(8, dtype=np.int32)
    _lib.check(_lib.lib().pychain_hip_den_plan_info(blob.ctypes.data_as(ctypes.c_void_p), blob.nbytes,
                                                    info.ctypes.data_as(ctypes.c_void_p)), "den_plan_info")
    return dict(num_states=int(info[0]), num_transitions=int(info[1]), num_pdfs=int(info[2]),
                bytes=int(info[3]), slot_rows=int(info[4]))


def graph_plan(graph, num_pdfs, device):
    """DevicePlan of a ChainGraph (shared denominator), cached on the graph."""
    key = (str(device), int(num_pdfs))
    if not hasattr(graph, "_plan_cache"):
        graph._plan_cache = {}
    hit = graph._plan_cache.get(key)
    if hit is None:
        blob = build_plan_blob(*[getattr(graph, n) for n in _NAMES], num_pdfs)
        hit = DevicePlan(torch.from_numpy(blob).to(device), 0, plan_info(blob)["slot_rows"], graph.num_states)
        graph._plan_cache[key] = hit
    return hit


def batch_plans(tensors, num_pdfs, device):
    """Per-sequence probability-domain graphs ([B,...] tensors) -> DevicePlan.
    Rows that are all identical collapse to one shared plan (stride 0)."""
    ts = [tensors[n].detach().cpu() for n in _NAMES]
    B = ts[0].shape[0]
    H = ts[1].shape[1]
    same = all(bool((t[1:] == t[:1]).all()) for t in ts) if B > 1 else True
    rows = [0] if same else range(B)
    blobs = [build_plan_blob(*[t[b] for t in ts], num_pdfs) for b in rows]
    hints = [plan_info(b)["slot_rows"] for b in blobs]
    slot_rows = sum(max((h >> sh) & 1023 for h in hints) << sh for sh in (0, 10, 20))
    if all((h >> 30) & 1 for h in hints):      # every plan fits the lazy-normalisation recursion (<= 4 groups per wave)
        slot_rows |= 1 << 30
    if same:
        return DevicePlan(torch.from_numpy(blobs[0]).to(device), 0, slot_rows, H)
    stride = (max(b.nbytes for b in blobs) + 255) // 256 * 256
    allb = np.zeros(stride * B, dtype=np.uint8)
    for i, b in enumerate(blobs):
        allb[i * stride:i * stride + b.nbytes] = b
    return DevicePlan(torch.from_numpy(allb).to(device), stride, slot_rows, H)
