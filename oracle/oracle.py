"""ctypes front-end of oracle/chain_oracle.c (CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY - importers allowed: tests/, bench.py's cpu_baseline
leg, __graft_entry__.smoke()/build().  Never imported by pychain_amd.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "chain_oracle.c")
_LIBS = {}


def lib_path(flavour):
    return os.path.join(_HERE, "libchain_oracle_%s.so" % flavour)


def build(force=False):
    """gcc the restatement in float32 and float64 flavours (in-tree, git-ignored)."""
    for flavour, real in (("f32", "float"), ("f64", "double")):
        out = lib_path(flavour)
        if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(_SRC):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-DREAL=" + real,
                                   "-DSUF=_" + flavour, _SRC, "-o", out, "-lm"])
    return [lib_path("f32"), lib_path("f64")]


def _lib(flavour):
    if flavour not in _LIBS:
        if not os.path.exists(lib_path(flavour)):
            build()
        _LIBS[flavour] = ctypes.CDLL(lib_path(flavour))
    return _LIBS[flavour]


def _c(t, dtype):
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _graph_args(gb, shared, with_leaky):
    """gb: ChainGraphBatch-like (batched [B,...] tensors).  If `shared`, row 0 is used for
    every sequence (graph_stride = 0)."""
    sel = (lambda t: t[:1]) if shared else (lambda t: t)
    names = ["forward_transitions", "forward_transition_indices", "forward_transition_probs",
             "backward_transitions", "backward_transition_indices", "backward_transition_probs"]
    if with_leaky:
        names.append("leaky_probs")
    names += ["initial_probs", "final_probs"]
    ints = ("forward_transitions", "forward_transition_indices",
            "backward_transitions", "backward_transition_indices")
    arrs = [_c(sel(getattr(gb, n)), np.int32 if n in ints else np.float32) for n in names]
    return arrs


def den(gb, exp_x, lengths, leaky_coef=1e-5, shared=None, flavour="f32"):
    """Reference `pychain_C.forward_backward` semantics (pychain.cc:26-79) on the CPU.
    Returns (objf_per_seq[B], grad[B,T,D], ok)."""
    if shared is None:
        shared = getattr(gb, "shared_graph", None) is not None
    real = np.float32 if flavour == "f32" else np.float64
    x = _c(exp_x, np.float32)
    B, T, D = x.shape
    L = _c(lengths, np.int64)
    g = _graph_args(gb, shared, True)
    H = g[1].shape[1]
    K = g[0].shape[1]
    objf = np.zeros(B, dtype=real)
    grad = np.empty((B, T, D), dtype=real)
    fn = getattr(_lib(flavour), "chain_oracle_den_" + flavour)
    fn.restype = ctypes.c_int
    ok = fn(*[_p(a) for a in g], ctypes.c_int(0 if shared else 1), _p(x), _p(L),
            ctypes.c_int(B), ctypes.c_int(T), ctypes.c_int(D), ctypes.c_int(H), ctypes.c_int(K),
            ctypes.c_float(leaky_coef), _p(objf), _p(grad))
    return objf, grad, bool(ok)


def num(gb, x_clamped, lengths, shared=False, flavour="f32"):
    """Reference `pychain_C.forward_backward_log_domain` semantics (pychain.cc:81-129).
    Returns (objf_per_seq[B], log_grad[B,T,D], ok)."""
    real = np.float32 if flavour == "f32" else np.float64
    x = _c(x_clamped, np.float32)
    B, T, D = x.shape
    L = _c(lengths, np.int64)
    g = _graph_args(gb, shared, False)
    H = g[1].shape[1]
    K = g[0].shape[1]
    objf = np.zeros(B, dtype=real)
    lg = np.empty((B, T, D), dtype=real)
    fn = getattr(_lib(flavour), "chain_oracle_num_" + flavour)
    fn.restype = ctypes.c_int
    ok = fn(*[_p(a) for a in g], ctypes.c_int(0 if shared else 1), _p(x), _p(L),
            ctypes.c_int(B), ctypes.c_int(T), ctypes.c_int(D), ctypes.c_int(H), ctypes.c_int(K),
            _p(objf), _p(lg))
    return objf, lg, bool(ok)


def chain_function(x, lengths, gb, leaky_coef=1e-5, flavour="f32"):
    """ChainFunction.forward + the saved gradient (pychain/loss.py:27-80):
    returns (objf scalar, input_grad[B,T,D])."""
    xc = torch.as_tensor(x).detach().cpu().float().contiguous().clamp(-30, 30)
    if not gb.log_domain:
        o, g, _ = den(gb, xc.exp(), lengths, leaky_coef, flavour=flavour)
        return o.sum(), g
    o, lg, _ = num(gb, xc, lengths, shared=getattr(gb, "shared_graph", None) is not None,
                   flavour=flavour)
    return o.sum(), np.exp(lg)


def chain_loss(x, lengths, den_graph, num_graphs, leaky_coef=1e-5, avg=True, flavour="f32"):
    """ChainLoss.forward + x.grad (pychain/loss.py:97-105)."""
    from pychain_amd.graph import ChainGraphBatch
    B = x.shape[0]
    d_o, d_g = chain_function(x, lengths, ChainGraphBatch(den_graph, B), leaky_coef, flavour)
    n_o, n_g = chain_function(x, lengths, num_graphs, flavour=flavour)
    loss = -(n_o - d_o)
    grad = d_g - n_g
    if avg:
        n = float(torch.as_tensor(lengths).sum())
        loss, grad = loss / n, grad / n
    return loss, grad
