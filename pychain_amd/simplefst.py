"""`simplefst`-compatible graph ingestion without OpenFST.

The reference's `simplefst` is a pybind11 wrapper over OpenFST 1.7.5
(openfst_binding/src/fstext.cc:174-184).  OpenFST is not on this image and is
not on the hot path; what IS on the path is the tensor layout `FstToTensor`
defines (fstext.cc:19-117) and the `SetLeakyProbs` recipe (fstext.cc:120-171).
This module restates both so `ChainGraph(fst, ...)` keeps its reference signature.
The heavy lifting (binary FST I/O, FstToTensor, SetLeakyProbs) is native C++ behind the
C ABI (pychain_amd/csrc/fst.cpp - the reference's is C++ too); the pure-Python
restatements below (`_py_*`) are kept as an independent second implementation that the
tests hold the native one against.

Layout rules reproduced (file:line in the reference):
  * pdf_id   = ilabel - 1                       fstext.cc:41
  * log_prob = -arc.weight                      fstext.cc:43-44
  * final    = -Final(s)                        fstext.cc:37
  * out-arcs listed per source state in insertion order            :49-61
  * in-arcs  listed per destination state, by ascending source
    state then insertion order (because the outer loop is over s)  :36-46, :63-76
  * probabilities exp()'d unless log_domain                        :89-107
  * leaky probs: float64, 100 iterations, returned as float32      :125-170
"""
import math
import struct

import ctypes

import numpy as np
import torch

from . import _lib

__all__ = ["StdVectorFst"]

_INF = float("inf")


class StdVectorFst(object):
    """Minimal mutable FST in the tropical semiring (weights are -log probs)."""

    def __init__(self):
        self._start = -1
        self._final = []   # weight per state; +inf = not final
        self._arcs = []    # per state: list of (ilabel, olabel, weight, nextstate)
        self._handle = None

    def __del__(self):
        self._drop_native()

    def _drop_native(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.lib().pychain_hip_fst_free(h)
            except Exception:
                pass
        self._handle = None

    def _native(self):
        """Opaque handle of the C++ twin of this FST (rebuilt after mutation)."""
        if self._handle is None:
            src, dst, il, w = [], [], [], []
            for s, arcs in enumerate(self._arcs):
                for (i, _o, wt, ns) in arcs:
                    src.append(s); dst.append(ns); il.append(i); w.append(wt)
            a = lambda v, t: np.ascontiguousarray(np.asarray(v, dtype=t))
            src, dst, il, w = a(src, np.int32), a(dst, np.int32), a(il, np.int32), a(w, np.float32)
            fin = a(self._final, np.float32)
            p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
            h = _lib.lib().pychain_hip_fst_from_arcs(len(self._final), int(self._start), len(src),
                                                     p(src), p(dst), p(il), p(w), p(fin))
            if not h:
                raise _lib.PychainHipError(_lib.lib().pychain_hip_last_error().decode())
            self._handle = h
        return self._handle

    # ---- construction -------------------------------------------------
    def add_state(self):
        self._drop_native()
        self._final.append(_INF)
        self._arcs.append([])
        return len(self._final) - 1

    def set_start(self, s):
        self._drop_native()
        self._start = int(s)

    def set_final(self, s, weight=0.0):
        self._drop_native()
        self._final[int(s)] = float(weight)

    def add_arc(self, s, ilabel, olabel, weight, nextstate):
        self._drop_native()
        self._arcs[int(s)].append((int(ilabel), int(olabel), float(weight), int(nextstate)))

    @classmethod
    def from_arcs(cls, num_states, start, arcs, finals):
        """arcs: iterable of (src, dst, pdf_id, log_prob); finals: {state: log_prob}
        or a length-num_states sequence of final log-probs (-inf = not final)."""
        f = cls()
        for _ in range(num_states):
            f.add_state()
        f.set_start(start)
        for (src, dst, pdf, lp) in arcs:
            f.add_arc(src, pdf + 1, pdf + 1, -float(lp), dst)
        if isinstance(finals, dict):
            for s, lp in finals.items():
                f.set_final(s, -float(lp))
        else:
            for s, lp in enumerate(finals):
                if lp != -_INF:
                    f.set_final(s, -float(lp))
        return f

    @classmethod
    def from_arrays(cls, num_states, start, src, dst, pdf, log_prob, final_log_prob):
        """Vectorised constructor used for large synthetic graphs."""
        f = cls()
        f._final = [(-float(v) if v != -_INF else _INF) for v in final_log_prob]
        f._arcs = [[] for _ in range(num_states)]
        for s, d, n, lp in zip(src.tolist(), dst.tolist(), pdf.tolist(), log_prob.tolist()):
            f._arcs[s].append((n + 1, n + 1, -lp, d))
        f._start = int(start)
        return f

    # ---- reference-visible API (fstext.cc:174-184) ----------------------
    def num_states(self):
        return len(self._final)

    def start_state(self):  # also callable as StdVectorFst.start_state(fst)
        return self._start

    @staticmethod
    def fst_to_tensor(fst, log_domain=False):
        """Returns the 7 tensors in the order of fstext.cc:109-116 (native C++)."""
        L = _lib.lib()
        h = fst._native()
        H, K = int(L.pychain_hip_fst_num_states(h)), int(L.pychain_hip_fst_num_arcs(h))
        ft, bt = torch.empty((K, 3), dtype=torch.int32), torch.empty((K, 3), dtype=torch.int32)
        fp, bp = torch.empty(K, dtype=torch.float32), torch.empty(K, dtype=torch.float32)
        fi, bi = torch.empty((H, 2), dtype=torch.int32), torch.empty((H, 2), dtype=torch.int32)
        fin = torch.empty(H, dtype=torch.float32)
        _lib.check(L.pychain_hip_fst_to_tensors(h, int(bool(log_domain)), ft.data_ptr(), fp.data_ptr(), fi.data_ptr(),
                                                bt.data_ptr(), bp.data_ptr(), bi.data_ptr(), fin.data_ptr()),
                   "pychain_hip_fst_to_tensors")
        return [ft, fp, fi, bt, bp, bi, fin]

    @staticmethod
    def set_leaky_probs(fst):
        """Averaged 100-step occupancy started from the start state (fstext.cc:120-171, native C++)."""
        L = _lib.lib()
        h = fst._native()
        out = torch.empty(int(L.pychain_hip_fst_num_states(h)), dtype=torch.float32)
        _lib.check(L.pychain_hip_fst_leaky_probs(h, out.data_ptr()), "pychain_hip_fst_leaky_probs")
        return out

    @staticmethod
    def _py_fst_to_tensor(fst, log_domain=False):
        """Pure-Python restatement of FstToTensor (second opinion for the tests)."""
        H = fst.num_states()
        out_src, out_dst, out_pdf, out_lp = [], [], [], []
        in_lists = [[] for _ in range(H)]
        fwd_idx = np.zeros((H, 2), dtype=np.int32)
        for s in range(H):
            fwd_idx[s, 0] = len(out_src)
            for (il, _ol, w, ns) in fst._arcs[s]:
                pdf = il - 1
                assert pdf >= 0, "epsilon input labels are not allowed"
                out_src.append(s); out_dst.append(ns); out_pdf.append(pdf); out_lp.append(-w)
                in_lists[ns].append((s, ns, pdf, -w))
            fwd_idx[s, 1] = len(out_src)
        K = len(out_src)
        fwd = np.zeros((K, 3), dtype=np.int32)
        fwd[:, 0] = out_src; fwd[:, 1] = out_dst; fwd[:, 2] = out_pdf
        fwd_lp = np.asarray(out_lp, dtype=np.float32)

        bwd = np.zeros((K, 3), dtype=np.int32)
        bwd_lp = np.zeros((K,), dtype=np.float32)
        bwd_idx = np.zeros((H, 2), dtype=np.int32)
        k = 0
        for s in range(H):
            bwd_idx[s, 0] = k
            for (a, b, n, lp) in in_lists[s]:
                bwd[k] = (a, b, n); bwd_lp[k] = lp
                k += 1
            bwd_idx[s, 1] = k
        final = np.asarray([-w for w in fst._final], dtype=np.float32)

        fwd_p = torch.from_numpy(fwd_lp)
        bwd_p = torch.from_numpy(bwd_lp)
        fin = torch.from_numpy(final)
        if not log_domain:
            fwd_p = fwd_p.exp_(); bwd_p = bwd_p.exp_(); fin = fin.exp_()
        return [torch.from_numpy(fwd), fwd_p, torch.from_numpy(fwd_idx),
                torch.from_numpy(bwd), bwd_p, torch.from_numpy(bwd_idx), fin]

    @staticmethod
    def _py_set_leaky_probs(fst):
        """Pure-Python restatement of SetLeakyProbs (second opinion for the tests)."""
        num_iters = 100
        H = fst.num_states()
        src, dst, p = [], [], []
        nf = np.zeros(H, dtype=np.float64)
        for s in range(H):
            tot = math.exp(-fst._final[s]) if fst._final[s] != _INF else 0.0
            for (_il, _ol, w, ns) in fst._arcs[s]:
                pw = math.exp(-w)
                tot += pw
                src.append(s); dst.append(ns); p.append(pw)
            nf[s] = 1.0 / tot
        src = np.asarray(src, dtype=np.int64); dst = np.asarray(dst, dtype=np.int64)
        p = np.asarray(p, dtype=np.float64)
        cur = np.zeros(H, dtype=np.float64)
        avg = np.zeros(H, dtype=np.float64)
        cur[fst._start] = 1.0
        for _ in range(num_iters):
            avg += cur * (1.0 / num_iters)
            nxt = np.zeros(H, dtype=np.float64)
            np.add.at(nxt, dst, (cur * nf)[src] * p)
            cur = nxt * (1.0 / nxt.sum())
        return torch.from_numpy(avg.astype(np.float32))

    # ---- OpenFST binary I/O (vector / standard), fstext.cc:7-16, :177-179 ----
    _MAGIC = 2125659606

    def write(self, filename):
        _lib.check(_lib.lib().pychain_hip_fst_write(self._native(), str(filename).encode()), "pychain_hip_fst_write")
        return True

    @classmethod
    def _from_native(cls, h):
        """Python twin of a native FST (so it stays inspectable/mutable)."""
        L = _lib.lib()
        f = cls()
        f._handle = h
        ts = cls.fst_to_tensor(f, log_domain=True)
        ft, fp, fin = ts[0].numpy(), ts[1].numpy(), ts[6].numpy()
        f._final = [(-float(v) if v != -_INF else _INF) for v in fin]
        f._arcs = [[] for _ in range(len(f._final))]
        for (s, d, n), lp in zip(ft.tolist(), fp.tolist()):
            f._arcs[s].append((n + 1, n + 1, -lp, d))
        f._start = int(L.pychain_hip_fst_start(h))
        return f

    def _to_bytes(self):
        def s(b):
            return struct.pack("<i", len(b)) + b
        H = self.num_states()
        narcs = sum(len(a) for a in self._arcs)
        out = [struct.pack("<i", self._MAGIC), s(b"vector"), s(b"standard"),
               struct.pack("<iiQqqq", 2, 0, 0, self._start, H, narcs)]
        for st in range(H):
            out.append(struct.pack("<fq", self._final[st], len(self._arcs[st])))
            for (il, ol, w, ns) in self._arcs[st]:
                out.append(struct.pack("<iifi", il, ol, w, ns))
        return b"".join(out)

    @classmethod
    def _from_stream(cls, f, name):
        def rd(fmt):
            n = struct.calcsize(fmt)
            b = f.read(n)
            if len(b) != n:
                raise IOError("truncated FST in " + name)
            return struct.unpack(fmt, b)

        def rstr():
            (n,) = rd("<i")
            return f.read(n)
        (magic,) = rd("<i")
        if magic != cls._MAGIC:
            raise IOError("bad FST magic in %s" % name)
        ftype, atype = rstr(), rstr()
        if ftype != b"vector" or atype != b"standard":
            raise IOError("only vector/standard FSTs are supported, got %r/%r" % (ftype, atype))
        version, flags, _props, start, nstates, _narcs = rd("<iiQqqq")
        if flags & 1 or flags & 2:
            raise IOError("FSTs with embedded symbol tables are not supported")
        if version < 2:
            raise IOError("unsupported vector FST version %d" % version)
        fst = cls()
        n = 0
        while nstates < 0 or n < nstates:          # -1: unknown, states follow until the stream ends
            b = f.read(4)
            if len(b) != 4:
                if nstates < 0:
                    break
                raise IOError("truncated FST in " + name)
            (fw,) = struct.unpack("<f", b)
            (na,) = rd("<q")
            st = fst.add_state()
            fst._final[st] = fw
            for _a in range(na):
                il, ol, w, ns = rd("<iifi")
                fst._arcs[st].append((il, ol, w, ns))
            n += 1
        fst._start = start
        return fst

    @classmethod
    def read(cls, filename):
        return cls.read_ark(filename, 0)

    @classmethod
    def read_ark(cls, filename, offset):
        h = _lib.lib().pychain_hip_fst_read(str(filename).encode(), int(offset))
        if not h:
            raise IOError(_lib.lib().pychain_hip_last_error().decode())
        return cls._from_native(h)

    @classmethod
    def _py_read(cls, filename, offset=0):
        with open(filename, "rb") as f:
            f.seek(offset)
            return cls._from_stream(f, filename)
