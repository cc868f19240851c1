"""`simplefst`-compatible graph ingestion without OpenFST.

The reference's `simplefst` is a pybind11 wrapper over OpenFST 1.7.5
(openfst_binding/src/fstext.cc:174-184).  OpenFST is not on this image and is
not on the hot path; what IS on the path is the tensor layout `FstToTensor`
defines (fstext.cc:19-117) and the `SetLeakyProbs` recipe (fstext.cc:120-171).
This module restates both over a plain arc list so `ChainGraph(fst, ...)` keeps
its reference signature.

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

import numpy as np
import torch

__all__ = ["StdVectorFst"]

_INF = float("inf")


class StdVectorFst(object):
    """Minimal mutable FST in the tropical semiring (weights are -log probs)."""

    def __init__(self):
        self._start = -1
        self._final = []   # weight per state; +inf = not final
        self._arcs = []    # per state: list of (ilabel, olabel, weight, nextstate)

    # ---- construction -------------------------------------------------
    def add_state(self):
        self._final.append(_INF)
        self._arcs.append([])
        return len(self._final) - 1

    def set_start(self, s):
        self._start = int(s)

    def set_final(self, s, weight=0.0):
        self._final[int(s)] = float(weight)

    def add_arc(self, s, ilabel, olabel, weight, nextstate):
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
        """Returns the 7 tensors in the order of fstext.cc:109-116."""
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
    def set_leaky_probs(fst):
        """Averaged 100-step occupancy started from the start state
        (fstext.cc:120-171).  float64 internally, float32 result."""
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
        with open(filename, "wb") as f:
            f.write(self._to_bytes())
        return True

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
        for _ in range(nstates):
            fw, na = rd("<fq")
            st = fst.add_state()
            fst._final[st] = fw
            for _a in range(na):
                il, ol, w, ns = rd("<iifi")
                fst._arcs[st].append((il, ol, w, ns))
        fst._start = start
        return fst

    @classmethod
    def read(cls, filename):
        with open(filename, "rb") as f:
            return cls._from_stream(f, filename)

    @classmethod
    def read_ark(cls, filename, offset):
        with open(filename, "rb") as f:
            f.seek(offset)
            return cls._from_stream(f, filename)
