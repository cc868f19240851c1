"""ChainGraph / ChainGraphBatch: the tensor containers of the LF-MMI path.

Mirror of the reference's pychain/graph.py (same class names, constructor
signatures, attribute names, shapes, dtypes and error behaviour; §8(a) rows A4
and A5 of SURVEY.md), re-implemented for a device-resident, shared-graph hot
path:

  * `ChainGraphBatch(one_graph, B)` keeps stride-0 *views* of the single graph
    instead of B physical copies (reference: `.repeat(B, ...)`,
    pychain/graph.py:101-120) and remembers the source graph, so the HIP
    denominator reads ONE compiled plan instead of B replicas;
  * a `ChainGraph` lazily owns its compiled device plan (see
    `pychain_amd/_plan.py`), built once and cached per device, instead of an
    H2D copy of every tensor on every call (chain-computation.cc:77-89).

All tensors live on the CPU exactly as in the reference.
"""
import ctypes

import numpy as np
import torch

from . import simplefst

__all__ = ["ChainGraph", "ChainGraphBatch"]


class ChainGraph(object):
    """One FST as 7 tensors + leaky/initial probs (pychain/graph.py:23-70)."""

    def __init__(self, fst, initial_mode="fst", final_mode="fst", log_domain=False):
        self.num_states = fst.num_states()
        assert initial_mode in ["fst", "leaky"]
        assert final_mode in ["fst", "ones"]
        self.log_domain = log_domain
        sfst = type(fst) if hasattr(type(fst), "fst_to_tensor") else simplefst.StdVectorFst
        (self.forward_transitions,
         self.forward_transition_probs,
         self.forward_transition_indices,
         self.backward_transitions,
         self.backward_transition_probs,
         self.backward_transition_indices,
         self.final_probs) = sfst.fst_to_tensor(fst, log_domain)
        self.num_transitions = self.forward_transitions.size(0)
        self.is_empty = (self.num_transitions == 0)
        self.start_state = sfst.start_state(fst)
        if self.is_empty:
            raise Exception("An empty graph encountered!")
        ptype = self.forward_transition_probs.dtype
        if log_domain:
            self.leaky_probs = None  # no leaky-HMM in the log domain
            assert initial_mode == "fst", "'leaky' mode is incompatible with log domain"
            self.initial_probs = torch.full([self.num_states], float("-inf"), dtype=ptype)
            self.initial_probs[self.start_state] = 0.0
            if final_mode == "ones":
                self.final_probs.fill_(0.0)
        else:
            self.leaky_probs = sfst.set_leaky_probs(fst)
            if initial_mode == "fst":
                self.initial_probs = torch.zeros([self.num_states], dtype=ptype)
                self.initial_probs[self.start_state] = 1.0
            else:
                self.initial_probs = self.leaky_probs.clone()
            if final_mode == "ones":
                self.final_probs.fill_(1.0)
        self._plan_cache = {}

    def __getstate__(self):
        # The caches hold a compiled device plan and RAW HOST ADDRESSES of this object's tensors (_pack_record): neither
        # may travel with a copy.deepcopy / pickle of the graph (DataLoader workers, torch.save) - the copy's tensors live
        # somewhere else, and in another process the addresses mean nothing.
        d = dict(self.__dict__)
        d.pop("_pack_cache", None)
        d["_plan_cache"] = {}
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.__dict__.pop("_pack_cache", None)
        self._plan_cache = {}

    @classmethod
    def from_tensors(cls, forward_transitions, forward_transition_probs, forward_transition_indices,
                     backward_transitions, backward_transition_probs, backward_transition_indices,
                     final_probs, initial_probs, leaky_probs=None, start_state=0, log_domain=False):
        """Build from already-laid-out tensors (what `fst_to_tensor` returns)."""
        g = cls.__new__(cls)
        g.num_states = int(final_probs.numel())
        g.log_domain = bool(log_domain)
        g.forward_transitions = forward_transitions
        g.forward_transition_probs = forward_transition_probs
        g.forward_transition_indices = forward_transition_indices
        g.backward_transitions = backward_transitions
        g.backward_transition_probs = backward_transition_probs
        g.backward_transition_indices = backward_transition_indices
        g.final_probs = final_probs
        g.num_transitions = forward_transitions.size(0)
        g.is_empty = (g.num_transitions == 0)
        if g.is_empty:
            raise Exception("An empty graph encountered!")
        g.start_state = int(start_state)
        g.leaky_probs = None if log_domain else leaky_probs
        g.initial_probs = initial_probs
        g._plan_cache = {}
        return g


_TENSORS = ("forward_transitions", "forward_transition_indices", "forward_transition_probs",
            "backward_transitions", "backward_transition_indices", "backward_transition_probs",
            "final_probs", "leaky_probs", "initial_probs", "start_state")

# ---- one buffer per batch (include/pychain_hip.h: batch containers) -------------------------------------------
# field order of pychain_hip_batch_layout, with the shape of a row and the dtype of each field
_PACKED = (("forward_transitions", lambda K, H: (K, 3), torch.int32),
           ("forward_transition_indices", lambda K, H: (H, 2), torch.int32),
           ("forward_transition_probs", lambda K, H: (K,), torch.float32),
           ("backward_transitions", lambda K, H: (K, 3), torch.int32),
           ("backward_transition_indices", lambda K, H: (H, 2), torch.int32),
           ("backward_transition_probs", lambda K, H: (K,), torch.float32),
           ("final_probs", lambda K, H: (H,), torch.float32),
           ("initial_probs", lambda K, H: (H,), torch.float32),
           ("leaky_probs", lambda K, H: (H,), torch.float32),
           ("start_state", lambda K, H: (), torch.int64))


def _native():
    """The C library, or None where it is not built (the containers then collate in Python, as the reference does)."""
    try:
        from . import _lib
        return _lib.lib()
    except (ImportError, OSError):
        return None


def _layout(L, B, K, H, log_domain):
    offs = np.zeros(10, dtype=np.int64)
    rows = np.zeros(10, dtype=np.int64)
    total = L.pychain_hip_batch_layout(B, K, H, int(log_domain), offs.ctypes.data_as(ctypes.c_void_p),
                                       rows.ctypes.data_as(ctypes.c_void_p))
    if total < 0:
        raise ValueError("bad batch sizes B=%d K=%d H=%d" % (B, K, H))
    return int(total), offs, rows


_REC_NAMES = tuple(name for name, _s, _d in _PACKED[:9])


def _may_pin():
    """Pinned host memory only where this process already talks to the GPU: `ChainGraphBatch(list)` is what a DataLoader
    `collate_fn` builds in FORKED workers, where `torch.cuda.is_available()` is still true (the parent's answer is
    cached) but touching the HIP runtime is an error or a hang; the reference collation is pure CPU
    (pychain/graph.py:122-175)."""
    try:
        import torch.utils.data as tud
        return torch.cuda.is_initialized() and tud.get_worker_info() is None
    except Exception:
        return False


def _staging_buffer(nbytes, like=None):
    pin = like.is_pinned() if like is not None else True
    if pin and _may_pin():
        try:
            return torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        except RuntimeError:
            pass                                               # (out of pinnable memory: pageable works, one sync copy)
    return torch.empty(nbytes, dtype=torch.uint8)


def _pack_record(g):
    """(num_transitions, num_states, start_state, addresses of the nine tensors in _PACKED order), or None if a
    tensor is not a contiguous CPU int32 / float32 tensor of the expected size.  Remembered on the graph for as long
    as it keeps the very same tensor objects (a trainer collates the same ChainGraph objects step after step)."""
    d = g.__dict__
    c = d.get("_pack_cache")
    if c is not None and c[1] == (g.num_transitions, g.num_states, g.start_state):
        ts, rec = c[0], c[2]
        for i in range(9):
            t = d[_REC_NAMES[i]]
            # the same tensor OBJECT is not enough: a cache that reached this graph through a copy of its __dict__ (or a
            # tensor whose storage was swapped by set_ / resize_) points at memory that is not this tensor's
            if t is not ts[i] or (rec is not None and (t.data_ptr() if t is not None else 0) != rec[3 + i]):
                break
        else:
            return rec
    rec = _pack_record_uncached(g)
    d["_pack_cache"] = (tuple(d[n] for n in _REC_NAMES), (g.num_transitions, g.num_states, g.start_state), rec)
    return rec


def _pack_record_uncached(g):
    rec = [g.num_transitions, g.num_states, int(g.start_state)]
    k, h = g.num_transitions, g.num_states
    want = ((k * 3, torch.int32), (h * 2, torch.int32), (k, torch.float32), (k * 3, torch.int32), (h * 2, torch.int32),
            (k, torch.float32), (h, torch.float32), (h, torch.float32), (h, torch.float32))
    for (name, _shape, _dt), (n, dt) in zip(_PACKED[:9], want):
        t = getattr(g, name)
        if t is None:
            if name != "leaky_probs":
                return None
            rec.append(0)
            continue
        if t.dtype != dt or t.device.type != "cpu" or not t.is_contiguous() or t.numel() != n:
            return None
        rec.append(t.data_ptr())
    return rec


class ChainGraphBatch(object):
    """B graphs as batched tensors (pychain/graph.py:73-194)."""

    def __init__(self, graphs, batch_size=None, max_num_transitions=None, max_num_states=None):
        self.shared_graph = None     # set when every row is the same ChainGraph
        self._device_cache = {}
        self._staging = None         # the one buffer behind a batch built from a list (initialized_by_list)
        if isinstance(graphs, ChainGraph):
            if not batch_size:
                raise ValueError("batch size should be specified to expand a single graph")
            self.batch_size = batch_size
            self.initialized_by_one(graphs)
        elif isinstance(graphs, list):
            if not max_num_transitions:
                raise ValueError(
                    "max_num_transitions should be specified if given a "
                    "a list of ChainGraph objects to initialize from")
            if not max_num_states:
                raise ValueError(
                    "max_num_states should be specified if given a "
                    "a list of ChainGraph objects to initialize from")
            self.batch_size = len(graphs)
            self.initialized_by_list(graphs, max_num_transitions, max_num_states)
        else:
            raise ValueError(
                "ChainGraphBatch should be either initialized by a "
                "single ChainGraph object or a list of ChainGraph objects "
                "but given {}".format(type(graphs)))

    def __getstate__(self):
        # device copies never travel; a batch built from a list travels as its ONE buffer (the ten tensors are views of it
        # and are re-made on arrival - pickled one by one they would stop sharing storage and the one-copy upload with them)
        d = dict(self.__dict__)
        d["_device_cache"] = {}
        d.pop("_pickle_keepalive", None)
        if self.shared_graph is None and self._packed_consistent():
            for name, _shape, _dt in _PACKED:
                d.pop(name, None)
            # a pageable copy travels (torch's pickler moves a tensor's storage into shared memory IN PLACE, which would
            # un-pin this batch's buffer and re-stage it); the receiver decides about pinning.  The copy has to outlive the
            # pickling - the receiver maps its shared-memory file later - so it stays with the batch (cost: one more host copy
            # of the batch's buffer, a few hundred KB at C3, until the next pickle replaces it or the batch dies).
            d["_staging"] = self.__dict__["_pickle_keepalive"] = self._staging.clone()
            d["_repack"] = True
        return d

    def __setstate__(self, d):
        repack = d.pop("_repack", False)
        self.__dict__.update(d)
        self._device_cache = {}
        if repack:
            st = self._staging
            if _may_pin():
                try:
                    st = st.pin_memory()
                except RuntimeError:
                    pass
            B, K, H = self._shape
            self._install(st, self._offs, self._rows, B, K, H)

    def initialized_by_one(self, graph):
        # Same shapes/values as the reference's .repeat(B, ...) but zero-stride
        # views: no B-fold host copy per training step (graph.py:101-120).
        B = self.batch_size
        self.log_domain = graph.log_domain
        self.num_states = graph.num_states
        self.forward_transitions = graph.forward_transitions.unsqueeze(0).expand(B, -1, -1)
        self.forward_transition_indices = graph.forward_transition_indices.unsqueeze(0).expand(B, -1, -1)
        self.forward_transition_probs = graph.forward_transition_probs.unsqueeze(0).expand(B, -1)
        self.backward_transitions = graph.backward_transitions.unsqueeze(0).expand(B, -1, -1)
        self.backward_transition_indices = graph.backward_transition_indices.unsqueeze(0).expand(B, -1, -1)
        self.backward_transition_probs = graph.backward_transition_probs.unsqueeze(0).expand(B, -1)
        self.final_probs = graph.final_probs.unsqueeze(0).expand(B, -1)
        self.leaky_probs = (graph.leaky_probs.unsqueeze(0).expand(B, -1)
                            if not self.log_domain else None)
        self.initial_probs = graph.initial_probs.unsqueeze(0).expand(B, -1)
        self.start_state = graph.start_state * torch.ones(B, dtype=torch.long)
        self.shared_graph = graph

    def _install(self, staging, offs, rows, B, K, H):
        """The batch tensors as views of the ONE buffer `staging` (uint8, pinned where a GPU is present)."""
        self._staging, self._shape = staging, (B, K, H)
        self._offs, self._rows = offs, rows
        for i, (name, shape, dt) in enumerate(_PACKED):
            if rows[i] == 0:
                setattr(self, name, None)
                continue
            flat = staging[int(offs[i]):int(offs[i]) + int(rows[i]) * B].view(dt)
            setattr(self, name, flat.view((B,) + shape(K, H)))

    def _packed_consistent(self):
        """True while every batch tensor still IS its view of the staging buffer (nobody replaced an attribute)."""
        st = getattr(self, "_staging", None)
        if st is None:
            return False
        base = st.data_ptr()
        for i, (name, _shape, _dt) in enumerate(_PACKED):
            t = getattr(self, name)
            if (t is None) != (self._rows[i] == 0):
                return False
            if t is not None and t.data_ptr() != base + int(self._offs[i]):
                return False
        return True

    def initialized_by_list(self, graphs, max_num_transitions, max_num_states):
        # One native pack into one (pinned) buffer instead of ~9 small tensor copies per utterance in Python
        # (graph.py:122-175: 3.3 ms for a 64-utterance batch, on the thread that launches the loss, every step)
        self._staging = None
        L = _native()
        recs = [_pack_record(g) for g in graphs] if L is not None else [None]
        if L is not None and all(r is not None for r in recs) and all(g.log_domain == graphs[0].log_domain for g in graphs):
            B, K, H = self.batch_size, int(max_num_transitions), int(max_num_states)
            self.log_domain = graphs[0].log_domain
            self.num_states, self.num_transitions = H, K
            total, offs, rows = _layout(L, B, K, H, self.log_domain)
            staging = _staging_buffer(total)
            rec = np.array(recs, dtype=np.uint64)
            from . import _lib
            _lib.check(L.pychain_hip_batch_pack(B, K, H, int(self.log_domain), rec.ctypes.data_as(ctypes.c_void_p),
                                                staging.data_ptr(), total), "pychain_hip_batch_pack")
            self._install(staging, offs, rows, B, K, H)
            return
        ttype = graphs[0].forward_transitions.dtype
        ptype = graphs[0].forward_transition_probs.dtype
        B, K, H = self.batch_size, max_num_transitions, max_num_states
        self.log_domain = graphs[0].log_domain
        self.num_states = H
        self.num_transitions = K
        self.forward_transitions = torch.zeros([B, K, 3], dtype=ttype)
        self.forward_transition_indices = torch.zeros([B, H, 2], dtype=ttype)
        self.forward_transition_probs = torch.zeros([B, K], dtype=ptype)
        self.backward_transitions = torch.zeros([B, K, 3], dtype=ttype)
        self.backward_transition_indices = torch.zeros([B, H, 2], dtype=ttype)
        self.backward_transition_probs = torch.zeros([B, K], dtype=ptype)
        if self.log_domain:
            self.leaky_probs = None
            pad = float("-inf")   # padded states are unreachable (graph.py:140-145)
        else:
            self.leaky_probs = torch.zeros([B, H], dtype=ptype)
            pad = 0.0
        self.initial_probs = torch.full([B, H], pad, dtype=ptype)
        self.final_probs = torch.full([B, H], pad, dtype=ptype)
        self.start_state = torch.zeros([B], dtype=torch.long)
        for i, g in enumerate(graphs):
            k, h = g.num_transitions, g.num_states
            self.forward_transitions[i, :k].copy_(g.forward_transitions)
            self.forward_transition_indices[i, :h].copy_(g.forward_transition_indices)
            self.forward_transition_probs[i, :k].copy_(g.forward_transition_probs)
            self.backward_transitions[i, :k].copy_(g.backward_transitions)
            self.backward_transition_indices[i, :h].copy_(g.backward_transition_indices)
            self.backward_transition_probs[i, :k].copy_(g.backward_transition_probs)
            if self.leaky_probs is not None:
                self.leaky_probs[i, :h].copy_(g.leaky_probs)
            self.initial_probs[i, :h].copy_(g.initial_probs)
            self.final_probs[i, :h].copy_(g.final_probs)
            self.start_state[i] = g.start_state

    def reorder(self, new_order):
        """Permute (or select from) the batch (pychain/graph.py:177-194).  A batch built from a list is re-gathered
        natively in its one buffer, and a copy already staged on a device is re-gathered THERE by one launch instead of
        being dropped and uploaded again."""
        if self.shared_graph is None and self._packed_consistent():
            L = _native()
            from . import _lib
            order = torch.as_tensor(new_order, dtype=torch.int64).cpu().contiguous()
            B_in, K, H = self._shape
            B_out = int(order.numel())
            total, offs, rows = _layout(L, B_out, K, H, self.log_domain)
            old, old_cache = self._staging, self._device_cache
            staging = _staging_buffer(total, like=old)
            _lib.check(L.pychain_hip_batch_reorder(B_in, B_out, K, H, int(self.log_domain), old.data_ptr(), staging.data_ptr(),
                                                   order.data_ptr()), "pychain_hip_batch_reorder")
            old_key = self._device_key_packed()
            self._install(staging, offs, rows, B_out, K, H)
            self.batch_size = B_out
            self._device_cache = {}
            for key, hit in old_cache.items():
                if key[1:] != old_key or "_buffer" not in hit:
                    continue                               # stale (an in-place edit since it was staged): upload afresh
                dbuf = hit["_buffer"]
                with torch.cuda.device(dbuf.device):
                    out = torch.empty(total, dtype=torch.uint8, device=dbuf.device)
                    od = order.to(dbuf.device, non_blocking=True)
                    _lib.check(L.pychain_hip_batch_reorder_dev(B_in, B_out, K, H, int(self.log_domain), dbuf.data_ptr(),
                                                               out.data_ptr(), od.data_ptr(),
                                                               torch.cuda.current_stream(dbuf.device).cuda_stream),
                               "pychain_hip_batch_reorder_dev")
                self._device_cache[(key[0],) + self._device_key_packed()] = self._device_views(out)
            return
        for name in _TENSORS:
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, t.index_select(0, new_order))
        self._staging = None
        self._device_cache = {}
        # every row of a shared batch is the same graph: still shared

    # ---- device staging for the HIP path (not in the reference API) -------
    def _device_key_packed(self):
        return (self._staging.data_ptr(), self._staging._version)

    def _device_views(self, dbuf):
        B, K, H = self._shape
        hit = {"_buffer": dbuf}
        for i, (name, shape, dt) in enumerate(_PACKED[:-1]):
            if self._rows[i]:
                flat = dbuf[int(self._offs[i]):int(self._offs[i]) + int(self._rows[i]) * B].view(dt)
                hit[name] = flat.view((B,) + shape(K, H))
        return hit

    def device_tensors(self, device):
        """Contiguous device copies of the per-sequence tensors, cached until
        `reorder`.  Replaces the per-call `.cuda()` of chain-computation.cc:77-89."""
        if self.shared_graph is None and self._packed_consistent():
            # ONE copy of the one (pinned) buffer; an in-place edit of any tensor bumps the buffer's version
            key = (str(device),) + self._device_key_packed()
            hit = self._device_cache.get(key)
            if hit is None:
                hit = self._device_views(self._staging.to(device, non_blocking=True))
                self._device_cache = {key: hit}
            return hit
        # (data_ptr, _version) of every tensor: an in-place edit or a replaced attribute re-stages
        key = (str(device),) + tuple((getattr(self, n).data_ptr(), getattr(self, n)._version)
                                     for n in _TENSORS[:-1] if getattr(self, n) is not None)
        hit = self._device_cache.get(key)
        if hit is None:
            hit = {}
            for name in _TENSORS[:-1]:
                t = getattr(self, name)
                if t is not None:
                    if self.shared_graph is not None:
                        t = t[:1]            # every row is the same graph: ship one (stride 0)
                    hit[name] = t.contiguous().to(device, non_blocking=True)
            self._device_cache = {key: hit}
        return hit
