"""Host-side API mirror: containers, errors, alias package,
C-ABI exports.  CPU only."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import GRAPH_FIELDS, graph_from_npz
from pychain_amd import ChainGraph, ChainGraphBatch, _lib, synthetic as syn
from pychain_amd.simplefst import StdVectorFst

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cmp_batch(gb, z, prefix):
    for f in GRAPH_FIELDS + ["start_state"]:
        if prefix + f in z.files:
            got = getattr(gb, f)
            assert got is not None, f
            ref = z[prefix + f]
            assert tuple(got.shape) == ref.shape and str(got.dtype).replace("torch.", "") == str(ref.dtype), f
            assert np.array_equal(got.numpy(), ref, equal_nan=True), f
        else:
            assert getattr(gb, f) is None, f
    assert gb.num_states == int(z[prefix + "num_states"])
    assert gb.batch_size == int(z[prefix + "batch_size"])
    assert gb.log_domain == bool(z[prefix + "log_domain"])


def test_containers_match_reference_snapshots(golden):
    """ChainGraphBatch by-one / by-list / reorder / prob-domain list == the reference classes."""
    z = golden("a5_containers")
    den = graph_from_npz(z, "den_")
    _cmp_batch(ChainGraphBatch(den, 3), z, "byone_")
    graphs = [graph_from_npz(z, "g%d_" % i) for i in range(3)]
    mk = max(g.num_transitions for g in graphs) + 2
    mh = max(g.num_states for g in graphs) + 1
    gl = ChainGraphBatch(graphs, max_num_transitions=mk, max_num_states=mh)
    _cmp_batch(gl, z, "bylist_")
    assert gl.num_transitions == int(z["bylist_num_transitions"])
    gl.reorder(torch.tensor([2, 0, 1]))
    _cmp_batch(gl, z, "reordered_")
    pg = [graph_from_npz(z, "pg%d_" % i) for i in range(2)]
    _cmp_batch(ChainGraphBatch(pg, max_num_transitions=9, max_num_states=5), z, "problist_")


def test_container_errors():
    g = syn.make_den_graph(6, 14, 9, seed=11)
    with pytest.raises(ValueError, match="batch size should be specified"):
        ChainGraphBatch(g)
    with pytest.raises(ValueError, match="max_num_transitions"):
        ChainGraphBatch([g])
    with pytest.raises(ValueError, match="max_num_states"):
        ChainGraphBatch([g], max_num_transitions=20)
    with pytest.raises(ValueError, match="should be either initialized"):
        ChainGraphBatch("nope", 2)
    with pytest.raises(Exception, match="empty graph"):
        f = StdVectorFst(); f.add_state(); f.set_start(0)
        ChainGraph(f)
    with pytest.raises(AssertionError):
        ChainGraph(syn.make_num_fst(4, 9, 1), initial_mode="leaky", log_domain=True)


def test_chain_graph_modes():
    fst = syn.make_den_fst(6, 14, 9, seed=11)
    a = ChainGraph(fst, initial_mode="leaky", final_mode="ones")
    assert torch.equal(a.initial_probs, a.leaky_probs) and bool((a.final_probs == 1).all())
    assert abs(float(a.leaky_probs.sum()) - 1.0) < 1e-6 and bool((a.leaky_probs >= 0).all())
    b = ChainGraph(fst)   # fst / fst
    assert float(b.initial_probs[b.start_state]) == 1.0 and float(b.initial_probs.sum()) == 1.0
    c = ChainGraph(syn.make_num_fst(5, 9, 3), log_domain=True, final_mode="ones")
    assert c.leaky_probs is None and float(c.initial_probs[0]) == 0.0 and bool((c.final_probs == 0).all())
    assert bool(torch.isinf(c.initial_probs[1:]).all())
    # layout rules of fstext.cc:36-76: out-arcs by source, in-arcs by destination then ascending source
    assert bool((a.forward_transitions[:-1, 0] <= a.forward_transitions[1:, 0]).all())
    assert bool((a.backward_transitions[:-1, 1] <= a.backward_transitions[1:, 1]).all())
    for h in range(a.num_states):
        lo, hi = a.backward_transition_indices[h].tolist()
        src = a.backward_transitions[lo:hi, 0]
        assert bool((src[:-1] <= src[1:]).all()) and bool((a.backward_transitions[lo:hi, 1] == h).all())


def test_native_ingestion_matches_python_restatement():
    """C++ FstToTensor / SetLeakyProbs (csrc/fst.cpp) == the independent pure-Python restatement."""
    for fst, logd in ((syn.make_den_fst(50, 400, 30, seed=2), False), (syn.make_num_fst(9, 30, 4), True)):
        for a, b in zip(StdVectorFst.fst_to_tensor(fst, logd), StdVectorFst._py_fst_to_tensor(fst, logd)):
            assert a.dtype == b.dtype and a.shape == b.shape
            assert torch.equal(a, b) if a.dtype == torch.int32 else torch.allclose(a, b, rtol=1e-6, atol=0)
    fst = syn.make_den_fst(50, 400, 30, seed=2)
    assert torch.allclose(StdVectorFst.set_leaky_probs(fst), StdVectorFst._py_set_leaky_probs(fst), rtol=1e-5, atol=1e-9)


def test_fst_binary_roundtrip(tmp_path):
    fst = syn.make_den_fst(7, 20, 11, seed=5)
    p = str(tmp_path / "g.fst")
    assert fst.write(p)
    back = StdVectorFst.read(p)
    for x, y in zip(StdVectorFst.fst_to_tensor(fst), StdVectorFst.fst_to_tensor(back)):
        assert torch.equal(x, y)
    with open(str(tmp_path / "ark"), "wb") as f:   # Kaldi-ark style: FST at a byte offset
        f.write(b"utt1 " + fst._to_bytes())
    assert StdVectorFst.read_ark(str(tmp_path / "ark"), 5).num_states() == 7
    # the native reader and the pure-Python reader agree on the bytes the native writer produced
    py = StdVectorFst._py_read(p)
    for x, y in zip(StdVectorFst.fst_to_tensor(back), StdVectorFst._py_fst_to_tensor(py)):
        assert torch.allclose(x.float(), y.float())
    with pytest.raises(IOError):
        StdVectorFst.read(str(tmp_path / "missing.fst"))


def test_alias_package_and_exports():
    import pychain
    import pychain.graph
    import pychain.loss
    assert pychain.ChainLoss is pychain.loss.ChainLoss and pychain.ChainGraph is pychain.graph.ChainGraph
    import pychain_C
    for name in ("forward_backward", "forward_backward_log_domain", "set_verbose_level"):
        assert callable(getattr(pychain_C, name))
    # every symbol the header declares is exported by the library (no compute calls here)
    hdr = open(os.path.join(REPO, "include", "pychain_hip.h")).read()
    declared = set(re.findall(r"\b(pychain_hip_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pychain_hip_abi_version() == _lib.ABI_VERSION
    _lib.lib().pychain_hip_set_verbose_level(2)
    assert _lib.lib().pychain_hip_get_verbose_level() == 2
    _lib.lib().pychain_hip_set_verbose_level(0)
    # the two workspace sizes of the denominator (include/pychain_hip.h): the full one holds the [B,T,D] buffer of the rows
    # exp'd ahead of the recursions on top of the other (256-byte granules), and both reject empty shapes
    L = _lib.lib()
    for (B, T, H, D) in ((64, 1500, 3000, 3456), (32, 2000, 3000, 8408), (3, 7, 5, 9)):
        full, least = L.pychain_hip_den_workspace_bytes(B, T, H, D), L.pychain_hip_den_workspace_min_bytes(B, T, H, D)
        assert 0 < least < full and full - least == (4 * B * T * D + 255) // 256 * 256
        assert least >= 4 * B * (2 * T + 1) * ((H + 63) // 64 * 64)          # the stored alpha' / beta rows
    assert L.pychain_hip_den_workspace_bytes(0, 7, 5, 9) == 0 and L.pychain_hip_den_workspace_min_bytes(3, 7, 5, 0) == 0


def test_no_cpu_fallback():
    """The HIP entry points take device tensors only and raise on anything else; CPU tensors are served by the library's
    host twins (tests/test_cpu_twins.py) through a door of their own - never one for the other."""
    from pychain_amd import native
    den = syn.make_den_graph(6, 14, 9, seed=11)
    gt = {n: getattr(den, n).unsqueeze(0) for n in ("forward_transitions", "forward_transition_indices", "forward_transition_probs",
                                                    "backward_transitions", "backward_transition_indices", "backward_transition_probs",
                                                    "initial_probs", "final_probs")}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.num_forward_backward(gt, 0, den.num_states, torch.zeros(2, 5, 9), torch.tensor([5, 5]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.den_forward_backward(None, torch.zeros(2, 5, 9), torch.tensor([5, 5]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.chain_loss_forward(None, gt, 0, den.num_states, torch.zeros(2, 5, 9), torch.tensor([5, 5]))
    # the product package never imports, links or opens anything under oracle/
    pat = re.compile(r"^\s*(import|from)\s+(oracle|ref_loader|build_ref)\b|oracle[/\\]|chain_oracle|libchain_oracle|_ref/",
                     re.M)
    for root, _d, files in os.walk(os.path.join(REPO, "pychain_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                assert not pat.search(open(os.path.join(root, fn)).read()), fn


def test_options_are_per_thread_and_restored():
    """A validation thread at verbose level 1 does not change what a training thread's calls see; the context manager puts
    back what was there (also a process-wide default taken from the environment at load time)."""
    import threading
    L_ = _lib.lib()
    L_.pychain_hip_set_option(b"den_segments", b"3")            # process-wide default (as PYCHAIN_DEN_SEGMENTS=3 would)
    try:
        assert _lib.get_option("den_segments") == "3"
        with _lib.option("den_segments", 5):
            assert _lib.get_option("den_segments") == "5"
            seen = []
            t = threading.Thread(target=lambda: seen.append(_lib.get_option("den_segments")))
            t.start(); t.join()
            assert seen == ["3"]                                 # the other thread sees the default, not this override
            with _lib.option("den_segments", ""):
                assert _lib.get_option("den_segments") is None   # "" = unset for this thread
            assert _lib.get_option("den_segments") == "5"
        assert _lib.get_option("den_segments") == "3"            # not lost
    finally:
        L_.pychain_hip_set_option(b"den_segments", None)
    assert _lib.get_option("den_segments") is None
    from pychain_amd import native
    native.set_verbose_level(2)
    assert L_.pychain_hip_get_verbose_level() == 2 and _lib.get_option("verbose") == "2"
    native.set_verbose_level(0)


def _same_batch(a, b):
    from pychain_amd.graph import _TENSORS
    for n in _TENSORS:
        x, y = getattr(a, n), getattr(b, n)
        assert (x is None) == (y is None), n
        if x is not None:
            assert x.dtype == y.dtype and x.shape == y.shape, n
            assert np.array_equal(x.numpy(), y.numpy(), equal_nan=True), n      # (-inf == -inf)
    assert a.num_states == b.num_states and a.batch_size == b.batch_size and a.log_domain == b.log_domain


def test_native_batch_pack_matches_the_python_collation(monkeypatch):
    """ChainGraphBatch(list) packs natively into ONE buffer (pychain_hip_batch_pack) and `reorder` re-gathers that buffer
    (pychain_hip_batch_reorder): bit-equal to the statement-by-statement Python collation of pychain/graph.py:122-194 kept
    as the fallback - log domain and probability domain, padding in K and H, a reorder that selects a subset, an attribute
    replaced by hand (the batch then stops trusting its buffer)."""
    import pychain_amd.graph as G
    num = [ChainGraph(syn.make_num_fst(h, 40, seed=20 + h), log_domain=True) for h in (5, 9, 3, 12)]
    den = [syn.make_den_graph(h, 4 * h, 40, seed=h) for h in (6, 11, 8)]
    for graphs in (num, den):
        mk, mh = max(g.num_transitions for g in graphs) + 3, max(g.num_states for g in graphs) + 2
        native = ChainGraphBatch(graphs, max_num_transitions=mk, max_num_states=mh)
        assert native._staging is not None and native._packed_consistent()
        with monkeypatch.context() as m:
            m.setattr(G, "_native", lambda: None)
            python = ChainGraphBatch(graphs, max_num_transitions=mk, max_num_states=mh)
        assert python._staging is None
        _same_batch(native, python)
        for order in (torch.tensor([2, 0, 1]), torch.tensor([1, 1]), torch.tensor([0])):
            a, b = ChainGraphBatch(graphs, max_num_transitions=mk, max_num_states=mh), None
            with monkeypatch.context() as m:
                m.setattr(G, "_native", lambda: None)
                b = ChainGraphBatch(graphs, max_num_transitions=mk, max_num_states=mh)
                b.reorder(order)
            a.reorder(order)
            b.batch_size = int(order.numel())              # (the reference leaves that to the caller)
            assert a._packed_consistent()
            _same_batch(a, b)
        # a replaced attribute: the buffer is no longer the truth
        c = ChainGraphBatch(graphs, max_num_transitions=mk, max_num_states=mh)
        c.final_probs = c.final_probs.clone()
        assert not c._packed_consistent()
        c.reorder(torch.tensor([1, 0, 2]))
        assert torch.equal(c.final_probs[0], native.final_probs[1]) and c._staging is None
    with pytest.raises(_lib.PychainHipError):
        ChainGraphBatch(num, max_num_transitions=2, max_num_states=50)     # a graph larger than the batch allows


def _small_num_graphs():
    return [ChainGraph(syn.make_num_fst(h, 9, 300 + i), log_domain=True) for i, h in enumerate([5, 3, 4])]


def _collate(graphs):
    return ChainGraphBatch(list(graphs), max_num_transitions=max(g.num_transitions for g in graphs),
                           max_num_states=max(g.num_states for g in graphs))


def test_pack_cache_does_not_follow_a_copied_graph():
    """the native collation remembers the host ADDRESSES of a graph's tensors on the graph.  A
    copy.deepcopy / pickle of a collated graph must not carry them along (its tensors live elsewhere; in another
    process the addresses are garbage): an edit of the copy has to show in a batch collated from the copy, also after
    the original is gone."""
    import copy
    import gc
    import pickle
    graphs = _small_num_graphs()
    _collate(graphs)                                           # (fills the caches)
    for clone in (copy.deepcopy, lambda g: pickle.loads(pickle.dumps(g)), copy.copy):
        g2 = clone(graphs[0])
        assert "_pack_cache" not in g2.__dict__ and g2._plan_cache == {}
        if clone is not copy.copy:                             # (a shallow copy shares its tensors: nothing to edit apart)
            g2.forward_transition_probs.fill_(-5.0)
        others = _small_num_graphs()[1:]
        gb = _collate([g2] + others)
        k = g2.num_transitions
        assert torch.equal(gb.forward_transition_probs[0, :k], g2.forward_transition_probs)
        assert torch.equal(gb.forward_transitions[0, :k], g2.forward_transitions)
    # the original dies, its memory is reused, the copy is collated again
    g2 = copy.deepcopy(graphs[0])
    g2.forward_transition_probs.fill_(-7.0)
    del graphs
    gc.collect()
    junk = [torch.full((64,), float(i)) for i in range(256)]     # noqa: F841  (recycles the freed blocks)
    gb = _collate([g2])
    assert torch.equal(gb.forward_transition_probs[0], g2.forward_transition_probs)
    # a tensor whose storage is swapped under the same object (set_) is noticed too
    g3 = _small_num_graphs()[0]
    _collate([g3])
    g3.forward_transition_probs.set_(torch.full_like(g3.forward_transition_probs, -9.0))
    assert torch.equal(_collate([g3]).forward_transition_probs[0], g3.forward_transition_probs)


def test_batch_survives_pickling_as_one_buffer():
    """A batch built from a list travels (DataLoader worker -> trainer) as its ONE buffer and arrives with the ten
    tensors as views of it again: the one-copy upload and the native reorder still apply on the other side."""
    import pickle
    gb = _collate(_small_num_graphs())
    assert gb._packed_consistent()
    gb2 = pickle.loads(pickle.dumps(gb))
    assert gb2._packed_consistent() and gb2._device_cache == {}
    for name in GRAPH_FIELDS + ["start_state"]:
        a, b = getattr(gb, name), getattr(gb2, name)
        assert (a is None and b is None) or torch.equal(a, b), name
    gb2.reorder(torch.tensor([2, 0, 1]))
    gb.reorder(torch.tensor([2, 0, 1]))
    assert gb2._packed_consistent() and torch.equal(gb.forward_transitions, gb2.forward_transitions)


class _UttSet(torch.utils.data.Dataset):
    def __init__(self, graphs):
        self.graphs = graphs

    def __len__(self):
        return 6

    def __getitem__(self, i):
        return self.graphs[i % len(self.graphs)]


def _collate_fn(graphs):
    gb = _collate(graphs)
    assert not gb._staging.is_pinned()                         # a worker never touches the GPU runtime
    return gb


@pytest.mark.parametrize("ctx", ["fork", "spawn"])
def test_collation_in_dataloader_workers(ctx):
    """`ChainGraphBatch(list)` inside a DataLoader collate_fn with num_workers > 0 (the upstream usage
    pattern) - pure CPU in the worker (no pinned allocation there), bit-equal to a collation in this process."""
    graphs = _small_num_graphs()
    want = _collate(graphs)
    dl = torch.utils.data.DataLoader(_UttSet(graphs), batch_size=3, collate_fn=_collate_fn, num_workers=2,
                                     multiprocessing_context=ctx)
    n = 0
    for gb in dl:
        assert gb._packed_consistent()
        for name in GRAPH_FIELDS:
            a, b = getattr(want, name), getattr(gb, name)
            assert (a is None and b is None) or torch.equal(a, b), name
        n += 1
    assert n == 2


def _child_sum(gb, q):
    q.put(float(gb.forward_transition_probs.sum()) if gb._packed_consistent() else None)


def test_batch_as_an_argument_of_a_spawned_process():
    """What tests/helpers.py:oracle_fanout and bench.py's all-cores CPU leg do: a collated batch (and a copy.copy of one)
    handed to `torch.multiprocessing` spawn workers.  The buffer that travels must outlive the pickling."""
    import copy
    import torch.multiprocessing as mp
    gb = _collate(_small_num_graphs())
    ctx = mp.get_context("spawn")
    for obj in (gb, copy.copy(gb)):
        q = ctx.Queue()
        p = ctx.Process(target=_child_sum, args=(obj, q))
        p.start()
        got = q.get(timeout=120)
        p.join()
        assert p.exitcode == 0 and got == float(gb.forward_transition_probs.sum())


def test_policy_queries_answer_without_a_device():
    """What a caller asks before it allocates (slices of a fused call, time segments, the [B,T,D] row buffer) is host
    arithmetic: it answers on a machine without a GPU (256 CUs assumed) and follows the documented rules."""
    L = _lib.lib()
    hint = 32 | (1 << 30)
    assert [L.pychain_hip_chain_loss_slices(0, hint, B) for B in (64, 128, 223, 224, 256, 320, 384, 512)] == [1, 1, 1, 2, 2, 2, 3, 4]
    assert L.pychain_hip_chain_loss_slices(4096, hint, 512) == 1                       # per-sequence plans: one call
    with _lib.option("chain_slices", "0"):
        assert L.pychain_hip_chain_loss_slices(0, hint, 512) == 1
    with _lib.option("chain_slices", "3"):
        assert L.pychain_hip_chain_loss_slices(0, hint, 90) == 3
    # time segments: few sequences only, never the fused step at B = 64, never below two burn-ins
    ts = lambda B, T, fused: L.pychain_hip_den_time_segments(0, hint, 3000, 3456, B, T, fused)
    assert ts(64, 1500, 1) == 1 and ts(16, 1500, 1) == 4 and ts(24, 1500, 1) == 4 and ts(32, 1500, 1) == 2
    assert ts(40, 1500, 1) == 2 and ts(48, 1500, 1) == 2 and ts(56, 1500, 1) == 1      # (a fused call leaves a quarter of the chip alone)
    assert ts(32, 1500, 0) == 4 and ts(64, 1500, 0) == 2 and ts(128, 1500, 0) == 1 and ts(8, 300, 0) == 1
    with _lib.option("den_tseg", "0"):
        assert ts(16, 1500, 1) == 1
