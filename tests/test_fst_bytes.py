"""Graph ingestion pinned against bytes this repo did not write:
  * OpenFST binary files assembled by hand from the library's documented VectorFst<StdArc> layout
    (tests/golden/make_fst_bytes.py), read by the native reader (csrc/fst.cpp) and by its Python twin;
  * the reference's real ChainGraph.__init__ (pychain/graph.py:25-70), run for every mode combination
    (tests/golden/make_golden.py: gen_chaingraph_init), against pychain_amd.ChainGraph."""
import json
import math
import os

import numpy as np
import pytest
import torch

from pychain_amd import ChainGraph
from pychain_amd.simplefst import StdVectorFst

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "fst_bytes_expected.json")) as f:
    EXPECTED = json.load(f)


def _check_fst(fst, exp):
    assert fst.num_states() == len(exp["final"])
    assert StdVectorFst.start_state(fst) == exp["start"]
    ft, fp, fi, bt, bp, bi, fin = StdVectorFst.fst_to_tensor(fst, log_domain=True)
    K = sum(len(a) for a in exp["arcs"])
    assert ft.shape == (K, 3) and bt.shape == (K, 3)
    k = 0
    for s, arcs in enumerate(exp["arcs"]):
        assert fi[s].tolist() == [k, k + len(arcs)]                        # fstext.cc:30-58
        for (il, _ol, w, ns) in arcs:
            assert ft[k].tolist() == [s, ns, il - 1]                        # pdf = ilabel - 1, fstext.cc:41
            assert float(fp[k]) == -float(np.float32(w))                    # log-prob = -weight, :43
            k += 1
    for s, fw in enumerate(exp["final"]):
        assert float(fin[s]) == (-math.inf if fw is None else -float(np.float32(fw)))   # :37
    # in-arcs: by destination, ascending source state, then arc order (:36-46, :63-76)
    k = 0
    for d in range(len(exp["arcs"])):
        inn = [(s, ns, il - 1, -float(np.float32(w))) for s, arcs in enumerate(exp["arcs"]) for (il, _ol, w, ns) in arcs if ns == d]
        assert bi[d].tolist() == [k, k + len(inn)]
        for (s, ns, pdf, lp) in inn:
            assert bt[k].tolist() == [s, ns, pdf] and float(bp[k]) == lp
            k += 1


@pytest.mark.parametrize("name", ["fst_a_plain.fst", "fst_a_flags.fst"])
@pytest.mark.parametrize("reader", ["native", "python"])
def test_hand_assembled_fst_file(name, reader):
    path = os.path.join(GOLDEN, name)
    fst = StdVectorFst.read(path) if reader == "native" else StdVectorFst._py_read(path)
    _check_fst(fst, EXPECTED[name])


@pytest.mark.parametrize("reader", ["native", "python"])
def test_two_fsts_inside_a_kaldi_ark(reader):
    exp = EXPECTED["fst_two.ark"]
    path = os.path.join(GOLDEN, "fst_two.ark")
    for key in ("utt_a", "utt_b"):
        off = exp["offsets"][key]
        fst = StdVectorFst.read_ark(path, off) if reader == "native" else StdVectorFst._py_read(path, off)
        _check_fst(fst, exp[key])
    with pytest.raises(IOError):                      # an offset that is not an FST header
        StdVectorFst.read_ark(path, 1)


def test_embedded_symbol_table_is_rejected():
    with pytest.raises(IOError):
        StdVectorFst.read(os.path.join(GOLDEN, "fst_with_symbols.fst"))
    with pytest.raises(IOError):
        StdVectorFst._py_read(os.path.join(GOLDEN, "fst_with_symbols.fst"))


def test_written_file_has_the_documented_layout(tmp_path):
    """The writer's bytes, parsed here with struct by the documented layout (not by the repo's readers)."""
    import struct
    z = EXPECTED["fst_a_plain.fst"]
    src = StdVectorFst.read(os.path.join(GOLDEN, "fst_a_plain.fst"))
    out = os.path.join(tmp_path, "w.fst")
    src.write(out)
    b = open(out, "rb").read()
    assert struct.unpack_from("<i", b, 0)[0] == 2125659606
    assert b[4:8] == struct.pack("<i", 6) and b[8:14] == b"vector"
    assert b[14:18] == struct.pack("<i", 8) and b[18:26] == b"standard"
    version, flags = struct.unpack_from("<ii", b, 26)
    _props, start, nstates, narcs = struct.unpack_from("<Qqqq", b, 34)
    assert version == 2 and flags & 3 == 0 and start == z["start"] and nstates == len(z["final"])
    assert narcs == sum(len(a) for a in z["arcs"])
    pos = 66
    for s in range(nstates):
        fw, na = struct.unpack_from("<fq", b, pos)
        pos += 12
        assert na == len(z["arcs"][s]) and (math.isinf(fw) if z["final"][s] is None else fw == np.float32(z["final"][s]))
        for a in z["arcs"][s]:
            il, ol, w, ns = struct.unpack_from("<iifi", b, pos)
            assert [il, ol, ns] == [a[0], a[1], a[3]] and w == np.float32(a[2])
            pos += 16
    assert pos == len(b)


# ---------------------------------------------------------------- the reference's ChainGraph.__init__
def _fst_from_golden(z):
    arcs = [(int(a[0]), int(a[1]), int(a[2]), float(a[3])) for a in z["arcs"]]
    finals = {int(s): float(w) for s, w in zip(z["final_states"], z["final_weights"])}
    return StdVectorFst.from_arcs(int(z["num_states"]), int(z["start"]), arcs, finals)


@pytest.mark.parametrize("log_domain", [False, True])
@pytest.mark.parametrize("initial_mode", ["fst", "leaky"])
@pytest.mark.parametrize("final_mode", ["fst", "ones"])
def test_chaingraph_constructor_matches_the_reference(golden, log_domain, initial_mode, final_mode):
    z = golden("a4_chaingraph_init")
    tag = "%s_%s_%d__" % (initial_mode, final_mode, int(log_domain))
    fst = _fst_from_golden(z)
    if tag + "raises" in z.files:
        with pytest.raises(AssertionError, match=str(z[tag + "raises"])):
            ChainGraph(fst, initial_mode=initial_mode, final_mode=final_mode, log_domain=log_domain)
        return
    g = ChainGraph(fst, initial_mode=initial_mode, final_mode=final_mode, log_domain=log_domain)
    assert g.num_states == int(z[tag + "num_states"]) and g.num_transitions == int(z[tag + "num_transitions"])
    assert g.is_empty == bool(z[tag + "is_empty"]) and g.start_state == int(z[tag + "start_state"])
    assert g.log_domain == bool(z[tag + "log_domain"]) and (g.leaky_probs is None) == bool(z[tag + "leaky_is_none"])
    for f in ("forward_transitions", "forward_transition_probs", "forward_transition_indices",
              "backward_transitions", "backward_transition_probs", "backward_transition_indices",
              "final_probs", "initial_probs", "leaky_probs"):
        if tag + f not in z.files:
            assert getattr(g, f) is None
            continue
        ref = z[tag + f]
        got = getattr(g, f)
        assert got.dtype == torch.from_numpy(ref).dtype and tuple(got.shape) == ref.shape, f
        assert np.array_equal(got.numpy(), ref), f


def test_empty_graph_raises_like_the_reference(golden):
    z = golden("a4_chaingraph_init")
    with pytest.raises(Exception, match=str(z["empty_raises"])):
        ChainGraph(StdVectorFst.from_arcs(2, 0, [], {1: 0.0}))
