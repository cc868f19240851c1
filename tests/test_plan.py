"""Host-side plan compiler + kernel decomposition, checked on the CPU against the oracle
and the reference-generated goldens (no GPU needed)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle as orc
import plan_emulator as emu
from helpers import graph_from_npz, rel_err
from pychain_amd import _lib, _plan, synthetic as syn
from pychain_amd.graph import ChainGraph, ChainGraphBatch
from pychain_amd.simplefst import StdVectorFst


def _blob(g, D):
    return _plan.build_plan_blob(g.forward_transitions, g.forward_transition_indices, g.forward_transition_probs,
                                 g.backward_transitions, g.backward_transition_indices,
                                 g.backward_transition_probs, g.leaky_probs, g.initial_probs, g.final_probs, D)


def test_plan_structure():
    g = syn.make_den_graph(200, 2000, 1000, seed=3)
    with _lib.option("plan_split", "0"):                           # (every state on one lane: test_states_on_several_lanes has the other form)
        hd = emu.parse(_blob(g, 1000))
    assert hd["H"] == 200 and hd["K"] == 2000 and hd["Hp"] == 256
    for name, nreal in (("alpha", 2000), ("beta", 2000), ("gamma", 2000)):
        t = hd[name]
        assert int((t["p"] != 0).sum()) == nreal                  # every arc exactly once
        assert t["waves"][:, 3].sum() == t["total_slot_rows"]
        assert t["waves"][:, 3].max() == t["max_wave_slot_rows"]
        assert sorted(t["groups"][:, 0].tolist()) == list(range(0, t["ngroups"] * 64, 64))
        # longest-processing-time balance: no wave carries more than ~2 groups above the mean
        assert t["waves"][:, 3].max() <= t["total_slot_rows"] / t["nwaves"] + 2 * t["groups"][:, 1].max()
    assert abs(hd["leaky_a"].sum() - g.leaky_probs.sum().item()) < 1e-6
    assert set(hd["row_pdf"][hd["row_pdf"] >= 0].tolist()) == set(g.forward_transitions[:, 2].tolist())


def test_plan_four_wave_dealing_of_small_graphs_and_checksums():
    """A graph whose recursion tiles fit FOUR waves (C2: 200 states, 2000 arcs) also carries them dealt to four waves
    (den_recursion_lazy_kernel<small>: 256-thread workgroups): same slot-rows and sums as the 16-wave dealing, at most
    40 rows and 4 groups per wave, the launch hint says so (bit 29) and carries THAT dealing's row count; a large graph
    (C3) does not.  A plan damaged on its way through the disk cache - payload OR header - is refused."""
    cfg = syn.CONFIGS["C2"]
    g = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    blob = _blob(g, cfg["D"])
    hd = emu.parse(blob)
    rng = np.random.default_rng(0)
    hint = _plan.plan_info(blob)["slot_rows"]
    assert (hint >> 29) & 1 and (hint >> 30) & 1
    for name in ("alpha", "beta"):
        t16, t4 = hd[name], hd[name + "4"]
        assert t4["nwaves"] == 4 and t16["nwaves"] == 16
        assert t4["total_slot_rows"] == t16["total_slot_rows"] and t4["ngroups"] == t16["ngroups"]
        assert t4["max_wave_slot_rows"] <= 40 and t4["waves"][:, 1].max() <= 4 and t4["max_wave_slot_rows"] <= (hint & 1023)
        U, V = rng.random(hd["Hp"]), rng.random(cfg["D"])
        assert np.array_equal(emu.tile_rows(t16, U, V, hd["Hp"], np.float64), emu.tile_rows(t4, U, V, hd["Hp"], np.float64))
    assert 1 <= hd["rec4_max_wave_groups"] <= 4
    big = syn.make_den_graph(1200, 12000, 2000, seed=4)
    bblob = _blob(big, 2000)
    bh = emu.parse(bblob)
    assert bh["alpha4"]["nwaves"] == 0 and bh["beta4"]["nwaves"] == 0 and not (_plan.plan_info(bblob)["slot_rows"] >> 29) & 1
    assert _plan.plan_info(bblob)["num_states"] == 1200
    info = np.zeros(8, dtype=np.int32)
    for where, what in ((len(bblob) // 2, b"checksum"), (8 * 4 + 2 * 4, b"checksum"), (6 * 4, b"")):     # payload, a tile offset, total_bytes
        bad = bblob.copy()
        bad[where] ^= 0x40
        rc = _lib.lib().pychain_hip_den_plan_info(bad.ctypes.data_as(ctypes.c_void_p), bad.nbytes, info.ctypes.data_as(ctypes.c_void_p))
        assert rc < 0 and what in _lib.lib().pychain_hip_last_error()
    # a header that points outside the blob is refused even with hashes that match it (a foreign writer)
    def fnv(b):
        h = 2166136261
        for x in b:
            h = ((h ^ int(x)) * 16777619) & 0xffffffff
        return h
    bad = bblob.copy()
    w = bad[:emu.HDR_INTS * 4].view(np.int32)
    w[8 + 4] = int(w[6]) - 64                                  # alpha.off_slots close to the end of the blob
    w[39] = 0; w[64] = 0
    hh = fnv(bad[:emu.HDR_INTS * 4])                           # (header hash: over the header with both hash words zero)
    w[64] = np.uint32(fnv(bad[emu.HDR_INTS * 4:int(w[6])])).view(np.int32)
    w[39] = np.uint32(hh).view(np.int32)
    rc = _lib.lib().pychain_hip_den_plan_info(bad.ctypes.data_as(ctypes.c_void_p), bad.nbytes, info.ctypes.data_as(ctypes.c_void_p))
    assert rc < 0 and b"outside the blob" in _lib.lib().pychain_hip_last_error()


def _lds_cycles(t):
    """Modelled LDS cycles per half slot-row of a tile: a wave64 ds_read_b32 is served as two 32-lane
    halves over 32 banks, a half costs as many cycles as its fullest bank (equal addresses broadcast)."""
    tot = 0
    n = 0
    for row in range(t["idx"].shape[0]):
        for half in (slice(0, 32), slice(32, 64)):
            for op in (t["idx"][row, half] & 0xffff, t["idx"][row, half] >> 16):
                addr = np.unique(op)
                tot += int(np.bincount(addr & 31, minlength=32).max())
            n += 1
    return tot / max(n, 1)


def test_plan_bank_conflicts_and_determinism():
    """The plan compiler's row placement + slack + slot order keep the recursion gathers close to
    conflict-free (2.0 cycles per half slot-row for the two gathers; natural order: ~7, slot order
    alone: ~3.8), the spare slot-rows stay inside the register-resident loop, and the same graph
    always compiles to the same bytes."""
    g = syn.make_den_graph(1200, 12000, 2000, seed=4)
    b1, b2 = _blob(g, 2000), _blob(g, 2000)
    assert bytes(b1) == bytes(b2)
    hd = emu.parse(b1)
    for name in ("alpha", "beta"):
        assert _lds_cycles(hd[name]) <= 2.6
        assert hd[name]["max_wave_slot_rows"] in range(1, 41)
    assert _lds_cycles(hd["gamma"]) <= 4.2


@pytest.mark.parametrize("case", ["leaky_ones", "fst_fst", "leaky_fst_coef01", "fst_ones_clamp"])
def test_decomposition_matches_reference(golden, case):
    """alpha/beta with independent normalisers + normalised gamma == reference numbers."""
    z = golden("g3_den_variants")
    p = case + "__"
    g = graph_from_npz(z, p + "den_")
    x = z[p + "x"]
    objf, grad = emu.den_forward_backward(_blob(g, x.shape[2]), x, z[p + "lengths"], float(z[p + "coef"]))
    assert abs(objf.sum() - z[p + "objf"]) <= 2e-6 * abs(z[p + "objf"])
    assert rel_err(grad, z[p + "grad"]) <= 1e-5


def test_decomposition_c1(golden):
    z = golden("g1_c1_den")
    g = graph_from_npz(z, "den_")
    objf, grad = emu.den_forward_backward(_blob(g, 40), z["x"], z["lengths"], 1e-5)
    assert abs(objf.sum() - z["objf"]) <= 2e-6 * abs(z["objf"])
    assert rel_err(grad, z["grad"]) <= 1e-5
    # float32 evaluation of the same scheme stays inside the 1e-4 bar with margin
    objf32, grad32 = emu.den_forward_backward(_blob(g, 40), z["x"], z["lengths"], 1e-5, dtype=np.float32)
    assert abs(objf32.sum() - z["objf"]) <= 1e-5 * abs(z["objf"])
    assert rel_err(grad32, z["grad"]) <= 5e-5


def test_plan_errors():
    g = syn.make_den_graph(20, 60, 40, seed=0)
    L = _lib.lib()
    bad = g.forward_transitions.clone()
    bad[3, 2] = 40    # pdf out of range
    with pytest.raises(_lib.PychainHipError, match="out of range"):
        _plan.build_plan_blob(bad, g.forward_transition_indices, g.forward_transition_probs,
                              g.backward_transitions, g.backward_transition_indices,
                              g.backward_transition_probs, g.leaky_probs, g.initial_probs, g.final_probs, 40)
    assert L.pychain_hip_den_plan_build(None, None, None, None, None, None, None, None, None, 1, 1, 1, None, 0) < 0
    assert b"null" in L.pychain_hip_last_error()


def test_batch_plans_shared_detection():
    g = syn.make_den_graph(20, 60, 40, seed=0)
    gb = ChainGraphBatch(g, 3)
    names = ["forward_transitions", "forward_transition_indices", "forward_transition_probs",
             "backward_transitions", "backward_transition_indices", "backward_transition_probs",
             "leaky_probs", "initial_probs", "final_probs"]
    dp = _plan.batch_plans({n: getattr(gb, n) for n in names}, 40, "cpu")
    assert dp.stride == 0 and np.array_equal(dp.blob.numpy(), _blob(g, 40))
    info = _plan.plan_info(_blob(g, 40))
    assert info["num_states"] == 20 and info["num_transitions"] == 60 and dp.slot_rows == info["slot_rows"] > 0


def test_occupancy_launch_map_covers_every_frame_once():
    """Every live frame of a sequence belongs to exactly one occupancy launch (time segment), and the
    compact grid of that launch visits the 32-frame chunk that holds it exactly once - for ragged
    lengths, every segment count and chunk size (host-side view of den_chunk_of_block /
    den_compact_grid_x / LaunchFrames in csrc/den_kernels.hip)."""
    import ctypes
    import random
    from pychain_amd import _lib
    L_ = _lib.lib()
    rng = random.Random(3)
    out = (ctypes.c_int32 * 4096)()
    for _ in range(300):
        T = rng.randint(1, 700)
        fpb = rng.choice([2, 8, 16, 32])
        nseg = rng.randint(1, 6)
        half = (T + 1) // 2
        ends = sorted(min(T, (max(half, rng.randint(0, T)) + 31) // 32 * 32) for _ in range(nseg - 1)) + [T]
        bounds = (ctypes.c_int32 * 16)(*ends)
        L = rng.choice([1, T, rng.randint(1, T)])
        owner = {}
        for seg in range(nseg):
            frames = [t for t in range(L)
                      if L_.pychain_hip_debug_launch_map(T, L, t, fpb, nseg, bounds, seg, None, 0) == 1]
            for t in frames:
                assert t not in owner, (T, L, ends, t)
                owner[t] = seg
            rc = L_.pychain_hip_debug_launch_map(T, L, 0, fpb, nseg, bounds, seg, out, 4096)
            assert rc >= 0
            chunks = [out[1 + k] for k in range(out[0]) if out[1 + k] >= 0]
            assert len(chunks) == len(set(chunks)), (T, L, ends, seg, chunks)
            assert {t // fpb for t in frames} <= set(chunks), (T, L, ends, seg)
            assert out[0] <= (T + fpb - 1) // fpb
        assert sorted(owner) == list(range(L)), (T, L, ends)
    # the unsegmented launch: one workgroup per chunk of [0, T), every frame belongs to it
    assert L_.pychain_hip_debug_launch_map(100, 70, 69, 32, 0, None, 0, out, 4096) == 1
    assert out[0] == 4 and [out[1 + k] for k in range(4)] == [0, 1, 2, 3]
    assert L_.pychain_hip_debug_launch_map(100, 70, 69, 32, 3, None, 0, out, 4096) < 0      # bad arguments


def test_plan_disk_cache_round_trip(tmp_path, monkeypatch):
    """A compiled plan is stored under a hash of the graph tensors, the pdf count and the compiler knobs
    and read back byte-identical; a changed tensor or knob is a different entry; a damaged file is ignored."""
    import os
    from pychain_amd import _plan, synthetic as syn
    monkeypatch.setenv("PYCHAIN_PLAN_CACHE_DIR", str(tmp_path))
    den = syn.make_den_graph(300, 2500, 512, seed=5)
    args = [getattr(den, n) for n in _plan._NAMES]
    a = _plan.build_plan_blob(*args, 512)
    files = os.listdir(tmp_path)
    assert len(files) == 1 and files[0].endswith(".plan")
    b = _plan.build_plan_blob(*args, 512)
    assert (a == b).all()
    c = _plan.build_plan_blob(*args, 512, use_cache=False)
    assert (a == c).all()
    # another pdf count, another final vector: new entries
    _plan.build_plan_blob(*args, 640)
    den.final_probs.mul_(0.5)
    _plan.build_plan_blob(*[getattr(den, n) for n in _plan._NAMES], 512)
    assert len(os.listdir(tmp_path)) == 3
    # a truncated entry is recompiled, not trusted
    path = os.path.join(tmp_path, files[0])
    with open(path, "r+b") as f:
        f.truncate(100)
    den.final_probs.mul_(2.0)
    d = _plan.build_plan_blob(*[getattr(den, n) for n in _plan._NAMES], 512)
    assert (a == d).all() and os.path.getsize(path) == a.nbytes
    # ... and so is a file of the right size and header whose payload was damaged (the kernels would follow its
    # offsets and packed LDS addresses unchecked)
    with open(path, "r+b") as f:
        f.seek(a.nbytes // 2)
        byte = f.read(1)
        f.seek(a.nbytes // 2)
        f.write(bytes([byte[0] ^ 0x01]))
    e = _plan.build_plan_blob(*[getattr(den, n) for n in _plan._NAMES], 512)
    assert (a == e).all()
    with open(path, "rb") as f:
        assert f.read() == a.tobytes()                  # rewritten with the good bytes
    # the key carries the library build: a rebuilt plan compiler does not get its predecessor's plans
    monkeypatch.setattr(_plan, "_BUILD_ID", "another build")
    _plan.build_plan_blob(*[getattr(den, n) for n in _plan._NAMES], 512)
    assert len(os.listdir(tmp_path)) == 4


def test_graph_plan_follows_in_place_edits():
    """The plan cached on a ChainGraph is rebuilt when one of its tensors is modified in place."""
    import torch
    from pychain_amd import _plan, synthetic as syn
    den = syn.make_den_graph(40, 200, 64, seed=2)
    p1 = _plan.graph_plan(den, 64, torch.device("cpu"))
    assert _plan.graph_plan(den, 64, torch.device("cpu")) is p1
    den.final_probs.fill_(0.25)
    p2 = _plan.graph_plan(den, 64, torch.device("cpu"))
    assert p2 is not p1 and not torch.equal(p1.blob, p2.blob)


def test_general_plan_format(monkeypatch):
    """Graphs beyond the tile kernels (here: forced, and a genuine one with more than 65 535 pdfs) compile to the general
    format: the reference layout + the arcs grouped by pdf-id, checksummed, launch hint PYCHAIN_HIP_HINT_GENERAL."""
    import struct
    g = syn.make_den_graph(50, 400, 70000, seed=9)                  # D > 65535: no packed 16-bit addresses
    blob = _blob(g, 70000)
    info = _plan.plan_info(blob)
    assert info["slot_rows"] == _plan.HINT_GENERAL and (info["num_states"], info["num_transitions"], info["num_pdfs"]) == (50, 400, 70000)
    magic, version, H, K, D, Hp, total = struct.unpack_from("6iq", blob.tobytes(), 0)
    assert magic == 0x47504843 and total == blob.nbytes and Hp == 64
    offs = struct.unpack_from("12q", blob.tobytes(), 32)
    g_idx = np.frombuffer(blob.tobytes(), dtype=np.int32, count=D + 1, offset=offs[6])
    g_arc = np.frombuffer(blob.tobytes(), dtype=np.int32, count=2 * K, offset=offs[7]).reshape(K, 2)
    ft = g.forward_transitions.numpy()
    assert g_idx[0] == 0 and g_idx[-1] == K and np.all(np.diff(g_idx) >= 0)
    for n in np.unique(ft[:, 2])[:50]:
        want = ft[ft[:, 2] == n][:, :2]                              # (state, arc) order of the reference
        assert np.array_equal(g_arc[g_idx[n]:g_idx[n + 1]], want)
    bad = blob.copy()
    bad[-5] ^= 1
    i8 = np.zeros(8, dtype=np.int32)
    assert _lib.lib().pychain_hip_den_plan_info(bad.ctypes.data_as(ctypes.c_void_p), bad.nbytes, i8.ctypes.data_as(ctypes.c_void_p)) < 0
    # a graph the tile kernels take, forced: same format
    monkeypatch.setenv("PYCHAIN_PLAN_GENERAL", "1")
    small = syn.make_den_graph(20, 60, 40, seed=0)
    assert _plan.plan_info(_blob(small, 40))["slot_rows"] == _plan.HINT_GENERAL
    monkeypatch.delenv("PYCHAIN_PLAN_GENERAL")
    assert _plan.plan_info(_blob(small, 40))["slot_rows"] != _plan.HINT_GENERAL
    # state vector + nnet-output row beyond the LDS of one CU: general too
    assert _plan.plan_info(_blob(syn.make_den_graph(300, 1200, 40000, seed=1), 40000))["slot_rows"] == _plan.HINT_GENERAL


def test_rings_of_the_streamed_occupancy_pass():
    """The streamed occupancy launch hands out, per sequence, rings of frames by the step count that makes them computable
    (den_kernels.h: stream_ring): the rings tile [0, T) without gaps, 16 steps wide, 4 over the last 64 steps and
    ONE frame pair over the last 8; every ring end is a step count at which the recursions report (else the ring would wait
    for a later report) and every report falls on an even step count (the recursion loop reports every second step); a
    sequence of length L <= T gets every one of its frames from exactly one (ring, side) item (den_kernels.hip: stream_take)."""
    import ctypes
    from pychain_amd import _lib
    L_ = _lib.lib()
    out = (ctypes.c_int32 * 8192)()
    due = (ctypes.c_int32 * 4200)()
    for T in list(range(1, 140)) + [150, 255, 256, 257, 751, 1500, 2000, 4097]:
        n = L_.pychain_hip_debug_stream_rings(T, out, 8192, due, T + 2)
        assert n > 0 and 2 * n <= 8192
        rings = [(out[2 * r], out[2 * r + 1]) for r in range(n)]
        assert rings[0][0] == 0 and rings[-1][1] >= T
        for (lo, hi), (lo2, _hi2) in zip(rings, rings[1:]):
            assert lo < hi and hi == lo2
        widths = [hi - lo for lo, hi in rings]
        assert set(widths) <= {16, 4, 2} and widths == sorted(widths, reverse=True)
        assert all(hi - lo == 2 for lo, hi in rings if lo >= max(0, T - 8))          # the end: one frame pair per ring
        assert all(hi - lo <= 4 for lo, hi in rings if lo >= max(0, T - 48))
        for lo, hi in rings:
            if hi < T:
                assert due[hi] == 1 and hi % 2 == 0, (T, lo, hi)
        assert all(d % 2 == 0 for d in range(1, T + 1) if due[d]), T
        # the items of a sequence: ring r, side 0 (left of the middle: need = L-1-t) and side 1 (right: need = t)
        for L in {1, 2, T, max(1, T // 2), max(1, T - 1), max(1, T // 3)}:
            half = L // 2
            seen = []
            for lo, hi in rings:
                seen += list(range(max(lo, half), min(hi, L)))                         # side 1
                seen += list(range(max(0, L - hi), min(half, L - lo)))                 # side 0
            assert sorted(seen) == list(range(L)), (T, L)
    assert L_.pychain_hip_debug_stream_rings(0, out, 8192, None, 0) < 0


def _hub_graph(seed=5, H=1000, D=600, extra=7300, hubs=12, fan=70):
    """1000 states / 10 000 random arcs plus twelve states with 70 arcs ENTERING them and few leaving, and twelve with 70
    LEAVING and few entering: one such state makes its group of 64 rows 70+ slot-rows long (nothing fits a register-resident
    loop) - unless the plan puts it on several lanes."""
    from pychain_amd import ChainGraph
    from pychain_amd.simplefst import StdVectorFst
    rs = np.random.RandomState(seed)
    lo = H // 10
    arcs = [(s, (s + 1) % H, int(rs.randint(D)), float(-0.1 - 2 * rs.rand())) for s in range(H)]
    arcs += [(int(rs.randint(lo, H)), int(rs.randint(lo, H)), int(rs.randint(D)), float(-0.1 - 2 * rs.rand())) for _ in range(extra)]
    for hub in range(hubs):
        arcs += [(int(rs.randint(lo, H)), hub, int(rs.randint(D)), float(-0.1 - 2 * rs.rand())) for _ in range(fan)]
        arcs += [(lo // 2 + hub, int(rs.randint(lo, H)), int(rs.randint(D)), float(-0.1 - 2 * rs.rand())) for _ in range(fan)]
    arcs.sort(key=lambda a: a[0])
    return ChainGraph(StdVectorFst.from_arcs(H, 0, arcs, {s: 0.0 for s in range(H)}), initial_mode="leaky", final_mode="ones"), D


@pytest.mark.parametrize("case", ["small_hubs", "hubs", "structured"])
def test_states_on_several_lanes(case):
    """A state with many arcs is put on several POSITIONS of a side's numbering where that gains a shorter register-resident
    loop (csrc/plan.cpp): every arc still exactly once per (position it belongs to, position it gathers), the loop class
    falls, and what the kernels compute from such a plan (the emulator, fp64) is the fp64 oracle's result."""
    if case == "small_hubs":                                       # (a graph of C2's size: four-wave workgroups)
        (g, D), T = _hub_graph(seed=6, H=200, D=300, extra=1500, hubs=3, fan=45), 14
    elif case == "hubs":
        (g, D), T = _hub_graph(), 9
    else:
        g, D, T = syn.make_structured_den_graph(), 3456, 5
    blob = _blob(g, D)
    info, hd = _plan.plan_info(blob), emu.parse(blob)
    with _lib.option("plan_split", "0"):
        blob0 = _blob(g, D)
    info0 = _plan.plan_info(blob0)
    assert info0["split_positions"] == 0 and info0["num_states"] == g.num_states
    assert info["split_positions"] > 0 and info["graph_states"] == g.num_states
    assert info["num_states"] == g.num_states + info["split_positions"] >= g.num_states + hd["split_b"] and hd["Hp"] == (info["num_states"] + 63) // 64 * 64
    assert (info["slot_rows"] & 1023) < (info0["slot_rows"] & 1023)            # a shorter loop: what it is for
    assert ((info["slot_rows"] >> 28) & 1) == (1 if hd["split_b"] else 0)      # beta positions without the constant: not for the pair kernel
    assert (info["slot_rows"] >> 30) & 1                                      # four groups per wave at most: the lazy recursions
    # only first lanes (and no padding position) take beta's constant / carry alpha's leaky and initial probability
    assert int((~hd["takes_c"]).sum()) == hd["split_b"] and (hd["split_b"] > 0 or case == "structured")
    assert abs(hd["leaky_a"].sum() - g.leaky_probs.sum().item()) < 1e-5 and abs(hd["leaky_b"][hd["takes_c"]].sum() - g.leaky_probs.sum().item()) < 1e-5
    x = syn.make_input(2, T, D, seed=21).numpy()
    L = np.array([T, T - 3])
    objf, grad = emu.den_forward_backward(blob, x, L, 1e-3)
    objf0, grad0 = emu.den_forward_backward(blob0, x, L, 1e-3)                # the same graph, every state on one lane
    assert np.abs(objf - objf0).max() <= 1e-11 * np.abs(objf0).max() and rel_err(grad, grad0) <= 1e-11
    ro, rg, ok = orc.den(ChainGraphBatch(g, 2), np.exp(np.clip(x, -30, 30)), L, 1e-3, flavour="f64")   # (exp(x) rounded to fp32 on its way in)
    assert ok and abs(objf.sum() - ro.sum()) <= 2e-6 * abs(ro.sum()) and rel_err(grad, rg) <= 1e-5


@pytest.mark.parametrize("seed", range(10))
def test_states_on_several_lanes_random_graphs(seed):
    """Random graphs with a few hub states (many arcs in, many arcs out, self-loops on hubs, hubs feeding hubs): whatever the
    compiler decides - states on several lanes on one side, on both, or none - the emulated kernels give the same numbers as
    with every state on one lane (fp64, 1e-11) and the oracle's, every arc is in each tile once per (position it belongs
    to, position it gathers), and the per-position vectors add up to the graph's."""
    from helpers import random_hub_graph
    g, D = random_hub_graph(seed)
    H = g.num_states
    blob = _blob(g, D)
    with _lib.option("plan_split", "0"):
        blob0 = _blob(g, D)
    info, hd, hd0 = _plan.plan_info(blob), emu.parse(blob), emu.parse(blob0)
    assert info["graph_states"] == H and info["num_states"] == H + info["split_positions"] and hd0["H"] == H
    if info["split_positions"] == 0:
        assert bytes(blob) == bytes(blob0)
    else:
        assert (info["slot_rows"] & 1023) < (_plan.plan_info(blob0)["slot_rows"] & 1023)
    for name in ("alpha", "beta"):                       # the probability mass of a tile: every arc once per position it gathers
        assert hd[name]["p"].astype(np.float64).sum() >= hd0[name]["p"].astype(np.float64).sum() - 1e-9
    assert abs(hd["leaky_a"].sum() - hd0["leaky_a"].sum()) < 1e-5 and abs(hd["init_a"].sum() - hd0["init_a"].sum()) < 1e-5
    assert abs(hd["final_b"].sum() - hd0["final_b"].sum()) < 1e-4
    T = 7
    x = syn.make_input(2, T, D, seed=200 + seed).numpy()
    L = np.array([T, T - 2])
    objf, grad = emu.den_forward_backward(blob, x, L, 1e-3)
    objf0, grad0 = emu.den_forward_backward(blob0, x, L, 1e-3)
    assert np.abs(objf - objf0).max() <= 1e-11 * np.abs(objf0).max() and rel_err(grad, grad0) <= 1e-11
    ro, rg, ok = orc.den(ChainGraphBatch(g, 2), np.exp(np.clip(x, -30, 30)), L, 1e-3, flavour="f64")
    assert ok and abs(objf.sum() - ro.sum()) <= 2e-6 * abs(ro.sum()) and rel_err(grad, rg) <= 1e-5


def test_hub_to_hub_arcs_do_not_blow_up_the_occupancy_tile(capfd, monkeypatch):
    """the occupancy tile gets an arc once per (alpha position of its source, beta position of its destination), so
    arcs from a state on several alpha lanes to a state on several beta lanes are repeated parts_a x parts_b times; the 5/4
    bound of the recursion sides does not see that.  The compiler bounds the tile (3/2 of the arcs; a side gives its split up).
    The sides' own bounds keep graphs that settle at all under about 2 K, so the test lowers the bound to see the rule act -
    and what the kernels compute stays what the unsplit plan gives."""
    monkeypatch.setenv("PYCHAIN_PLAN_STATS", "1")
    monkeypatch.setenv("PYCHAIN_PLAN_CACHE_DIR", "off")
    g, D = _hub_graph(seed=9)
    H = g.num_states
    K = int(g.forward_transitions.shape[0])
    blob_free = _blob(g, D)
    free = int((emu.parse(blob_free)["gamma"]["p"] != 0).sum())
    err0 = capfd.readouterr().err
    assert "states on several lanes: loop class" in err0 and "given up" not in err0 and free > K      # split, every copy kept
    monkeypatch.setenv("PYCHAIN_PLAN_GAMMA_BOUND", str(int(100.0 * (free - 65) / K)))                 # (just under what it takes)
    blob = _blob(g, D)
    err = capfd.readouterr().err
    with _lib.option("plan_split", "0"):
        blob0 = _blob(g, D)
    hd, hd0 = emu.parse(blob), emu.parse(blob0)
    n_gamma = int((hd["gamma"]["p"] != 0).sum())
    assert int((hd0["gamma"]["p"] != 0).sum()) == K and "given up" in err, err[-1500:]
    assert n_gamma == K and K < free, (K, n_gamma, free)              # every state on one lane again
    T = 6
    x = syn.make_input(2, T, D, seed=77).numpy()
    L = np.array([T, T - 1])
    objf, grad = emu.den_forward_backward(blob, x, L, 1e-3)
    objf0, grad0 = emu.den_forward_backward(blob0, x, L, 1e-3)
    assert np.abs(objf - objf0).max() <= 1e-11 * np.abs(objf0).max() and rel_err(grad, grad0) <= 1e-11


def test_pdf_by_state_plans_and_their_one_gather_form():
    """A graph whose arcs carry the pdf of the state they enter is marked (plan format 14, launch-hint bit 27) and carries that
    pdf by alpha / beta position; the recursions' one-gather form (den_lazy.inc.h: SG; emulated from the tables alone, the arcs'
    own pdf operand unread) gives what the ordinary form gives and what the fp64 oracle gives.  The benchmark graph (a pdf per
    arc) and a graph where ONE state breaks the rule are not marked; PYCHAIN_PLAN_SG=0 marks nothing."""
    g = syn.make_structured_den_graph(200, 5, 300)
    D = 300
    blob = _blob(g, D)
    info, hd = _plan.plan_info(blob), emu.parse(blob)
    assert hd["pdf_by_state"] and (info["slot_rows"] >> 27) & 1
    # every arc of the alpha tile carries the pdf of its ROW's position, every arc of the beta tile that of the position it GATHERS
    for name, by_row in (("alpha", True), ("beta", False)):
        t = hd[name]
        for w in range(t["nwaves"]):
            first, ng, row, _ = t["waves"][w]
            for gi in range(first, first + ng):
                base, ns = t["groups"][gi]
                for j in range(ns):
                    idx, p = t["idx"][row + j], t["p"][row + j]
                    want = hd["pdf_a"][base:base + 64] if by_row else hd["pdf_b"][idx & 0xffff]
                    assert np.array_equal((idx >> 16)[p != 0], want[p != 0])
                row += ns
    T = 9
    x = syn.make_input(2, T, D, seed=3).numpy()
    L = np.array([T, T - 4])
    o1, g1 = emu.den_forward_backward(blob, x, L, 1e-3, sg=True)
    o0, g0 = emu.den_forward_backward(blob, x, L, 1e-3)
    assert np.abs(o1 - o0).max() <= 1e-11 * np.abs(o0).max() and rel_err(g1, g0) <= 1e-11
    ro, rg, ok = orc.den(ChainGraphBatch(g, 2), np.exp(np.clip(x, -30, 30)), L, 1e-3, flavour="f64")
    assert ok and abs(o1.sum() - ro.sum()) <= 2e-6 * abs(ro.sum()) and rel_err(g1, rg) <= 1e-5
    # not marked: a pdf per arc; one arc that breaks the rule; the knob
    rnd = syn.make_den_graph(200, 2000, D, seed=0)
    assert not emu.parse(_blob(rnd, D))["pdf_by_state"]
    import os
    os.environ["PYCHAIN_PLAN_SG"] = "0"
    try:
        assert not emu.parse(_blob(g, D))["pdf_by_state"]
    finally:
        del os.environ["PYCHAIN_PLAN_SG"]


def test_pdf_by_state_plans_carry_the_tables_of_the_crossing():
    """For the crossing (den_lazy.inc.h: XF, option den_cross) a pdf-by-state plan maps every alpha position to its state's beta
    position and back (the FIRST alpha position), and lists the further alpha positions of states dealt to several
    (PlanHeader::off_a2b / off_b2a / off_extra_a): a pair of positions is one state - same pdf; a beta position and its first alpha
    position the same leaky probability - and the
    further positions are exactly those the round trip does not return to."""
    D = 300
    base = syn.make_structured_den_graph(200, 5, D)
    ft = base.forward_transitions
    src, dst, pdf = ft[:, 0].numpy().copy(), ft[:, 1].numpy().copy(), ft[:, 2].numpy().copy()
    lp = np.log(base.forward_transition_probs.numpy())
    rng = np.random.default_rng(5)
    hub, hub_pdf = 7, int(pdf[np.nonzero(dst == 7)[0][0]])
    more = rng.choice(400, size=300, replace=False)                         # a hub: 300 more arcs enter state 7, with its pdf
    fst = StdVectorFst.from_arrays(400, 0, np.concatenate([src, more]), np.concatenate([dst, np.full(300, hub)]),
                                   np.concatenate([pdf, np.full(300, hub_pdf)]), np.concatenate([lp, np.full(300, np.log(0.01))]), np.zeros(400))
    for g in (base, ChainGraph(fst, initial_mode="leaky", final_mode="ones", log_domain=False)):
        hd = emu.parse(_blob(g, D))
        assert hd["pdf_by_state"]
        S, a2b, b2a, ex = hd["graph_states"], hd["a2b"], hd["b2a"], hd["extra_a"]
        HA = S + len(ex)                                                    # alpha positions: the states + their further positions
        assert np.all((a2b[:HA] >= 0) & (a2b[:HA] < S)) and np.all((b2a[:S] >= 0) & (b2a[:S] < HA))
        assert np.array_equal(a2b[b2a[:S]], np.arange(S))                   # beta -> first alpha -> the same beta position
        assert np.array_equal(hd["pdf_a"][:HA], hd["pdf_b"][a2b[:HA]]) and np.array_equal(hd["leaky_a"][b2a[:S]], hd["leaky_b"][:S])
        back = b2a[a2b[:HA]]
        further = np.nonzero(back != np.arange(HA))[0]
        assert sorted(further.tolist()) == sorted(ex[:, 1].tolist()) and np.array_equal(a2b[ex[:, 1]], ex[:, 0])
    assert len(ex) > 0                                                      # (the hub graph has such states)


def test_hint_bit_19_one_word_state_vectors():
    """ABI 17: bit 19 of the launch hint = every leaky probability positive and one position per state on either side (the lazy
    recursions' one-word state vectors, option den_q); the occupancy field beside it has nine bits."""
    D = 64
    g = syn.make_den_graph(40, 200, D, seed=3)
    t = [getattr(g, n) for n in _plan._NAMES]
    info = _plan.plan_info(_plan.build_plan_blob(*t, D, use_cache=False))
    assert (info["slot_rows"] >> 19) & 1 and ((info["slot_rows"] >> 10) & 511) > 0
    lk = g.leaky_probs.clone()
    lk[7] = 0.0                                                             # a state no leak enters: alpha cannot gather a / (coef leaky)
    t[_plan._NAMES.index("leaky_probs")] = lk
    info0 = _plan.plan_info(_plan.build_plan_blob(*t, D, use_cache=False))
    assert not (info0["slot_rows"] >> 19) & 1
