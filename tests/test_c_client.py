"""examples/c_abi_client.c: the hot path through the C ABI from a plain C host (gcc, the HIP runtime's C API, libpychain_hip.so;
no Python, no torch in the process).  The CPU suite compiles and links it against include/pychain_hip.h; the GPU suite runs it
on the C1 workload - the denominator alone and the fused loss with per-utterance numerator graphs - and holds what it writes to
the oracle and to the Python path."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

import oracle as orc
from helpers import rel_err
from pychain_amd import ChainGraphBatch, ChainLoss, _lib, build_ext, synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(tmp_path):
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(ROCM, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc or the HIP runtime headers are not here")
    _lib.lib()                                               # (builds libpychain_hip.so if it is stale)
    exe = str(tmp_path / "c_abi_client")
    libdir = os.path.dirname(build_ext.LIB)
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROCM, "include"),
           os.path.join(ROOT, "examples", "c_abi_client.c"), "-L", libdir, "-lpychain_hip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_client_compiles_and_links_against_the_header(tmp_path):
    """A C compiler (not C++) takes include/pychain_hip.h as it is, with warnings as errors, and every entry point the client
    uses resolves in libpychain_hip.so."""
    exe = _build(tmp_path)
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    used = sorted(l.split()[-1] for l in out.splitlines() if "pychain_hip_" in l)
    assert {"pychain_hip_den_plan_build", "pychain_hip_den_plan_info", "pychain_hip_den_forward_backward",
            "pychain_hip_chain_loss_forward_backward", "pychain_hip_abi_version", "pychain_hip_den_tseg_state"} <= set(used)
    exported = subprocess.run(["nm", "-D", "--defined-only", build_ext.LIB], capture_output=True, text=True, check=True).stdout
    for sym in used:
        assert (" T " + sym) in exported, sym


def _write_problem(path, w, fused):
    g, cfg = w["den_graph"], w["cfg"]
    B, T, D = w["x"].shape
    ng = w["num_graphs"]
    Hn = int(ng.forward_transition_indices.shape[1]) if fused else 0
    Kn = int(ng.forward_transitions.shape[1]) if fused else 0
    i32 = lambda t: np.ascontiguousarray(t.cpu().numpy(), dtype=np.int32).tobytes()
    f32 = lambda t: np.ascontiguousarray(t.cpu().numpy(), dtype=np.float32).tobytes()
    with open(path, "wb") as f:
        f.write(np.array([B, T, D, g.num_states, g.num_transitions, Hn, Kn, int(fused)], dtype=np.int32).tobytes())
        f.write(np.array([1e-5], dtype=np.float32).tobytes())
        for t, conv in ((g.forward_transitions, i32), (g.forward_transition_indices, i32), (g.forward_transition_probs, f32),
                        (g.backward_transitions, i32), (g.backward_transition_indices, i32), (g.backward_transition_probs, f32),
                        (g.leaky_probs, f32), (g.initial_probs, f32), (g.final_probs, f32)):
            f.write(conv(t))
        f.write(f32(w["x"]))
        f.write(np.ascontiguousarray(w["lengths"].numpy(), dtype=np.int64).tobytes())
        if fused:
            for t, conv in ((ng.forward_transitions, i32), (ng.forward_transition_indices, i32), (ng.forward_transition_probs, f32),
                            (ng.backward_transitions, i32), (ng.backward_transition_indices, i32), (ng.backward_transition_probs, f32),
                            (ng.initial_probs, f32), (ng.final_probs, f32)):
                f.write(conv(t))


def _read_result(path, B, T, D, fused):
    raw = np.fromfile(path, dtype=np.uint8)
    off = 0

    def take(n, dtype):
        nonlocal off
        a = raw[off:off + 4 * n].view(dtype).copy()
        off += 4 * n
        return a
    out = {"den_objf": take(B, np.float32)}
    if fused:
        out["num_objf"] = take(B, np.float32)
    out["grad"] = take(B * T * D, np.float32).reshape(B, T, D)
    out["bad"] = take(2, np.int32)
    out["totals"] = take(8, np.float32)
    assert off == raw.size
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_c_client_runs_the_hot_path(tmp_path, fused):
    exe = _build(tmp_path)
    w = syn.make_workload("C1")
    B, T, D = w["x"].shape
    prob, res = str(tmp_path / "problem.bin"), str(tmp_path / "result.bin")
    _write_problem(prob, w, fused)
    r = subprocess.run([exe, prob, res], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = _read_result(res, B, T, D, fused)
    assert int(got["bad"].sum()) == 0 and float(got["totals"][1]) == float(w["lengths"].sum())
    if not fused:
        ro, rg = orc.chain_function(w["x"], w["lengths"], ChainGraphBatch(w["den_graph"], B), 1e-5, flavour="f64")
        assert abs(float(got["den_objf"].sum()) - ro) <= 1e-5 * abs(ro) and rel_err(got["grad"], rg) <= 1e-5
        assert float(got["totals"][0]) == pytest.approx(float(got["den_objf"].astype(np.float64).sum()), rel=1e-6)
    else:
        rl, rg = orc.chain_loss(w["x"], w["lengths"], w["den_graph"], w["num_graphs"], 1e-5, avg=False, flavour="f64")
        assert abs(float(got["totals"][0]) - rl) <= 1e-5 * abs(rl) and rel_err(got["grad"], rg) <= 1e-5
        # ... and the Python path over the same library gives the same bits
        x = w["x"].to("cuda:0").requires_grad_(True)
        loss = ChainLoss(w["den_graph"], 1e-5, avg=False)(x, w["lengths"], w["num_graphs"])
        loss.backward()
        assert float(loss.detach()) == float(got["totals"][0]) and np.array_equal(x.grad.cpu().numpy(), got["grad"])


@pytest.mark.gpu
def test_c_client_gets_the_burn_in_controller(tmp_path):
    """the controller that lengthens the burn-in of a time-segmented call after a miss - or stops cutting a
    plan that keeps missing - lives in the library (a device-resident state attached to the plan), so a C caller gets what the
    Python layer gets.  The C3 graph, four sequences of 640 frames, network outputs N(0,1) x 4 (they forget slowly: the default
    burn-in of 192 frames misses, and a longer one does not fit three times): the first call misses and is redone, no later one
    does - and every call writes the gradient the uncut call writes."""
    import re
    exe = _build(tmp_path)
    cfg = syn.CONFIGS["C3"]
    B, T = 4, 640
    w = dict(den_graph=syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0), cfg=cfg, num_graphs=None,
             x=syn.make_input(B, T, cfg["D"], seed=33) * 2.0, lengths=torch.full((B,), T))
    prob, res = str(tmp_path / "problem.bin"), str(tmp_path / "result.bin")
    _write_problem(prob, w, False)
    r = subprocess.run([exe, prob, res, "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = re.findall(r"call (\d): segments (\d+) rows_missed (\d+) .* next_burn_in (\d+) cooling_down (\d+)", r.stdout)
    assert len(lines) == 5, r.stdout
    segs, missed = [int(l[1]) for l in lines], [int(l[2]) for l in lines]
    assert segs[0] > 1 and missed[0] > 0, r.stdout                       # cut, missed (and redone)
    assert sum(1 for m in missed if m > 0) <= 3 and all(m == 0 for m in missed[3:]), r.stdout     # misses stop after <= 3 calls
    got = _read_result(res, B, T, cfg["D"], False)
    assert int(got["bad"].sum()) == 0
    from pychain_amd import _plan, native
    plan = _plan.graph_plan(w["den_graph"], cfg["D"], torch.device("cuda:0"))
    with _lib.option("den_tseg", 0):
        o, g, b = native.den_forward_backward(plan, w["x"].to("cuda:0"), w["lengths"], 1e-5)
    assert rel_err(got["grad"], g.cpu().numpy()) <= 1e-5 and np.abs(got["den_objf"] - o.cpu().numpy()).max() <= 1e-5 * np.abs(o.cpu().numpy()).max()
