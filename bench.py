#!/usr/bin/env python3
"""LF-MMI forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3] [--no-cpu-baseline]

A step = one ChainLoss forward + backward (denominator + per-utterance numerators,
x.grad produced) over one synthetic minibatch already resident in HBM.  For N > 1 there is
one rank per GPU: either the caller launched them (torch.distributed.run: WORLD_SIZE is
set and must equal --gpus), or - WORLD_SIZE unset - this script re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` on 127.0.0.1.  The run is the PRODUCT'S
sharded path (pychain_amd/parallel.py): every rank builds the same global minibatch of N*B
utterances (weak scaling; --workload C3 at N = 8 is BASELINE.json's C5: global B = 512),
takes its shard with `parallel.shard_batch` (length-sorted serpentine deal), draws the network
output of exactly the utterances it owns, and steps `parallel.ShardedChainLoss`: the only
exchange per step is ONE RCCL all-reduce of [objf, n_frames, n_bad].  N = 1
goes through the same code with a world of one.

`--dry-run` runs that very path on a box without a GPU (gloo, tiny shapes, a stand-in loss that
is a plain torch function of the shard - no kernels, "value" is meaningless and the line says so)
and checks what sharding can get wrong: every utterance owned once, frames add up, the global
loss equals the single-process loss (tests/test_bench_launch.py).

Rank 0 prints ONE JSON line: the driver contract plus
  "roofline"     for the dominant kernel (the denominator's alpha/beta recursion launch), measured here with
                 HIP events on the launch stream; algorithmic bytes are stated in DESIGN.md §4; `traffic` is
                 the HBM byte count of the SAME launch from the committed rocprofv3 PMC passes
                 (profiles/r*_hbm_traffic.json, `traffic_source` names the file), not re-measured per run;
  "cpu_baseline" the reference's own CPU path (oracle/_ref, kind "reference") or, when that
                 binary is absent, the C restatement (kind "port"), timed on a bounded
                 sample of the same workload on this box's host cores (N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
# A step of the loss runs on three streams (the caller's and two side streams); RCCL's collectives add a fourth.  The HIP runtime
# maps streams onto FOUR hardware queues by default, and a stream of the loss that shares its queue with RCCL's serialises behind
# it: measured on MI355X with the step's 12-byte all-reduce forced through RCCL in a world of one, 4.21 ms per step against 3.08
# without the collective - and 3.15 ms with 16 queues (profiles/r06_rccl_one_rank.txt).  Read by the runtime when it initialises:
# set before anything touches the device (pychain_amd does the same on import where it is imported early enough).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grad-slab", action="store_true",
                    help="skip the optional fused all-reduce of the gradient slab (N > 1 only)")
    ap.add_argument("--cpu-sample", type=int, default=48, help="utterances in the CPU baseline sample")
    ap.add_argument("--dry-run", action="store_true",
                    help="harness check without a GPU: gloo, no-op steps (tests/test_bench_launch.py)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short runs of the other BASELINE.json configurations reported beside the metric (N = 1)")
    ap.add_argument("--no-rooflines", action="store_true",
                    help="skip the per-kernel roofline measurement on rank 0 (at N > 1 it is a short one: 3 launches per kernel, "
                         "no d2d copy probe - the other ranks wait for it in a barrier)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test mode for boxes with fewer GPUs than ranks: rank r runs on GPU r %% (visible GPUs) and the collectives go "
                         "over gloo (device tensors through the host) - the sharded GPU path in N processes without N GPUs; never a "
                         "scaling measurement (tests/test_gpu_configs.py)")
    ap.add_argument("--force-collective", action="store_true",
                    help="under a launcher with ONE rank (torch.distributed.run --nproc-per-node 1): initialise RCCL and run the step's "
                         "all-reduce of [objf, frames, bad] although the world is one - the real library beside the loss's side streams "
                         "and spin-wait kernels on a box with a single GPU (tests/test_gpu_configs.py)")
    ap.add_argument("--no-fresh-num-graphs", action="store_true",
                    help="skip the host-side leg: a fresh numerator ChainGraphBatch per step, as a trainer builds it")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 without a launcher: become `torch.distributed.run` with N ranks on this node."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
    raise SystemExit(subprocess.call(cmd, env=env))


def event_time_ms(fn, iters, stream):
    """Average duration of fn() bracketed by events on `stream` (the stream the library launches on)."""
    start = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    stop = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        start[i].record(stream)
        fn()
        stop[i].record(stream)
    torch.cuda.synchronize()
    return sum(s.elapsed_time(e) for s, e in zip(start, stop)) / iters


def step_roofline(workload, frames, step_ms):
    """`roofline.step`: the fused step's algorithmic bytes / counter bytes / their ratio, from the newest committed
    profiles/r*_<workload>_step_hbm_traffic.json of this very workload (labelled, not re-measured), over the step time of THIS run."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_step_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                tj = json.load(f)
            if tj.get("workload") == workload and tj.get("frames") == frames:
                algo, cnt = int(tj["algorithmic_bytes_per_call"]), int(tj["hbm_bytes_per_call"])
                return {"algorithmic_bytes": algo, "traffic": cnt, "traffic_over_algorithmic": round(cnt / algo, 3),
                        "ms": round(step_ms, 4),
                        "achieved": round(algo / (step_ms * 1e-3) / 1e9, 2), "frac": round(algo / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "traffic_GBps": round(cnt / (step_ms * 1e-3) / 1e9, 1), "traffic_frac_of_peak": round(cnt / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "per_kernel_traffic": {k: v["hbm_bytes_per_call"] for k, v in tj.get("kernels", {}).items() if v["hbm_bytes_per_call"] >= (1 << 20)},
                        "traffic_source": "profiles/" + os.path.basename(path) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same step; not re-measured in this run)"}
        except Exception:
            continue
    return {"traffic": None}


def kernel_rooflines(w, dev, iters, d2d=True, step_ms=None):
    """Per-kernel launch time of the denominator, each launch isolated with the phase mask."""
    from pychain_amd import _lib, _plan, native
    L = _lib.lib()
    cfg = w["cfg"]
    D, H = cfg["D"], cfg["H"]
    plan = _plan.graph_plan(w["den_graph"], D, dev)
    stream = torch.cuda.current_stream(dev)
    frames = int(w["lengths"].sum())
    call = lambda: native.den_forward_backward(plan, w["x"], w["lengths_dev"], 1e-5)
    out = {}
    # the launches as the TIMED step runs them: a call of the denominator alone has its rows exp'd ahead of the recursions
    # (den_exp_rows_kernel), the fused loss has not - option den_dma = 2 is that form
    import contextlib
    fused = w.get("num_graphs") is not None
    as_in_step = _lib.option("den_dma", 2) if fused else contextlib.nullcontext()
    # ... and cut into as many time segments as the step's call is (a fused call at B = 64 is not cut, a call of
    # the denominator alone at B = 64 would be)
    tsegs = int(L.pychain_hip_den_time_segments(plan.stride, plan.slot_rows, H, D, cfg["B"], cfg["T"], int(fused)))
    as_cut = _lib.option("den_tseg", tsegs if tsegs > 1 else 0)
    try:
        with as_in_step, as_cut:
            call(); torch.cuda.synchronize()
            for name, mask in (("den_recursion_kernel", 1), ("den_gamma_kernel", 2), ("den_call", 3)):
                L.pychain_hip_set_den_phase_mask(mask)
                call(); torch.cuda.synchronize()
                out[name] = event_time_ms(call, iters, stream)
    finally:
        L.pychain_hip_set_den_phase_mask(3)
    # algorithmic bytes per live sequence-frame (DESIGN.md §4 / SURVEY.md §8(d)):
    #   recursion launch: x row read by the alpha and by the beta pass (8D) + alpha' row written (4(H+1))
    #   occupancy launch: grad row written (4D) + alpha' row read (4(H+1))
    bytes_rec = (8 * D + 4 * (H + 1)) * frames
    bytes_gam = (4 * D + 4 * (H + 1)) * frames
    ms_rec, ms_gam = out["den_recursion_kernel"], out["den_gamma_kernel"]
    # the kernels this shape runs, as the library's own launcher decides (include/pychain_hip.h: pychain_hip_den_kernel_names)
    rec_name, occ_name = _lib.den_kernel_names(plan.slot_rows, plan.num_states, D, cfg["B"])
    # HBM bytes per launch from the PMC counters (separate rocprofv3 passes, summary committed under
    # profiles/ by tools/profile_round.sh): the newest file measured on this very workload and kernel
    traffic, traffic_source = None, None
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                tj = json.load(f)
            base = rec_name.split("<")[0]           # (the lazy recursion's shapes share their counters' kernel name)
            if tj.get("workload") != cfg.get("name") or tj.get("frames") != frames:
                continue
            kn = tj.get("kernels", {}).get(base)    # (round 6: the passes over the whole step, every kernel - tools/step_traffic_json.py)
            if kn is not None and kn.get("dispatches_per_call"):
                traffic = int(kn["hbm_bytes_per_call"] / kn["dispatches_per_call"])
            elif base in tj:
                traffic = tj[base]["hbm_bytes_per_launch"]
            if traffic is not None:
                traffic_source = "profiles/" + os.path.basename(path) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same launch; not re-measured in this run)"
                break
        except Exception:
            continue
    call_traffic = None
    try:
        if traffic_source is not None and "den_call_hbm_bytes" in tj:
            call_traffic = int(tj["den_call_hbm_bytes"])
    except Exception:
        pass
    roof = {
        "bound": "hbm", "kernel": rec_name, "time_segments": tsegs,
        "achieved": round(bytes_rec / (ms_rec * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(bytes_rec / (ms_rec * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        "traffic": traffic, "traffic_source": traffic_source,
        "ms_per_launch": round(ms_rec, 4), "algorithmic_bytes_per_launch": bytes_rec,
        # the occupancy launch is den_gamma2_kernel (two frames per pass) where the graph fits it
        # (pdf count <= 4096 and a multiple of 4, <= 4032 states), else den_gamma_kernel
        "other_kernels": {occ_name: {"ms_per_launch": round(ms_gam, 4),
                                               "achieved": round(bytes_gam / (ms_gam * 1e-3) / 1e9, 2),
                                               "algorithmic_bytes_per_launch": bytes_gam}},
        # the whole denominator call as shipped (occupancy launches overlapped with the recursion
        # segments on a side stream), bracketed by events on the caller's stream
        "den_forward_backward": {
            "algorithmic_bytes": bytes_rec + bytes_gam, "ms": round(out["den_call"], 4),
            "achieved": round((bytes_rec + bytes_gam) / (out["den_call"] * 1e-3) / 1e9, 2),
            "frac": round((bytes_rec + bytes_gam) / (out["den_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            # PMC bytes of every launch of the call (recursion + occupancy + finish [+ rows exp'd ahead]) from the same committed
            # file, and their ratio to the algorithmic bytes: what is re-read
            "traffic": call_traffic,
            "traffic_over_algorithmic": round(call_traffic / (bytes_rec + bytes_gam), 3) if call_traffic else None},
    }
    # the numerator's launches (side stream, hidden beside the first half of the recursion at B = 64, CU time from B = 128
    # on): the fused call with the denominator's launches masked out - recursions only (num_fb_kernel), then with the
    # compact occupancy rows (num_prep + num_fb + num_occ_wave)
    if w.get("num_graphs") is not None:
        try:
            roof["other_kernels"].update(numerator_rooflines(w, plan, dev, iters, stream))
        except Exception as e:          # a context figure: never lose the bench line to it
            roof["other_kernels"]["numerator"] = {"error": str(e)[:200]}
    # the WHOLE fused step: algorithmic bytes (denominator 12 D + 8 (H + 1), numerator 8 U_n + 8 (H_n + 1) per
    # live frame) against the counter bytes of every kernel of the step from the committed PMC passes (tools/profile_round6.sh)
    if fused and step_ms:
        roof["step"] = step_roofline(cfg.get("name"), frames, step_ms)
    # context: what a plain device-to-device copy reaches on this box
    if not d2d:
        return roof
    try:
        src = torch.empty(256 << 20, dtype=torch.float32, device=dev)      # 1 GiB
        dst = torch.empty_like(src)
        dst.copy_(src); torch.cuda.synchronize()
        ms = event_time_ms(lambda: dst.copy_(src), 5, stream)
        roof["d2d_copy_GBps"] = round(2 * src.numel() * 4 / (ms * 1e-3) / 1e9, 1)
        del src, dst
    except Exception:
        roof["d2d_copy_GBps"] = None
    return roof


def numerator_rooflines(w, plan, dev, iters, stream):
    """ms per launch of the numerator kernels and their algorithmic bytes (8 U_n + 8 (H_n + 1) per live
    frame in a fused loss - the utterance's distinct pdfs read by both recursions, its state row written and read - split as
    recursions 8 U_n + 4 (H_n + 1), occupancy 4 (H_n + 1))."""
    from pychain_amd import _lib, native
    ng = w["num_graphs"]
    gt = ng.device_tensors(dev)
    gstride = 0 if ng.shared_graph is not None else 1
    x, ld = w["x"].detach(), w["lengths_dev"]
    call = lambda g: native.chain_loss_forward(plan, gt, gstride, ng.num_states, x, ld, 1e-5, with_grad=g)
    with _lib.option("den_phase_mask", 0):
        call(True); torch.cuda.synchronize()
        ms_fb = event_time_ms(lambda: call(False), iters, stream)
        ms_all = event_time_ms(lambda: call(True), iters, stream)
    native.release_workspaces()
    L = w["lengths"].tolist()
    ft, fi = ng.forward_transitions, ng.forward_transition_indices
    bytes_fb = bytes_occ = 0
    for b, Lb in enumerate(L):
        kused = int(fi[b, :, 1].max())
        U = int(torch.unique(ft[b, :kused, 2]).numel())
        Hn = int((fi[b, :, 1] > fi[b, :, 0]).sum())                 # states with arcs (the padding has none)
        bytes_fb += Lb * (8 * U + 4 * (Hn + 1))
        bytes_occ += Lb * 4 * (Hn + 1)
    ms_occ = max(ms_all - ms_fb, 1e-6)
    mk = lambda ms, nbytes: {"ms_per_launch": round(ms, 4), "algorithmic_bytes_per_launch": int(nbytes),
                             "achieved": round(nbytes / (ms * 1e-3) / 1e9, 2), "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    return {"num_fb_kernel": dict(mk(ms_fb, bytes_fb), bound="latency: T dependent frames of one barrier each, one state per thread; not HBM"),
            "num_prep_kernel+num_occ_wave_kernel": dict(mk(ms_occ, bytes_occ), note="time-parallel; by difference: fused call with - without the compact occupancy rows"),
            "numerator_forward_backward": mk(ms_all, bytes_fb + bytes_occ)}


def _adhoc_workload(name, B, dev, equal=False, den_only=False, dtype=None, structured=False, num_compat=False):
    """BASELINE config `name`, optionally at another batch size (same graph, ragged lengths drawn for B), with all
    sequences of the full length (`equal`), without numerators (`den_only`), with a 2-byte network output (`dtype`)."""
    from pychain_amd import synthetic as syn
    if B is None and not equal and not den_only and not structured:
        w = syn.make_workload(name, device=dev)
    else:
        cfg = dict(syn.CONFIGS[name])
        cfg["B"] = B = B or cfg["B"]
        if equal:
            cfg["lengths"] = "equal"
        if den_only:
            cfg["num"] = False
        lengths = syn.make_lengths(B, cfg["T"], cfg["lengths"], seed=2)
        # (`structured`: a phone-LM-like graph of the same size - arcs entering a state share its pdf, strong self-loops)
        den = syn.make_structured_den_graph(cfg["H"] // 2, (cfg["K"] // (cfg["H"] // 2) - 2) // 2, cfg["D"]) if structured else \
            syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
        w = dict(cfg=cfg, lengths=lengths, den_graph=den,
                 num_graphs=syn.make_num_graphs(lengths.tolist(), cfg["D"], seed=100) if cfg["num"] else None,
                 x=syn.make_input(B, cfg["T"], cfg["D"], seed=1, device=dev))
    if dtype is not None:
        w["x"] = w["x"].to(dtype)
    w["lengths_dev"] = w["lengths"].to(dev)
    return w


def _committed_call_traffic(workload, frames, algorithmic):
    """PMC bytes of the whole denominator call (every launch: recursion, occupancy, finish, the rows exp'd ahead) from the
    newest committed profiles/r*_<workload>_hbm_traffic.json measured on this very workload - labelled, not re-measured."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                tj = json.load(f)
            if tj.get("workload") == workload and tj.get("frames") == frames and "den_call_hbm_bytes" in tj:
                t = int(tj["den_call_hbm_bytes"])
                return {"traffic": t, "traffic_over_algorithmic": round(t / algorithmic, 3),
                        "traffic_source": "profiles/" + os.path.basename(path) + " (PMC passes of the same call; not re-measured in this run)"}
        except Exception:
            continue
    return {"traffic": None}


def other_workloads(dev, steps=6, warmup=3):
    """The other single-GPU configurations of BASELINE.json and the bench shape at larger batches, a few steps each:
    reported BESIDE the metric (`other_workloads`), never in `value`.  C4 is BASELINE.json's "HBM-roofline run"."""
    from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, native
    out = {}
    # "C3-equal": B = 64, every sequence 1500 frames, denominator only - the exact configuration BASELINE.md §3 states the
    # 40 % target on; "C3-bf16": the bench batch with a bf16 network output read by the kernels as it is
    for label, name, B, kw in (("C4", "C4", None, {}), ("C2", "C2", None, {}), ("C3-equal", "C3", None, dict(equal=True, den_only=True)),
                               ("C3-bf16", "C3", None, dict(dtype=torch.bfloat16)),
                               ("C3-structured", "C3", None, dict(structured=True)),
                               # ... and its denominator alone on 64 x 1500 frames: the configuration the 40 % target is stated on, on a
                               # graph whose arcs carry the pdf of the state they enter (what a phone-LM denominator looks like)
                               ("C3-structured-equal", "C3", None, dict(structured=True, equal=True, den_only=True)),
                               # the configuration that meets the LITERAL 1e-4 against the reference at benchmark length (option
                               # num_compat: the numerator in the reference's own fp32 arithmetic, a checking mode - DESIGN.md §3.10)
                               ("C3-num_compat", "C3", None, dict(num_compat=True)),
                               # few sequences: the recursions are cut into time segments
                               ("C3@B=16", "C3", 16, {}), ("C3@B=32", "C3", 32, {}),
                               ("C3@B=128", "C3", 128, {}), ("C3@B=256", "C3", 256, {})):
        try:
            w = _adhoc_workload(name, B, dev, **kw)
            cfg = w["cfg"]
            half = w["x"].dtype != torch.float32
            x = w["x"].requires_grad_(True)
            crit = ChainLoss(w["den_graph"], 1e-5, avg=False)
            gb = ChainGraphBatch(w["den_graph"], cfg["B"])

            last = {}

            def step():
                x.grad = None
                if w["num_graphs"] is not None:
                    last["loss"] = crit(x, w["lengths_dev"], w["num_graphs"])
                else:
                    last["loss"] = ChainFunction.apply(x, w["lengths_dev"], gb, 1e-5)
                last["loss"].backward()
            import contextlib
            compat = bool(kw.get("num_compat"))
            nsteps = 2 if compat else steps                       # (a ~40 ms per sequence-slice checking mode: two steps state its cost)
            with (_lib.option("num_compat", 1) if compat else contextlib.nullcontext()):
                for _ in range(1 if compat else warmup):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(nsteps):
                    step()
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / nsteps * 1e3
            frames = int(w["lengths"].sum())
            plan = _plan.graph_plan(w["den_graph"], cfg["D"], dev)
            rec, occ = _lib.den_kernel_names(plan.slot_rows, plan.num_states, cfg["D"], cfg["B"], fused=bool(cfg["num"]))
            stream = torch.cuda.current_stream(dev)
            # time segments of the call the step makes (totals[5..7] of include/pychain_hip.h: segments per (sequence, direction),
            # speculated rows that did not verify, the worst mismatch seen; 1 / 0 / 0 where the call is not cut)
            fused_call = w["num_graphs"] is not None
            tsegs = int(_lib.lib().pychain_hip_den_time_segments(plan.stride, plan.slot_rows, plan.num_states, cfg["D"], cfg["B"], cfg["T"], int(fused_call)))
            tot8 = last["loss"].totals_all.detach().float().cpu().tolist() if getattr(last["loss"], "totals_all", None) is not None else None
            call = lambda: native.den_forward_backward(plan, w["x"].detach(), w["lengths_dev"], 1e-5)
            parts = {}
            import contextlib
            for key, mask in (("recursion_ms", 1), ("occupancy_ms", 2), ("den_ms", 3)):
                # (as the step runs them: kernel_rooflines)
                with _lib.option("den_phase_mask", mask), (_lib.option("den_dma", 2) if cfg["num"] else contextlib.nullcontext()), \
                        _lib.option("den_tseg", tsegs if tsegs > 1 else 0):
                    call(); torch.cuda.synchronize()
                    parts[key] = event_time_ms(call, 3, stream)
            # (2-byte rows: x read twice at 2 B, the gradient written at 2 B: 6 D instead of 12 D per frame)
            den_bytes = ((6 if half else 12) * cfg["D"] + 8 * (cfg["H"] + 1)) * frames
            uses_rows = bool(_lib.lib().pychain_hip_den_uses_row_buffer(plan.stride, plan.slot_rows, plan.num_states, cfg["D"], cfg["B"], cfg["T"], 0)) and not cfg["num"]
            out[label] = {
                "workload": "%s: B=%d T<=%d (%d frames), %d pdfs, den %d states/%d arcs%s%s"
                            % (name, cfg["B"], cfg["T"], frames, cfg["D"], cfg["H"], cfg["K"], " + numerators" if cfg["num"] else ", denominator only",
                               (", %s network output and gradient" % str(w["x"].dtype).replace("torch.", "") if half else "") +
                               (", numerator in the reference's own fp32 arithmetic (option num_compat: within 1e-4 of the reference at "
                                "this length, tests/test_gpu_compat.py)" if compat else "")),
                # the [B,T,D] fp32 buffer of the rows exp'd ahead of the recursions: only calls of the denominator alone
                "rows_exp_ahead_workspace_bytes": 4 * cfg["B"] * cfg["T"] * cfg["D"] if uses_rows else 0,
                "ms_per_step": round(ms, 4), "frames_per_s": round(frames / ms * 1e3, 1), "steps": nsteps,
                "recursion_kernel": rec, "occupancy_kernel": occ,
                "time_segments": tsegs,
                "splices_redone": None if tot8 is None else int(tot8[5]), "worst_splice_mismatch": None if tot8 is None else tot8[7],
                "recursion_ms": round(parts["recursion_ms"], 4), "occupancy_ms": round(parts["occupancy_ms"], 4),
                "den_forward_backward": dict({"algorithmic_bytes": den_bytes, "ms": round(parts["den_ms"], 4),
                                              "frac": round(den_bytes / (parts["den_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                                             **_committed_call_traffic(label, frames, den_bytes)),
                "n_bad": int(last["loss"].bad_count.sum()),
            }
            del w, x, crit, gb
        except Exception as e:          # context figures: never lose the bench line to them
            out[label] = {"error": str(e)[:200]}
        native.release_workspaces()
        torch.cuda.empty_cache()
    return out


def fresh_num_graphs(w, dev, reps=5):
    """Host work the metric does not see: a drop-in trainer builds a NEW numerator ChainGraphBatch from its list of
    per-utterance ChainGraph objects every step (pychain/graph.py:122-175) and the loss uploads it.  Timed on the bench
    batch: constructor (one native pack into a pinned staging buffer) + upload (ONE H2D copy) + on-device reorder."""
    from pychain_amd import ChainGraph, ChainGraphBatch, synthetic as syn
    lengths = w["lengths"].tolist()
    D = w["cfg"]["D"]
    graphs = [ChainGraph(syn.make_num_fst(int(min(max(round(Tb / 4.0), 4), 400, Tb)), D, 100 + b), log_domain=True)
              for b, Tb in enumerate(lengths)]
    mk, mh = max(g.num_transitions for g in graphs), max(g.num_states for g in graphs)
    order = torch.randperm(len(graphs), generator=torch.Generator().manual_seed(3))
    t_build = t_up = t_re = 0.0
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        gb = ChainGraphBatch(graphs, max_num_transitions=mk, max_num_states=mh)
        t1 = time.perf_counter()
        gb.device_tensors(dev)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        gb.reorder(order)
        gb.device_tensors(dev)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if _ > 0:
            t_build += t1 - t0; t_up += t2 - t1; t_re += t3 - t2
    return {"utterances": len(graphs), "construct_ms": round(t_build / reps * 1e3, 3), "upload_ms": round(t_up / reps * 1e3, 3),
            "reorder_and_restage_ms": round(t_re / reps * 1e3, 3),
            "note": "per step on the launching thread; NOT in `value` (the reference pays the same work in Python: graph.py:122-194)"}


def grad_slab_allreduce(x, idx, global_batch, dev, iters=2):
    """OPTION measured beside the hot path: one fused all-reduce of the scalars
    and the [B_global,T,D] gradient slab, so that every rank holds the whole gradient.  `idx`: this rank's utterances."""
    from pychain_amd.parallel import allreduce_grad_slab
    B, T, D = x.shape
    idx = idx.to(dev)
    stats = torch.zeros(3, device=dev)
    buf = torch.empty(3 + global_batch * T * D, dtype=torch.float32, device=dev)
    g = x.grad if x.grad is not None else torch.zeros_like(x)
    allreduce_grad_slab(g, idx, global_batch, stats, out=buf)
    torch.cuda.synchronize(); dist.barrier(device_ids=[dev.index]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        allreduce_grad_slab(g, idx, global_batch, stats, out=buf)
    torch.cuda.synchronize(); dist.barrier(device_ids=[dev.index]); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"ms_per_step": round(ms, 3), "bytes": int(buf.numel()) * 4,
            "note": "option, not in `value`: fused all_reduce(SUM) of 3 scalars + [B_global,T,D] fp32 slab"}


def _cpu_sample(w, idx):
    """CPU-resident (x, lengths, den graph batch, numerator graph batch) of the utterances `idx`."""
    from pychain_amd import ChainGraphBatch
    idx = torch.as_tensor(idx, dtype=torch.long)
    lengths = w["lengths"].index_select(0, idx).clone()
    T = int(lengths.max())
    x = w["x_cpu"].index_select(0, idx)[:, :T].contiguous()
    den_b = ChainGraphBatch(w["den_graph"], int(idx.numel()))
    num_b = None
    if w["num_graphs"] is not None:
        num_b = ChainGraphBatch.__new__(ChainGraphBatch)
        num_b.__dict__.update(w["num_graphs"].__dict__)
        num_b._device_cache = {}
        num_b.reorder(idx)
        num_b.batch_size = int(idx.numel())
    return x, lengths, den_b, num_b


def _cpu_run(x, lengths, den_b, num_b):
    """One evaluation of den (+ num) on the CPU: the reference binary when oracle/_ref holds it
    (kind "reference"), else the C restatement (kind "port").  Lengths must be sorted descending."""
    try:
        import ref_loader
        ref = ref_loader.load() if ref_loader.available() else None
    except Exception:
        ref = None
    if ref is not None:
        bs = torch.nn.utils.rnn.pack_padded_sequence(x, lengths, batch_first=True).batch_sizes
        xc = x.clamp(-30, 30)
        g = lambda t: t.contiguous()
        ref.forward_backward(g(den_b.forward_transitions), g(den_b.forward_transition_indices),
                             g(den_b.forward_transition_probs), g(den_b.backward_transitions),
                             g(den_b.backward_transition_indices), g(den_b.backward_transition_probs),
                             g(den_b.leaky_probs), g(den_b.initial_probs), g(den_b.final_probs),
                             den_b.start_state, xc.exp(), bs, lengths, den_b.num_states, 1e-5)
        if num_b is not None:
            out = ref.forward_backward_log_domain(
                g(num_b.forward_transitions), g(num_b.forward_transition_indices),
                g(num_b.forward_transition_probs), g(num_b.backward_transitions),
                g(num_b.backward_transition_indices), g(num_b.backward_transition_probs),
                g(num_b.initial_probs), g(num_b.final_probs), num_b.start_state, xc, bs, lengths,
                num_b.num_states)
            out[1].exp()
        return "reference"
    import oracle as orc
    orc.chain_function(x, lengths, den_b)
    if num_b is not None:
        orc.chain_function(x, lengths, num_b)
    return "port"


def _cpu_worker(rank, nworkers, w, idx, reps, barrier, out_q):
    """All-cores leg: worker `rank` evaluates its share of the sample `reps` times, one thread."""
    torch.set_num_threads(1)
    mine = idx[rank::nworkers]
    x, lengths, den_b, num_b = _cpu_sample(w, mine)
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(reps):
        kind = _cpu_run(x, lengths, den_b, num_b)
    out_q.put((rank, time.perf_counter() - t0, int(lengths.sum()) * reps, kind))


def cpu_baseline(w, nsample, all_cores=True):
    """Reference CPU path (or its C restatement) on the first `nsample` utterances of the workload:
    (i) one thread - the faithful counterpart of the reference, whose loops over sequences, states and
    arcs are serial (chain-computation.cc:113-176); (ii) all host cores, utterances dealt to one
    single-threaded worker process per core."""
    import copy
    torch.set_num_threads(1)
    n = min(nsample, w["cfg"]["B"])
    # (the worker processes get the graphs WITHOUT what the GPU run cached on them: device-resident plans and uploads)
    den_graph = copy.copy(w["den_graph"])
    den_graph._plan_cache = {}
    num_graphs = w["num_graphs"]
    if num_graphs is not None:
        num_graphs = copy.copy(num_graphs)
        num_graphs._device_cache = {}
    w = {"cfg": w["cfg"], "lengths": w["lengths"], "den_graph": den_graph, "num_graphs": num_graphs,
         "x_cpu": w["x"].detach().float().cpu().contiguous().share_memory_()}
    order = torch.argsort(w["lengths"], descending=True, stable=True)      # (already sorted in C1-C4)
    sample = _cpu_sample(w, order[:n])
    frames = int(sample[1].sum())
    t0 = time.perf_counter()
    kind = _cpu_run(*sample)
    dt = time.perf_counter() - t0
    res = {"value": round(frames / dt, 1), "unit": "frames/s", "cores": 1, "kind": kind,
           "sample": "first %d utterances (%d frames) of the same %s batch, den%s, 1 thread, %.1f s"
                     % (n, frames, w["cfg"]["name"], "+num" if sample[3] is not None else "", dt)}
    if all_cores:
        try:
            res["all_cores"] = _cpu_all_cores(w, order, dt / n)
        except Exception as e:            # a context figure: never lose the bench line to it
            res["all_cores"] = {"error": str(e)[:200]}
    return res


def _cpu_all_cores(w, order, sec_per_utt):
    import torch.multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nw = max(1, min(cores, int(order.numel())))
    per = -(-int(order.numel()) // nw)                                    # utterances per worker
    reps = max(1, min(20, int(8.0 / max(1e-3, sec_per_utt * per))))       # about 8 s of work per worker
    ctx = mp.get_context("spawn")                                         # (fork after HIP init is unsafe)
    barrier, q = ctx.Barrier(nw), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(r, nw, w, order, reps, barrier, q)) for r in range(nw)]
    for p_ in procs:
        p_.start()
    got = []
    deadline = time.time() + 180
    while len(got) < nw:
        try:
            got.append(q.get(timeout=2))
        except Exception:
            if time.time() > deadline or any(p_.exitcode not in (None, 0) for p_ in procs):
                for p_ in procs:
                    p_.terminate()          # (our own children, by handle)
                raise RuntimeError("cpu worker failed or timed out")
    for p_ in procs:
        p_.join()
    tmax = max(g[1] for g in got)
    frames = sum(g[2] for g in got)
    # (`cores` = the threads that did work: one utterance per worker leaves the other host cores idle - said in the line)
    return {"value": round(frames / tmax, 1), "unit": "frames/s", "cores": nw, "host_cores": cores, "workers": nw, "kind": got[0][3],
            "sample": "all %d utterances of the batch dealt to %d single-threaded worker processes (%d of the %d host cores used), "
                      "%d repetitions each, slowest worker %.1f s" % (int(order.numel()), nw, nw, cores, reps, tmax)}


class _DenOnlyLoss(torch.nn.Module):
    """`loss_cls` of ShardedChainLoss for the workloads without numerators (C2, C4): the denominator ChainFunction."""
    reports_bad_count = True

    def __init__(self, den_graph, leaky, avg=False):
        super().__init__()
        self.den_graph, self.leaky = den_graph, leaky

    def forward(self, x, lengths, num_graphs):
        from pychain_amd import ChainFunction, ChainGraphBatch
        return ChainFunction.apply(x, lengths, ChainGraphBatch(self.den_graph, x.size(0)), self.leaky)


class _DryLoss(torch.nn.Module):
    """--dry-run stand-in for the per-rank loss (no kernels, no GPU): a plain torch function of the shard that depends on
    every input the sharding moves - the network output of each utterance up to ITS length and ITS numerator graph - so a
    wrong deal, a wrong reorder or a lost utterance changes the global sum."""

    def __init__(self, den_graph, leaky, avg=False):
        super().__init__()

    def forward(self, x, lengths, num_graphs):
        T = x.size(1)
        live = (torch.arange(T)[None, :] < torch.as_tensor(lengths)[:, None]).to(x.dtype)
        per_utt = (x.double().sum(-1) * live).sum(-1)
        tag = num_graphs.forward_transitions.double().sum((1, 2)) + num_graphs.final_probs.clamp(min=-1e3).double().sum(1)
        return (per_utt * (1.0 + 1e-3 * tag)).sum().float()


def rank_shard(args, world, rank, dev, dry=False):
    """The global minibatch of the run, identical on every rank, and THIS rank's shard of it through the product's
    partitioner (parallel.shard_batch).  Returns (shard dict like synthetic.make_workload's, global dict, idx)."""
    from pychain_amd import synthetic as syn
    from pychain_amd.parallel import shard_batch
    if dry:
        B, T, D = 3, 24, 16                                # per rank; tiny: the stand-in loss is a torch expression
        cfg = dict(B=B, T=T, H=12, K=40, D=D, lengths="ragged", num=True, B_global=B * world, name="dry")
        lengths = syn.make_lengths(B * world, T, "ragged", seed=2)
        g = dict(cfg=cfg, lengths=lengths, den_graph=syn.make_den_graph(12, 40, D, seed=0),
                 num_graphs=syn.make_num_graphs(lengths.tolist(), D, seed=100, max_states=6))
    elif world == 1:
        # the configuration BASELINE.json's metric is quoted on, the same bytes as in every earlier round
        w = syn.make_workload(args.workload, device=dev)
        g = dict(cfg=dict(w["cfg"], B_global=w["cfg"]["B"]), lengths=w["lengths"], den_graph=w["den_graph"],
                 num_graphs=w["num_graphs"], x=w["x"])
    else:
        g = syn.make_global_workload(args.workload, world)
    cfg = g["cfg"]
    xs, ls, gs, idx = shard_batch(g.pop("x", None), g["lengths"], g["num_graphs"], world, rank)
    if xs is None:
        xs = syn.make_input_utterances(idx, cfg["T"], cfg["D"], seed=1, device=dev)
    shard = dict(cfg=dict(cfg, B=int(idx.numel()), name=args.workload), x=xs, lengths=ls, lengths_dev=ls.to(dev),
                 den_graph=g["den_graph"], num_graphs=gs)
    return shard, g, idx


def workload_label(args, cfg, world, local_frames, global_frames):
    name = "C5" if (args.workload == "C3" and world == 8) else (args.workload if world == 1 else "%sx%d" % (args.workload, world))
    what = "%d pdfs, den %d states/%d arcs%s" % (cfg["D"], cfg["H"], cfg["K"], " + per-utt log-domain numerators" if cfg["num"] else "")
    if world == 1:
        return "%s: B=%d ragged T<=%d (%d frames), %s" % (name, cfg["B"], cfg["T"], local_frames, what)
    return ("%s: global B=%d ragged T<=%d (%d frames), batch-sharded over %d GPUs by parallel.shard_batch (%d utterances, "
            "%d frames on rank 0), %s" % (name, cfg["B_global"], cfg["T"], global_frames, world, cfg["B"], local_frames, what))


def main():
    args = parse()
    # fewer visible devices than ranks: one clear line instead of N ranks dying in set_device / RCCL init
    if not args.dry_run and not args.share_gpu and torch.cuda.is_available() and args.gpus > torch.cuda.device_count():
        raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) are visible on this node" % (args.gpus, torch.cuda.device_count()))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the LF-MMI path has no CPU fallback")
    if dry:
        dev = torch.device("cpu")
    else:
        gpu = local_rank % torch.cuda.device_count() if args.share_gpu else local_rank
        torch.cuda.set_device(gpu)
        dev = torch.device("cuda", gpu)
    use_dist = world > 1 or (args.force_collective and "WORLD_SIZE" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry or args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    from pychain_amd.parallel import ShardedChainLoss

    w, glob, idx = rank_shard(args, world, rank, dev, dry)
    cfg = w["cfg"]
    x = w["x"].requires_grad_(True)
    loss_cls = _DryLoss if dry else (None if cfg["num"] else _DenOnlyLoss)
    loss_fn = ShardedChainLoss(w["den_graph"], 1e-5, avg=False, loss_cls=loss_cls, force_collective=use_dist and world == 1)
    local_frames = int(w["lengths"].sum())
    lengths_step = w["lengths"] if dry else w["lengths_dev"]

    def step():
        x.grad = None
        loss = loss_fn(x, lengths_step, w["num_graphs"])     # local evaluation + the ONE all-reduce of [objf, frames, bad]
        loss.backward()
        return loss

    def fence():
        if not dry:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=None if (dry or args.share_gpu) else [local_rank])
        if not dry:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    stats = loss_fn.last_stats
    # per-rank view (imbalance between the shards is the only thing that can cost the scaling): each rank's own time
    # for the K steps, its frame count, its longest sequence and its utterance count, gathered once after the timed region
    mine = torch.tensor([dt, float(local_frames), float(w["lengths"].max()), float(idx.numel())], device=dev, dtype=torch.float64)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    owned = [torch.full((cfg["B"],), -1, dtype=torch.int64, device=dev) for _ in range(world)]
    if world > 1:
        dist.all_gather(owned, idx.to(dev))
    else:
        owned = [idx.to(dev)]
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    total_frames = float(stats[1])            # all-reduced frame count of one step
    n_bad = int(stats[2])
    global_frames = int(glob["lengths"].sum())
    slab = None
    if world > 1 and not args.no_grad_slab and not dry:
        try:
            slab = grad_slab_allreduce(x, idx, cfg["B_global"], dev)
        except Exception as e:      # an option beside the metric: never lose the bench line to it
            slab = {"error": str(e)[:200]}

    if rank == 0:
        frames_pr = [int(p[1]) for p in per_rank]
        ms_pr = [float(p[0]) / args.steps * 1e3 for p in per_rank]
        all_owned = torch.cat([o.cpu() for o in owned]).sort().values
        sharding = {
            "partitioner": "pychain_amd.parallel.shard_batch (length-sorted serpentine deal) + ShardedChainLoss",
            "every_utterance_owned_once": bool(torch.equal(all_owned, torch.arange(cfg["B_global"]))),
            "frames_add_up": bool(sum(frames_pr) == global_frames and abs(total_frames - global_frames) < 0.5),
            "frames_imbalance_max_over_mean": round(max(frames_pr) * world / max(1, sum(frames_pr)), 4),
            "ms_imbalance_max_over_min": round(max(ms_pr) / max(1e-9, min(ms_pr)), 4),
        }
        per_rank_out = {"ms_per_step": [round(m, 4) for m in ms_pr], "frames": frames_pr,
                        "longest_sequence": [int(p[2]) for p in per_rank], "utterances": [int(p[3]) for p in per_rank],
                        "ms_per_step_min": round(min(ms_pr), 4), "ms_per_step_max": round(max(ms_pr), 4)}
        config = {"workload": "dry run (no kernels): " + workload_label(args, cfg, world, local_frames, global_frames) if dry
                  else workload_label(args, cfg, world, local_frames, global_frames),
                  "global_batch": cfg["B_global"], "parallelism": "utterance-sharded dp%d" % world,
                  "collective": "1 all_reduce(SUM) of 3 fp32 scalars per step" + (" (forced: a world of one under a launcher, RCCL)" if world == 1 else "")
                                if use_dist else "none"}
        if dry:
            # what only a CPU run can afford: the whole global batch in ONE process against the sharded sum
            from pychain_amd import synthetic as syn
            xg = syn.make_input_utterances(range(cfg["B_global"]), cfg["T"], cfg["D"], seed=1)
            single = float(_DryLoss(None, 0)(xg, glob["lengths"], glob["num_graphs"]))
            sharding["loss_equals_single_process"] = bool(abs(float(loss.detach()) - single) <= 1e-5 * max(1.0, abs(single)))
            print(json.dumps({
                "metric": "LF-MMI frames/sec (fwd+bwd)", "value": None, "unit": "frames/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none",
                "dry_run": True, "frames_per_step_all_ranks": total_frames, "global_frames": global_frames,
                "loss": float(loss.detach()), "loss_single_process": single,
                "per_rank": per_rank_out, "sharding": sharding, "config": config}))
        else:
            # (N > 1: the other ranks idle in a barrier behind this: 3 launches per kernel there, ~50 ms)
            roof = None if args.no_rooflines else kernel_rooflines(w, dev, 3 if world > 1 else max(3, min(args.steps, 10)), d2d=world == 1,
                                                                   step_ms=dt / args.steps * 1e3 if world == 1 else None)
            out = {
                "metric": "LF-MMI frames/sec (fwd+bwd)", "value": round(total_frames * args.steps / dt, 1),
                "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": config,
                "n_bad": n_bad, "roofline": roof, "per_rank": per_rank_out, "sharding": sharding,
            }
            if roof is None:
                out["roofline_note"] = "per-kernel rooflines skipped (--no-rooflines)"
            if slab is not None:
                out["grad_slab_allreduce"] = slab
            if world == 1 and cfg["num"] and not args.no_fresh_num_graphs:
                try:
                    out["host_graph_batch"] = fresh_num_graphs(w, dev)
                except Exception as e:
                    out["host_graph_batch"] = {"error": str(e)[:200]}
            if world == 1 and not args.no_other_workloads and args.workload == "C3":
                keep = {k: w[k] for k in ("cfg", "lengths", "den_graph", "num_graphs")}
                x_cpu = w["x"].detach().float().cpu() if not args.no_cpu_baseline else None
                del x
                w.pop("x")
                torch.cuda.empty_cache()
                out["other_workloads"] = other_workloads(dev)
                if x_cpu is not None:
                    keep["x"] = x_cpu
                w = keep
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(w, args.cpu_sample)
            print(json.dumps(out))
    if use_dist:
        dist.barrier(device_ids=None if (dry or args.share_gpu) else [local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
