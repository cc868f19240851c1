#!/usr/bin/env python3
"""LF-MMI forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3] [--no-cpu-baseline]

A step = one ChainLoss forward + backward (denominator + per-utterance numerators,
x.grad produced) over one synthetic minibatch already resident in HBM.  For N > 1 the
driver launches one rank per GPU (torch.distributed.run); every rank owns B utterances
(weak scaling: global batch = N*B), and the only exchange per step is one RCCL all-reduce
of [den_objf, num_objf, n_frames, n_bad].

Rank 0 prints ONE JSON line: the driver contract plus
  "roofline"     for the dominant kernel (den_recursion_kernel), measured here with HIP
                 events on the launch stream; algorithmic bytes are stated in DESIGN.md §4;
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
    return ap.parse_args()


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


def kernel_rooflines(w, dev, iters):
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
    try:
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
    two_frame = D % 4 == 0 and D <= 4096 and H <= 4032 and not os.environ.get("PYCHAIN_GAMMA16")
    occ_name = "den_gamma2_kernel" if two_frame else "den_gamma_kernel"
    # HBM bytes per launch from the PMC counters (separate rocprofv3 passes, summary committed
    # under profiles/): only quoted when it was measured on this very workload
    traffic = None
    try:
        with open(os.path.join(REPO, "profiles", "r01_hbm_traffic.json")) as f:
            tj = json.load(f)
        if tj.get("workload") == cfg.get("name") and tj.get("frames") == frames:
            traffic = tj["den_recursion_kernel"]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    roof = {
        "bound": "hbm", "kernel": "den_recursion_kernel",
        "achieved": round(bytes_rec / (ms_rec * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(bytes_rec / (ms_rec * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        "traffic": traffic, "ms_per_launch": round(ms_rec, 4), "algorithmic_bytes_per_launch": bytes_rec,
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
            "frac": round((bytes_rec + bytes_gam) / (out["den_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
    }
    # context: what a plain device-to-device copy reaches on this box
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


def grad_slab_allreduce(x, world, rank, dev, iters=2):
    """OPTION measured beside the hot path: one fused all-reduce of the scalars
    and the [B_global,T,D] gradient slab, so that every rank holds the whole gradient."""
    from pychain_amd.parallel import allreduce_grad_slab
    B, T, D = x.shape
    idx = torch.arange(B, device=dev) * world + rank
    stats = torch.zeros(3, device=dev)
    buf = torch.empty(3 + B * world * T * D, dtype=torch.float32, device=dev)
    g = x.grad if x.grad is not None else torch.zeros_like(x)
    allreduce_grad_slab(g, idx, B * world, stats, out=buf)
    torch.cuda.synchronize(); dist.barrier(device_ids=[dev.index]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        allreduce_grad_slab(g, idx, B * world, stats, out=buf)
    torch.cuda.synchronize(); dist.barrier(device_ids=[dev.index]); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"ms_per_step": round(ms, 3), "bytes": int(buf.numel()) * 4,
            "note": "option, not in `value`: fused all_reduce(SUM) of 3 scalars + [B_global,T,D] fp32 slab"}


def cpu_baseline(w, nsample):
    """Reference CPU path (or its C restatement) on the first `nsample` utterances of the workload."""
    import numpy as np
    from pychain_amd import ChainGraphBatch
    torch.set_num_threads(1)
    n = min(nsample, w["cfg"]["B"])
    lengths = w["lengths"][:n].clone()
    T = int(lengths.max())
    x = w["x"][:n, :T].detach().float().cpu().contiguous()
    frames = int(lengths.sum())
    den_b = ChainGraphBatch(w["den_graph"], n)
    num_b = None
    if w["num_graphs"] is not None:
        num_b = ChainGraphBatch.__new__(ChainGraphBatch)
        num_b.__dict__.update(w["num_graphs"].__dict__)
        num_b.reorder(torch.arange(n))
        num_b.batch_size = n
    kind = "port"
    try:
        import ref_loader
        ref = ref_loader.load() if ref_loader.available() else None
    except Exception:
        ref = None
    t0 = time.perf_counter()
    if ref is not None:
        kind = "reference"
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
    else:
        import oracle as orc
        orc.chain_function(x, lengths, den_b)
        if num_b is not None:
            orc.chain_function(x, lengths, num_b)
    dt = time.perf_counter() - t0
    return {"value": round(frames / dt, 1), "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": "first %d utterances (%d frames) of the same %s batch, den%s, 1 thread, %.1f s"
                      % (n, frames, w["cfg"]["name"], "+num" if num_b is not None else "", dt)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the LF-MMI path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    from pychain_amd import ChainFunction, ChainLoss, synthetic as syn
    from pychain_amd.parallel import allreduce_stats

    # every rank draws its own utterances (seed offset by rank): weak scaling, global B = world * B
    w = syn.make_workload(args.workload, device=dev, seed=0, data_seed=1000 * rank)
    w["cfg"]["name"] = args.workload
    cfg = w["cfg"]
    w["lengths_dev"] = w["lengths"].to(dev)
    x = w["x"].requires_grad_(True)
    loss_fn = ChainLoss(w["den_graph"], 1e-5, avg=False)
    local_frames = int(w["lengths"].sum())
    frames_t = torch.tensor([float(local_frames)], device=dev)

    def step():
        x.grad = None
        if w["num_graphs"] is not None:
            loss = loss_fn(x, w["lengths_dev"], w["num_graphs"])
        else:
            from pychain_amd import ChainGraphBatch
            loss = ChainFunction.apply(x, w["lengths_dev"], ChainGraphBatch(w["den_graph"], cfg["B"]))
        loss.backward()
        stats = allreduce_stats(loss.detach(), frames_t, ChainFunction.last_bad_count)
        return stats

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = step()
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    total_frames = float(stats[1])            # all-reduced frame count of one step
    n_bad = int(stats[2])
    slab = None
    if world > 1 and not args.no_grad_slab:
        try:
            slab = grad_slab_allreduce(x, world, rank, dev)
        except Exception as e:      # an option beside the metric: never lose the bench line to it
            slab = {"error": str(e)[:200]}

    if rank == 0:
        roof = kernel_rooflines(w, dev, max(3, min(args.steps, 10)))
        out = {
            "metric": "LF-MMI frames/sec (fwd+bwd)", "value": round(total_frames * args.steps / dt, 1),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: B=%d/GPU ragged T<=%d (%d frames/GPU), %d pdfs, den %d states/%d arcs%s"
                                   % (args.workload, cfg["B"], cfg["T"], local_frames, cfg["D"], cfg["H"],
                                      cfg["K"], " + per-utt log-domain numerators" if cfg["num"] else ""),
                       "global_batch": cfg["B"] * world, "parallelism": "utterance-sharded dp%d" % world,
                       "collective": "1 all_reduce(SUM) of 3 fp32 scalars per step" if world > 1 else "none"},
            "n_bad": n_bad, "roofline": roof,
        }
        if slab is not None:
            out["grad_slab_allreduce"] = slab
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_sample)
        print(json.dumps(out))
    if world > 1:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
