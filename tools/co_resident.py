"""A fused ChainLoss step beside a long-lived kernel on another stream of the same process (what the
channels of an overlapped RCCL all-reduce look like to the loss): step time with n CUs pinned for 30 ms, for the HIP
runtime's default number of hardware queues and for more (GPU_MAX_HW_QUEUES; streams that share a hardware queue serialise).
usage (GPU box): python tools/co_resident.py [--json out]     (spawns itself once per GPU_MAX_HW_QUEUES value)"""
import json, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]


def child():
    import torch
    from pychain_amd import ChainFunction, ChainLoss, _lib, synthetic as syn
    dev = "cuda:0"
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    B, T = 64, 400
    L = syn.make_lengths(B, T, "ragged", seed=5)
    num = syn.make_num_graphs(L.tolist(), cfg["D"], seed=700)
    x = syn.make_input(B, T, cfg["D"], seed=71, device=dev)
    crit = ChainLoss(den, 1e-5)
    Ld = L.to(dev)
    work = torch.cuda.Stream()          # the loss on a stream of its own too (a trainer's compute stream is torch's current stream)
    side = torch.cuda.Stream()

    def step():
        xx = x.clone().requires_grad_(True)
        crit(xx, Ld, num).backward()
        return int(ChainFunction.last_bad_count.sum())
    rows = []
    for on_work_stream in (False, True):
        ctx = torch.cuda.stream(work) if on_work_stream else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            step(); step(); torch.cuda.synchronize()
            for pinned in (0, 32, 64, 224):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if pinned:
                    _lib.check(_lib.lib().pychain_hip_debug_occupy(pinned, 30000, side.cuda_stream), "occupy")
                bad = step()
                torch.cuda.current_stream().synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                torch.cuda.synchronize()
                rows.append(dict(loss_on="a torch.cuda.Stream" if on_work_stream else "the default stream", pinned_cus=pinned,
                                 step_ms=round(ms, 2), bad=bad))
    print("ROWS " + json.dumps(rows))


if __name__ == "__main__":
    if os.environ.get("CO_RESIDENT_CHILD"):
        child()
        sys.exit(0)
    out = {}
    for q in ("default", "8", "16"):
        env = dict(os.environ, CO_RESIDENT_CHILD="1")
        if q != "default":
            env["GPU_MAX_HW_QUEUES"] = q
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("ROWS ")]
        out["GPU_MAX_HW_QUEUES=" + q] = json.loads(line[0][5:]) if line else {"error": (r.stdout + r.stderr)[-400:]}
    txt = json.dumps(dict(what="fused ChainLoss step (C3 graph, B=64, T<=400 ragged) beside a 30 ms kernel that pins n CUs on another stream of the process",
                          results=out), indent=1)
    print(txt)
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            f.write(txt + "\n")
