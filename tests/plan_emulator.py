"""NumPy emulation of what the HIP denominator kernels compute from a compiled plan.

Test infrastructure: lets the CPU suite validate the host-side plan compiler
(pychain_amd/csrc/plan.cpp) and the kernel decomposition (independent alpha/beta
normalisers + per-frame-normalised occupancies, pychain_amd/csrc/den_kernels.hip)
against the oracle without a GPU.  Mirrors the kernels step by step.
"""
import numpy as np

HDR_INTS = 8 + 3 * 8 + 8 + 3 * 8 + 4 + 4 + 16 + 4  # PlanHeader of csrc/plan_format.h (PLAN_VERSION 15), in int32 words


def parse(blob):
    b = np.asarray(blob, dtype=np.uint8)
    h = b[:HDR_INTS * 4].view(np.int32)
    assert h[0] == 0x4C504843, "bad magic"
    hd = dict(version=int(h[1]), H=int(h[2]), K=int(h[3]), D=int(h[4]), Hp=int(h[5]), total_bytes=int(h[6]))

    def tile(o):
        t = dict(zip(["ngroups", "nwaves", "off_wave_tab", "off_group_tab", "off_slots",
                      "total_slot_rows", "max_wave_slot_rows", "nrows"], [int(v) for v in h[o:o + 8]]))
        t["waves"] = b[t["off_wave_tab"]:t["off_wave_tab"] + 16 * t["nwaves"]].view(np.int32).reshape(-1, 4)
        t["groups"] = b[t["off_group_tab"]:t["off_group_tab"] + 8 * t["ngroups"]].view(np.int32).reshape(-1, 2)
        n = t["total_slot_rows"] * 64
        s = b[t["off_slots"]:t["off_slots"] + 8 * n].view(np.uint32).reshape(-1, 64, 2)
        t["idx"] = s[:, :, 0].copy()
        t["p"] = s[:, :, 1].copy().view(np.float32)
        return t
    hd["alpha"], hd["beta"], hd["gamma"] = tile(8), tile(16), tile(24)
    hd["gamma2"], hd["alpha4"], hd["beta4"] = tile(40), tile(48), tile(56)
    hd["rec_max_wave_groups"], hd["rec4_max_wave_groups"], hd["header_hash"], hd["payload_hash"] = int(h[7]), int(h[38]), int(h[39]), int(h[64])
    # the graph's states (H counts positions), and the beta positions that take no constant c(t) (plan.cpp, "states on several lanes")
    hd["graph_states"], off_nc, n_nc = int(h[65]), int(h[66]), int(h[67])
    hd["no_const"] = b[off_nc:off_nc + 4 * n_nc].view(np.int32).copy()
    hd["split_b"] = n_nc
    offs = [int(v) for v in h[32:38]]
    Hp = hd["Hp"]
    vec = lambda o: b[o:o + 4 * Hp].view(np.float32).copy()
    hd["init_a"], hd["leaky_a"], hd["final_a"], hd["leaky_b"], hd["final_b"] = [vec(o) for o in offs[:5]]
    # beta positions that take the constant c(t): all but the listed ones (a state's lanes after its first)
    hd["takes_c"] = np.ones(Hp, dtype=bool)
    hd["takes_c"][hd["no_const"]] = False
    ng = max(hd["gamma"]["ngroups"] * 64, 64)
    hd["row_pdf"] = b[offs[5]:offs[5] + 4 * ng].view(np.int32).copy()
    # "pdf by state" (format 14): every arc entering a state carries one pdf; its id by alpha / beta position
    hd["pdf_by_state"] = bool(int(h[68]) & 1)
    if hd["pdf_by_state"]:
        hd["pdf_a"] = b[int(h[69]):int(h[69]) + 4 * Hp].view(np.int32).copy()
        hd["pdf_b"] = b[int(h[70]):int(h[70]) + 4 * Hp].view(np.int32).copy()
        # ... and the occupancy tiles over states (one pseudo-arc per state position: {alpha position, beta position, 1})
        hd["gamma_sg"], hd["gamma2_sg"] = tile(72), tile(80)
        ngs = max(hd["gamma_sg"]["ngroups"] * 64, 64)
        hd["row_pdf_sg"] = b[int(h[71]):int(h[71]) + 4 * ngs].view(np.int32).copy()
        # the crossing's tables: alpha position -> the state's beta position, beta position -> its FIRST alpha position, and
        # the further alpha positions of states on several {beta position, alpha position}
        hd["a2b"] = b[int(h[88]):int(h[88]) + 4 * Hp].view(np.int32).copy()
        hd["b2a"] = b[int(h[89]):int(h[89]) + 4 * Hp].view(np.int32).copy()
        hd["extra_a"] = b[int(h[90]):int(h[90]) + 8 * int(h[91])].view(np.int32).reshape(-1, 2).copy()
    return hd


def tile_rows(t, U, V, nout, dtype):
    """out[row] = sum_k p_k U[i0_k] V[i1_k], evaluated wave by wave / group by group."""
    out = np.zeros(nout, dtype=dtype)
    for w in range(t["nwaves"]):
        first, ng, row, _ = t["waves"][w]
        for g in range(first, first + ng):
            base, ns = t["groups"][g]
            acc = np.zeros(64, dtype=dtype)
            for j in range(ns):
                idx = t["idx"][row + j]
                acc += t["p"][row + j].astype(dtype) * U[idx & 0xffff] * V[idx >> 16]
            row += ns
            out[base:base + 64] = acc
    return out


def tile_rows_sg(t, U, nout, dtype):
    """The one-gather form (den_lazy.inc.h: SG): out[row] = sum_k p_k U[i0_k] - the nnet output is not gathered per arc."""
    out = np.zeros(nout, dtype=dtype)
    for w in range(t["nwaves"]):
        first, ng, row, _ = t["waves"][w]
        for g in range(first, first + ng):
            base, ns = t["groups"][g]
            acc = np.zeros(64, dtype=dtype)
            for j in range(ns):
                acc += t["p"][row + j].astype(dtype) * U[t["idx"][row + j] & 0xffff]
            row += ns
            out[base:base + 64] = acc
    return out


def den_forward_backward(blob, x, lengths, coef, input_is_exp=False, dtype=np.float64, sg=False):
    """Returns (objf_per_seq[B], grad[B,T,D]) exactly as den_kernels.hip computes them.  `sg`: the recursions in the one-gather
    form of a "pdf by state" plan - alpha: x(t, pdf of the row) times the row's sum; beta: the gathered vector pre-multiplied by
    x(t, pdf of the gathered state) - from the plan's per-position pdf tables (the arcs' own pdf operand is not read)."""
    hd = parse(blob)
    if sg:
        assert hd["pdf_by_state"], "not a pdf-by-state plan"
    H, Hp, D = hd["H"], hd["Hp"], hd["D"]
    B, T, _ = x.shape
    ex = x.astype(dtype) if input_is_exp else np.exp(np.clip(x.astype(dtype), -30, 30))
    objf = np.zeros(B, dtype=dtype)
    grad = np.zeros((B, T, D), dtype=dtype)
    mask = hd["takes_c"].astype(dtype)
    la, lb = hd["leaky_a"].astype(dtype), hd["leaky_b"].astype(dtype)
    for b in range(B):
        L = int(lengths[b])
        A = np.zeros((L + 1, Hp), dtype=dtype)
        Bt = np.zeros((L + 1, Hp), dtype=dtype)
        v = hd["init_a"].astype(dtype)
        tot = v.sum()
        logsum = np.log(tot)
        A[0] = v / tot + coef * la
        Araw = np.zeros((L + 1, Hp), dtype=dtype)            # a(t,.) before the leaky term (any per-frame scale): what an sg alpha recursion stores
        for t in range(1, L + 1):
            v = ex[b, t - 1][hd["pdf_a"]] * tile_rows_sg(hd["alpha"], A[t - 1], Hp, dtype) if sg else tile_rows(hd["alpha"], A[t - 1], ex[b, t - 1], Hp, dtype)
            tot = v.sum()
            logsum += np.log(tot)
            A[t] = v / tot + coef * la
            Araw[t] = v
        objf[b] = logsum + np.log((A[L] * hd["final_a"].astype(dtype)).sum())
        v = hd["final_b"].astype(dtype)
        Bt[L] = (v + mask * coef * (v * lb).sum()) / v.sum()
        for t in range(L - 1, 0, -1):
            v = tile_rows_sg(hd["beta"], ex[b, t][hd["pdf_b"]] * Bt[t + 1], Hp, dtype) if sg else tile_rows(hd["beta"], Bt[t + 1], ex[b, t], Hp, dtype)
            Bt[t] = (v + mask * coef * (v * lb).sum()) / v.sum()
        g = hd["gamma_sg"] if sg else hd["gamma"]
        rp = hd["row_pdf_sg"] if sg else hd["row_pdf"]
        for t in range(L):
            # sg: a sum over STATES of a(t+1,j) beta(t+1,j) per pdf - no nnet-output row (DenArgs::sg)
            st = tile_rows(g, Araw[t + 1] if sg else A[t], Bt[t + 1], max(g["ngroups"] * 64, 64), dtype)
            q = np.zeros(D, dtype=dtype)
            ok = rp[:len(st)] >= 0
            q[rp[:len(st)][ok]] = st[ok]
            gm = q if sg else ex[b, t] * q
            grad[b, t] = gm / gm.sum()
    return objf, grad
