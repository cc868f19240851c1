#!/usr/bin/env python3
"""HBM bytes of EVERY kernel of a call from two PMC passes (FETCH_SIZE, WRITE_SIZE - each its own rocprofv3 run, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes), per call of tools/time_call.py:
    step_traffic_json.py fetch.db write.db <label> <frames> <calls> <algorithmic bytes per call> out.json
hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (KiB counters; FETCH doubled per the guide for wide coalesced reads on gfx950;
calibration: profiles/r01_hbm_traffic.json).  Kernels of the host framework (fills, copies) are listed too: the sum is the step's."""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter, calls):
    """kernel name -> (sum over the dispatches that belong to the `calls` calls, dispatches per call).  What runs BEFORE the calls
    (the script's set-up: input generation, uploads) is left out: of a kernel's n dispatches, in dispatch order, the last
    (n // calls) * calls are the calls' own."""
    db = sqlite3.connect(path)
    per = defaultdict(float)
    for k, c, v, d in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if c == counter:
            per[(k, d)] += v
    by = defaultdict(list)
    for (k, d), v in per.items():
        by[k].append((d, v))
    tot, n = {}, {}
    for k, lst in by.items():
        lst.sort()
        keep = (len(lst) // calls) * calls
        if keep == 0:
            continue
        tot[k] = sum(v for _, v in lst[len(lst) - keep:])
        n[k] = keep
    return tot, n


def short(k):
    m = re.match(r"(?:void )?(?:pychain_hip::)?(?:\(anonymous namespace\)::)?([A-Za-z_0-9:]+)", k)
    s = m.group(1) if m else k
    s = s.replace("pychain_hip::", "").replace("(anonymous namespace)::", "")
    if "den_recursion_lazy_kernel" in k:
        t = re.search(r"den_recursion_lazy_kernel<[^>]*?,\s*(true|false),\s*(true|false)(?:,\s*(true|false))?>", k)
        if t and t.group(1) == "true":
            s += " [time segments]"
    return s[:80]


label, frames, calls, algo = sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE", calls)
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE", calls)
out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a SEPARATE pass, --pmc WRITE_SIZE over `python tools/time_call.py %s` "
               "(tools/profile_round6.sh); the dispatches of the %d calls the script makes (warm-up calls included; its set-up - input generation, "
               "uploads - left out), divided by the number of calls; counter unit KiB; hbm_bytes = (2 x FETCH + WRITE) x 1024" % (label, calls),
       "workload": label, "frames": frames, "calls": calls, "kernels": {}}
names = defaultdict(lambda: [0.0, 0.0, 0])
for k in set(fetch) | set(write):
    e = names[short(k)]
    e[0] += fetch.get(k, 0.0); e[1] += write.get(k, 0.0); e[2] += max(nf.get(k, 0), nw.get(k, 0))
total = 0
for n, (f, w, d) in sorted(names.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
    b = int((2 * f + w) * 1024 / calls)
    if b < 1024:
        continue
    out["kernels"][n] = {"FETCH_SIZE_KiB_per_call": round(f / calls, 1), "WRITE_SIZE_KiB_per_call": round(w / calls, 1),
                         "dispatches_per_call": round(d / calls, 2), "hbm_bytes_per_call": b}
    total += b
out["hbm_bytes_per_call"] = total
out["algorithmic_bytes_per_call"] = algo
out["traffic_over_algorithmic"] = round(total / algo, 3) if algo else None
if not label.startswith("C3-num") and label not in ("C3", "C3-structured"):
    out["den_call_hbm_bytes"] = total            # (a call of the denominator alone: what bench.py's other_workloads look up)
elif label in ("C3", "C3-structured"):           # (a fused step: its denominator's kernels)
    out["den_call_hbm_bytes"] = sum(v["hbm_bytes_per_call"] for k, v in out["kernels"].items() if k.startswith("den_") or k.startswith("zero_words"))
with open(sys.argv[7], "w") as fo:
    json.dump(out, fo, indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_how"}, indent=1))
