#!/usr/bin/env python3
"""profiles/rNN_hbm_traffic.json from the two PMC passes of tools/profile_round.sh.
usage: traffic_json.py fetch.db write.db workload frames out.json

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, WRITE as counted;
calibrated in round 1 on the recursion kernel's own known byte counts, profiles/r01_hbm_traffic.json)."""
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    per = defaultdict(float)
    for k, c, v, d in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if c == counter:
            per[(k, d)] += v
    out = defaultdict(list)
    for (k, d), v in per.items():
        out[k].append(v)
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def short(k):
    for n in ("den_recursion_lazy_kernel", "den_recursion_pair_kernel", "den_recursion_kernel", "den_gamma2_kernel", "den_gamma_kernel",
              "den_finish_kernel", "den_exp_rows_kernel", "den_splice_check_kernel"):
        if n in k:
            # a call cut into time segments launches the lazy recursion twice: the segmented kernel (last
            # template argument true) and, behind the check, the uncut one, which leaves at once unless a splice missed
            if n == "den_recursion_lazy_kernel":
                # (template arguments: rows, LDS map, row form, TS, NC)
                import re
                m = re.search(r"den_recursion_lazy_kernel<[^>]*?,\s*(true|false),\s*(true|false)>", k)
                return n + (" [time segments]" if (m and m.group(1) == "true") else "")
            return n
    return None


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a SEPARATE pass, --pmc WRITE_SIZE over "
               "`PYCHAIN_DEN_SEGMENTS=1 python tools/time_den.py %s` (tools/profile_round.sh), averaged over the "
               "dispatches of each kernel; counter unit KiB; hbm_bytes_per_launch = (2 x FETCH + WRITE) x 1024 "
               "(FETCH doubled per the microarch guide, calibration in profiles/r01_hbm_traffic.json)" % sys.argv[3],
       "time_segments": None,
       "workload": sys.argv[3], "frames": int(sys.argv[4])}
cut = any(short(k) == "den_recursion_lazy_kernel [time segments]" for k in set(fetch) | set(write))
for k in sorted(set(fetch) | set(write)):
    n = short(k)
    if n is None:
        continue
    if cut and n == "den_recursion_lazy_kernel":
        n += " [fallback launch behind the check: leaves at once]"
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out[n] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "dispatches": nf.get(k, 0),
              "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
# the whole denominator call (one launch of each kernel under PYCHAIN_DEN_SEGMENTS=1): recursion + occupancy + finish
# [+ the rows exp'd ahead: den_exp_rows_kernel, calls of the denominator alone]
out["time_segments"] = bool(cut)
out["den_call_hbm_bytes"] = int(sum(v["hbm_bytes_per_launch"] for k, v in out.items() if isinstance(v, dict)))
with open(sys.argv[5], "w") as fo:
    json.dump(out, fo, indent=1)
print(json.dumps(out, indent=1))
