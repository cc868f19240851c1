"""Averages over waves and calls of the phase timers of tools/xf_phases.sh (gpurun_out/xf_phases_{1,0}.txt)."""
import re, collections, sys
names = "arcs rereads rowstore barrier totals vmwait hk_totals hk_rows xf_tail -".split()
for c in (1, 0):
    acc = collections.defaultdict(lambda: [0] * (len(names) + 1))
    for l in open(f'gpurun_out/xf_phases_{c}.txt'):
        if not l.startswith("lazy dir"): continue
        v = [int(x) for x in re.findall(r"\d+", l)]
        a = acc[v[0]]
        for i in range(min(len(names), len(v) - 4)): a[i] += v[4 + i]
        a[-1] += 1
    for d, a in sorted(acc.items()):
        n = a[-1]
        print("cross", c, "dir", d, "n", n, " ".join("%s %d" % (k, x // n) for k, x in zip(names, a)), "| step", sum(a[:5]) // n)
