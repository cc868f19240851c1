"""Mean cycles per phase and wave from the output of `tools/phase_timers.sh run [pair]` (one line per wave and launch)."""
import collections, re, sys
PATS = {
    "lazy": (re.compile(r"lazy dir (\d) wave\s+(\d+) rows\s+(\d+) steps (\d+) cycles/step: arcs (\d+) rereads\+x (\d+) "
                        r"rowstore\+sums (\d+) barrier (\d+) totals (\d+)"),
             ["arcs", "rereads+x", "rowstore+sums", "barrier-wait", "totals"], "den_recursion_lazy_kernel<40>, sequence 0"),
    "pair": (re.compile(r"pair dir (\d) wave\s+(\d+) rows\s+(\d+) steps (\d+) cycles/step: arcs (\d+) x (\d+) wsums (\d+) "
                        r"bar1 (\d+) normalise (\d+) bar2 (\d+)"),
             ["arcs", "x-rows", "wave-sums", "barrier-1", "totals+normalise+stores", "barrier-2"],
             "den_recursion_pair_kernel<40>, sequences 0 and 1"),
}
text = open(sys.argv[1]).read()
for kind, (pat, names, what) in PATS.items():
    rows = collections.defaultdict(list)
    for line in text.splitlines():
        m = pat.search(line)
        if m:
            head = line.split()[0]
            cnt = int(head) if head.isdigit() else 1
            d, w, r, st, *v = map(int, m.groups())
            rows[(d, w, r)] += [v] * cnt
    if not rows:
        continue
    print("cycles per frame step of %s of C3 (T = 1500), mean over %d launches;" % (what, max(map(len, rows.values()))))
    print("s_memtime around each phase (the timers themselves add ~5 % to the step); dir 1 = alpha, 0 = beta")
    print("dir wave rows | " + "  ".join(names) + " |  step")
    for k in sorted(rows, key=lambda k: (-k[0], k[1])):
        v = rows[k]
        m = [sum(x[i] for x in v) / len(v) for i in range(len(names))]
        print("%3d %4d %4d | " % k + "  ".join("%*.0f" % (len(n), x) for n, x in zip(names, m)) + " | %5.0f" % sum(m))
