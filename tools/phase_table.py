"""Mean cycles per phase and wave from the output of `tools/phase_timers.sh run` (one line per wave and launch)."""
import collections, re, sys
rows = collections.defaultdict(list)
pat = re.compile(r"lazy dir (\d) wave\s+(\d+) rows\s+(\d+) steps (\d+) cycles/step: arcs (\d+) rereads\+x (\d+) "
                 r"rowstore\+sums (\d+) barrier (\d+) totals (\d+)")
for line in open(sys.argv[1]):
    m = pat.search(line)
    if m:
        head = line.split()[0]
        cnt = int(head) if head.isdigit() else 1
        d, w, r, st, *v = map(int, m.groups())
        rows[(d, w, r)] += [v] * cnt
print("cycles per frame step of den_recursion_lazy_kernel<40>, sequence 0 of C3 (T = 1500), mean over %d launches;" % max(map(len, rows.values())))
print("s_memtime around each phase (the timers themselves add ~5 % to the step); dir 1 = alpha, 0 = beta")
print("dir wave rows |  arcs  rereads+x  rowstore+sums  barrier-wait  totals |  step")
for k in sorted(rows, key=lambda k: (-k[0], k[1])):
    v = rows[k]
    m = [sum(x[i] for x in v) / len(v) for i in range(5)]
    print("%3d %4d %4d | %5.0f %10.0f %14.0f %13.0f %7.0f | %5.0f" % (k + tuple(m) + (sum(m),)))
