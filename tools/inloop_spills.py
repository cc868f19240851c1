"""Scratch reloads inside the frame loops of the recursion kernels, as compiled (no GPU): tools/inloop_spills.py [source.hip] [-a]
A register spilled and reloaded inside a frame loop costs per FRAME - round 6 lost 2.6 % of the bench step to one such reload that a
dead line of code had caused in den_recursion_lazy_kernel<32, LzNarrowDma> (profiles/r06_inloop_spills.txt).  Lists every kernel of
the translation unit (default den_lazy.hip) that has reloads inside an outermost loop with inner loops; -a: every kernel with reloads."""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("-")]
src = os.path.join(REPO, "pychain_amd", "csrc", args[0] if args else "den_lazy.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-I", os.path.join(REPO, "include"),
                    "--cuda-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
funcs, cur = [], None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\S+):\s*; @", l)
    if m:
        cur = [m.group(1), i, None]
        funcs.append(cur)
    if l.startswith(".Lfunc_end") and cur and cur[2] is None:
        cur[2] = i
worst = 0
for name, a, b in funcs:
    body = lines[a:b]
    ranges = []
    for j, l in enumerate(body):
        if "=>This Loop Header: Depth=1" in l:
            h = re.search(r"^\.(LBB\d+_\d+):", l).group(1)[1:]
            ranges.append((j, max([k for k in range(len(body)) if ("Header=" + h + " ") in body[k]] + [j])))
    reloads = [j for j, l in enumerate(body) if "scratch_load" in l]
    inl = [j for j in reloads if any(x <= j <= y for x, y in ranges)]
    if inl or ("-a" in sys.argv and reloads):
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"pychain_hip::\(anonymous namespace\)::|pychain_hip::|void |\(DenArgs\)", "", dn)
        print("%-88s reloads %3d, in frame loops %3d" % (dn[:88], len(reloads), len(inl)))
        worst = max(worst, len(inl))
print("kernels:", len(funcs), " most reloads inside a frame loop:", worst)
