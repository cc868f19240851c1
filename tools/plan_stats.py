"""Per-wave load of the three tile plans of a synthetic config (host only)."""
import os, struct, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
from pychain_amd import synthetic as syn, _plan
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = syn.CONFIGS[name]
den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
b = _plan.build_plan_blob(*[getattr(den, n) for n in _plan._NAMES], cfg["D"]).tobytes()
off = 32
for tname in ("alpha", "beta", "gamma"):
    ng, nw, owt, ogt, osl, tot, mx, nrows = struct.unpack_from("8i", b, off); off += 32
    print(tname, "ngroups", ng, "slot-rows", tot, "max per wave", mx, "ideal", tot / nw)
    for w in range(nw):
        fg, n, srb, nsr = struct.unpack_from("4i", b, owt + 16 * w)
        gs = [struct.unpack_from("2i", b, ogt + 8 * (fg + i))[1] for i in range(n)]
        print("  wave %2d groups %2d rows %3d" % (w, n, nsr), gs)
