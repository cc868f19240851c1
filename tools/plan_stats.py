"""Per-wave load of the tile plans of a synthetic config + the plan compiler's own statistics (host only)."""
import os, struct, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
os.environ.setdefault("PYCHAIN_PLAN_STATS", "1")
from pychain_amd import synthetic as syn, _plan
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = syn.CONFIGS[name]
den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
t0 = time.time()
b = _plan.build_plan_blob(*[getattr(den, n) for n in _plan._NAMES], cfg["D"]).tobytes()
print("plan build %.2f s, %d bytes" % (time.time() - t0, len(b)))
def tile(off, tname):
    ng, nw, owt, ogt, osl, tot, mx, nrows = struct.unpack_from("8i", b, off)
    print(tname, "ngroups", ng, "waves", nw, "slot-rows", tot, "max per wave", mx, "ideal %.1f" % (tot / nw))
    if "-v" in sys.argv:
        for w in range(nw):
            fg, n, srb, nsr = struct.unpack_from("4i", b, owt + 16 * w)
            gs = [struct.unpack_from("2i", b, ogt + 8 * (fg + i))[1] for i in range(n)]
            print("  wave %2d groups %2d rows %3d" % (w, n, nsr), gs)
for i, tname in enumerate(("alpha", "beta", "gamma")):
    tile(32 + 32 * i, tname)
tile(160, "gamma2"); tile(192, "alpha4"); tile(224, "beta4")
