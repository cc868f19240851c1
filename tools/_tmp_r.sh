for n in 2 3 4 5 6; do echo -n "nseg=$n : "; PYCHAIN_DEN_SEGMENTS=$n TIME_DEN_ONLY=both python tools/time_den.py C3 2>&1 | grep " ms "; done
