export TIME_DEN_ONLY=recursion PYCHAIN_DEN_SEGMENTS=1
for s in 0 1 2 3 4; do for b in 0 170; do echo -n "slack=$s balance=$b : "; PYCHAIN_PLAN_SLACK=$s PYCHAIN_PLAN_BALANCE=$b python tools/time_den.py C3 2>&1 | grep " ms "; done; done
