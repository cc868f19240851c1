#!/bin/bash
# segment-schedule experiments: ms per step and per denominator call
run() { echo -n "$1 segs=$2 bounds=$3 relaunch=$4 : "; env PYCHAIN_DEN_SEGMENTS=$2 PYCHAIN_DEN_BOUNDS=$3 ${4:+PYCHAIN_DEN_RELAUNCH=1} timeout 200 python bench.py --workload $1 --steps ${STEPS:-30} --warmup 6 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['den_forward_backward']['ms'])"; }
if [ $# -gt 0 ]; then run "$@"; exit; fi
run C3 3 "" 1
run C3 3 ""
run C3 4 ""
run C3 5 ""
run C3 6 ""
run C3 8 ""
run C3 4 "0.5,0.7,0.87"
run C3 5 "0.5,0.65,0.8,0.92"
run C4 3 "" 1
run C4 3 ""
run C4 5 ""
run C4 6 "0.5,0.62,0.74,0.85,0.94"
