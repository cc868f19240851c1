#!/bin/bash
# times every build/variants/lib_*.so (ablation builds) with tools/time_den.py
for f in build/variants/lib_*.so; do
  echo -n "$f : "
  PYCHAIN_HIP_LIB=$f PYCHAIN_DEN_SEGMENTS=${SEGS:-1} TIME_DEN_ONLY=${1:-gamma} python tools/time_den.py ${2:-C3} 2>&1 | grep " ms " | tr '\n' ' '
  echo
done
