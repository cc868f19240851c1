#!/bin/bash
# Cycles per phase of a frame step of den_recursion_lazy_kernel / den_recursion_pair_kernel (s_memtime, per wave,
# first workgroup of C3).  Run HERE to build the instrumented library, then on the GPU box:
#   tools/phase_timers.sh run [pair] > gpurun_out/phase_timers.txt ;  python tools/phase_table.py gpurun_out/phase_timers.txt
# (tools/variants/ is scratch: git-ignored .so files travel with gpurun)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "run" ]; then
  if [ "$2" = "pair" ]; then export PYCHAIN_DEN_PAIR=1; fi
  PYCHAIN_HIP_LIB=$ROOT/tools/variants/phases.so TIME_DEN_ONLY=recursion PYCHAIN_DEN_SEGMENTS=1 python $ROOT/tools/time_den.py C3 2>&1 | grep -E "^lazy dir|^pair dir|recursion ms" | sort | uniq -c | sort -k3,3n -k5,5n
else
  mkdir -p $ROOT/tools/variants
  cd $ROOT
  python -c "from pychain_amd import build_ext; print('built', build_ext.build(extra_flags=['-DPYCHAIN_PROFILE_PHASES'], lib='$ROOT/tools/variants/phases.so'))"
fi
