mkdir -p gpurun_out
for c in 1 0; do PYCHAIN_DEN_CROSS=$c PYCHAIN_HIP_LIB=tools/variants/phases.so TIME_DEN_ONLY=both TIME_DEN_STRUCTURED=1 PYCHAIN_DEN_TSEG=0 python tools/time_den.py C3 > gpurun_out/xf_phases_$c.txt 2>&1; done
