# crossing on / off: the denominator call on the structured graph, uncut and cut into time segments; the fused step
for c in 1 0; do
echo -n "cross=$c uncut: "; PYCHAIN_DEN_CROSS=$c PYCHAIN_DEN_TSEG=0 TIME_DEN_STRUCTURED=1 TIME_DEN_ONLY=both python tools/time_den.py C3 2>&1 | grep -E " ms" | tr '\n' ' '; echo
echo -n "cross=$c : "; PYCHAIN_DEN_CROSS=$c python tools/time_call.py C3-structured-den 10 2>&1 | tail -n 1
done
for f in tools/variants/xf_exp*.so; do [ -f $f ] || continue; echo -n "$f : "; PYCHAIN_DEN_TSEG=0 TIME_DEN_STRUCTURED=1 TIME_DEN_ONLY=both PYCHAIN_HIP_LIB=$f python tools/time_den.py C3 2>&1 | grep -E " ms" | tr '\n' ' '; echo; done
