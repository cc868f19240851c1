# crossing on / off on the structured graph's denominator at B = 96 and 128 (2B workgroups on 256 CUs)
for B in 96 128; do for c in 0 1; do
echo -n "B=$B cross=$c : "; PYCHAIN_DEN_CROSS=$c timeout 120 python tools/time_call.py "C3-structured@B=$B" 10 2>&1 | tail -n 1
done; done
