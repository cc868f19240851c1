# the bench step with the library of this tree and with others (tools/variants/<name>.so ...), alternating on ONE box
for i in 1 2; do for lib in "" "$@"; do
[ -n "$lib" ] && lib=tools/variants/$lib.so
echo -n "${lib:-this tree} : "; PYCHAIN_HIP_LIB=$lib python bench.py --steps 50 --no-cpu-baseline --no-other-workloads --no-rooflines 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done
