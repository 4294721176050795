# config 3, one box: matrix-core kernel (2 workgroups per CU, default) / 3 workgroups per CU / the VALU kernel it replaces
for rep in 1 2; do
for v in "grid_oversub=0" "tdl_mfma_waves=3" "no_mfma=1"; do
python bench.py --opt $v --config c3 --steps 20 --warmup 3 --no-cpu --pmc off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.4g' % d['value'], '%.3f' % d['roofline']['kernel_ms_per_launch'])"
done; done
