# config 4, one box: default matrix-core kernel (variant 36) / twiddles resident (variant 32) / 2 workgroups per CU / the round-1 VALU kernel
for rep in 1 2; do
for v in "grid_oversub=0" "mfma_variant=${OTHER:-32}" "mfma_variant=21" "no_mfma=1"; do
python bench.py --opt $v --config c4 --steps 20 --warmup 3 --no-cpu --pmc off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k: '%.4g' % v for k, v in d['config']['demod_rates'].items()}, '%.3f' % d['roofline']['kernel_ms_per_launch'])"
done; done
