# config 4, one box: default matrix-core kernel (variant 36) / twiddles resident (variant 32) / 2 workgroups per CU / the round-1 VALU kernel
for rep in 1 2; do
for v in "MCLE_X=1" "MCLE_MFMA_VARIANT=${OTHER:-32}" "MCLE_MFMA_VARIANT=21" "MCLE_NO_MFMA=1"; do
env $v python bench.py --config c4 --steps 20 --warmup 3 --no-cpu --pmc off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k: '%.4g' % v for k, v in d['config']['demod_rates'].items()}, '%.3f' % d['roofline']['kernel_ms_per_launch'])"
done; done
