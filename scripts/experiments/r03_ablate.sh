#!/bin/bash
# stage ablation of k_run_mimo_ofdm_f64 (temporary patch: mask in the upper bits of grid_oversub)
for ab in 0 1 2 4 8 16 32 6 63 33 9; do
  v=$((ab * 256))
  python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --dtype f64 --config c4 --single-demod --opt grid_oversub=$v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate $ab', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"
done
