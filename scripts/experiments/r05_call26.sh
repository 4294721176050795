#!/bin/bash
# round 5, call 26: what the decode of f1's complex64 kernel waits for (experiments build, option mimo_tdl_kernel = 16 + code; ablations
# give wrong results by construction): 41 one bin per lane; 46 means from registers (no wave-uniform LDS reads); 47 bins from registers
# (no LDS plane reads); 48 twiddles from lane arithmetic (no gathers); 49 all three; 50 = 49 without H(f) and Gram; 51 = 50 without
# solve and demodulation
export TMPDIR=/tmp
export MCLE_LIBRARY=$PWD/pyphysim_amd/csrc/libmcle_exp.so
for round in 1 2; do
for code in 0 41 46 47 48 49 50 51 8; do
  sel=$((code == 0 ? 0 : 16 + code))
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --config f1 --dtype f32 --demod slicer --batch 98304 --opt mimo_tdl_kernel=$sel 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('code $code', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done; done
