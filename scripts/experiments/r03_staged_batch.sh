#!/bin/bash
# staged config 4: realizations/s and algorithmic-byte fraction against the batch (working set per pass = 214 KB x batch:
# 1024 -> 219 MB, inside the 256 MB Infinity Cache; 8192 -> 1.75 GB)
export TMPDIR=/tmp
for b in 512 1024 2048 4096 8192 16384 32768; do
  python scripts/bench_staged_c4.py --seconds 1 --batch $b | python -c "import json,sys; d=json.load(sys.stdin); print('batch', d['batch'], '%.4g /s' % d['realizations_per_s'], 'frac %.3f' % d['frac'], '%.3f ms/pass' % d['ms_per_pass'])"
done
