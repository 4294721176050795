#!/bin/bash
# round 4, call 30: backend scheduling strategy for pipeline_mimo_planar.hip (-mllvm -amdgpu-sched-strategy=max-ilp /
# max-memory-clause; alternative libraries built by hand into scripts/experiments/) against the default, alternating
export TMPDIR=/tmp
run() {
  for dt in f64 f32; do for dm in mindist slicer; do
    python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --demod $dm --dtype $dt --config c4 --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 c4 $dt $dm', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
  done; done
}
cp pyphysim_amd/csrc/libmcle.so /tmp/libmcle_default.so
run default
cp scripts/experiments/libmcle_max-ilp.so pyphysim_amd/csrc/libmcle.so; run max-ilp
cp scripts/experiments/libmcle_max-memory-clause.so pyphysim_amd/csrc/libmcle.so; run max-memory-clause
cp /tmp/libmcle_default.so pyphysim_amd/csrc/libmcle.so; run default
