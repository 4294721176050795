#!/bin/bash
# round 4, call 24: complex64 planar kernel at (1024, 4x4): register bound 4 wavefronts per SIMD (259 fused / 260 unfused), and the
# counters of the planar forms next to the matrix-core kernel's
export TMPDIR=/tmp
mkdir -p gpurun_out
for dm in mindist slicer; do
for t in 0 257 259 260; do
  python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --demod $dm --dtype f32 --config c4 --batch 262144 --opt f32_planar=1 --opt f64_threads=$t 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 f32 $dm planar threads=$t', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --demod $dm --dtype f32 --config c4 --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 f32 $dm mfma', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
TAG=c4p257 bash scripts/prof_one.sh c4 --dtype f32 --demod mindist --batch 262144 --opt f32_planar=1 --opt f64_threads=257 > /dev/null 2>&1
TAG=c4p0 bash scripts/prof_one.sh c4 --dtype f32 --demod mindist --batch 262144 --opt f32_planar=1 > /dev/null 2>&1
TAG=c4mf bash scripts/prof_one.sh c4 --dtype f32 --demod mindist --batch 262144 > /dev/null 2>&1
for t in c4p257 c4p0; do echo "== $t"; for d in gpurun_out/prof_${t}_*; do python scripts/pmc_summary.py $d k_run_mimo_ofdm_planar 262144 2>/dev/null | grep -v "^None"; done; done
echo "== c4mf"; for d in gpurun_out/prof_c4mf_*; do python scripts/pmc_summary.py $d k_run_mimo_ofdm_mfma 262144 2>/dev/null | grep -v "^None"; done
