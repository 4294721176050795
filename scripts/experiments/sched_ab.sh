#!/bin/bash
# A/B of the backend's scheduling strategy for the three fused OFDM kernels: libmcle.so as built by the Makefile against a
# copy whose pipeline_{siso_tdl,mimo_mfma,mimo_tdl}.hip were compiled with -mllvm -amdgpu-sched-strategy=max-ilp
# (scripts/experiments/libmcle_ilp.so, built by hand next to it).  Runs on the GPU box's scratch copy of the repo.
cd "$(dirname "$0")/../.."
run() { for c in c4 c3 f1; do python bench.py --config $c --steps 10 --warmup 2 --no-cpu --pmc off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $c %.4g realizations/s, kernel %.3f ms, ser %.6f' % (d['value'], d['roofline']['kernel_ms_per_launch'], d['ser']))"; done; }
run default
cp pyphysim_amd/csrc/libmcle.so /tmp/libmcle_default.so
cp scripts/experiments/libmcle_ilp.so pyphysim_amd/csrc/libmcle.so
run max-ilp
cp /tmp/libmcle_default.so pyphysim_amd/csrc/libmcle.so
run default
