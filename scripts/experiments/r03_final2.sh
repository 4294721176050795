#!/bin/bash
# end-of-round: full GPU suite, smoke, the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/final_gpu_suite.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_default_final.json 2> gpurun_out/bench_default_final.err
grep -E "passed|failed" gpurun_out/final_gpu_suite.log | tail -2; tail -1 gpurun_out/final_smoke.log; python -c "
import json; d=json.loads(open('gpurun_out/bench_default_final.json').read().strip().splitlines()[-1]); print(d['value'], d['rates'], d['other_workloads']['c4_staged']['frac'])"
