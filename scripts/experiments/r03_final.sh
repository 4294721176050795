#!/bin/bash
# end-of-round artefacts: the full GPU suite, the default bench line, the smoke entry, and fresh profiles of the kernels that
# changed after the first profile pass
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -5 > gpurun_out/final_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_default_final.json 2> gpurun_out/bench_default_final.err
bash scripts/prof_r03.sh c4_f64 c4_f64sl c3_f64 c2_f64 f1_f64 c5_f64 f6_f64 > gpurun_out/final_prof.log 2>&1
tail -3 gpurun_out/final_gpu_suite.log; cat gpurun_out/final_smoke.log | tail -2; tail -c 600 gpurun_out/bench_default_final.json
