#!/bin/bash
# staged config 4 with the one-pass gen + modulate operator; multiuser Rek test; config 3 re-profile at one dispatch per step
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_staged_c4.py tests/test_gpu_multiuser.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -5
python scripts/bench_staged_c4.py --seconds 2 > gpurun_out/staged_c4.json; cat gpurun_out/staged_c4.json | python -c "import json,sys; d=json.load(sys.stdin); print('staged f32', d['realizations_per_s'], d['frac'], d['ms_per_pass'])"
python scripts/bench_staged_c4.py --seconds 2 --dtype f64 > gpurun_out/staged_c4_f64.json; cat gpurun_out/staged_c4_f64.json | python -c "import json,sys; d=json.load(sys.stdin); print('staged f64', d['realizations_per_s'], d['frac'], d['ms_per_pass'])"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_staged_c4 -o staged_c4 -- python scripts/bench_staged_c4.py --seconds 0.3 > /dev/null 2>&1
bash scripts/prof_r03.sh c3 > /dev/null 2>&1; echo profiled
