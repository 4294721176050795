#!/bin/bash
# round 3: everything profiles/r03 holds, in one GPU call
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/prof_r03.sh
hipcc -O2 --offload-arch=gfx950 scripts/experiments/f64_rates.hip -o /tmp/f64_rates 2>/dev/null && /tmp/f64_rates > gpurun_out/f64_rates.txt 2>&1
python scripts/bench_staged_c4.py --batch 8192 --seconds 2 > gpurun_out/staged_c4.json 2>/dev/null
python scripts/bench_staged_c4.py --batch 8192 --seconds 2 --dtype f64 > gpurun_out/staged_c4_f64.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_staged_c4 -o staged -- python scripts/bench_staged_c4.py --batch 8192 --seconds 0.3 > /dev/null 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_torchrun_1rank.json 2> gpurun_out/bench_torchrun_1rank.err
timeout 300 python scripts/bench_operators.py > gpurun_out/operators.json 2> gpurun_out/operators.err
timeout 300 python scripts/bench_operators.py --dtype=f64 > gpurun_out/operators_f64.json 2> gpurun_out/operators_f64.err
timeout 300 python scripts/bench_simulators.py > gpurun_out/simulators.json 2> gpurun_out/simulators.err
timeout 300 python scripts/bench_mimo_schemes.py > gpurun_out/mimo_schemes.json 2> gpurun_out/mimo_schemes.err
timeout 300 python scripts/bench_ia_solvers.py > gpurun_out/ia_solvers.json 2> gpurun_out/ia_solvers.err
ls gpurun_out | head -80
