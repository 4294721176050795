#!/bin/bash
# round 5, final evidence run on the product build: the default bench line (driver's command), the one-rank torchrun line, every
# profile tag of scripts/prof_r05.sh, the family / staged / operator rate scripts, then the whole GPU suite.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "default bench rc=$?"; tail -c 700 gpurun_out/bench_default.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_torchrun_1rank.log 2>&1
echo "torchrun 1-rank rc=$?"
bash scripts/prof_r05.sh > gpurun_out/prof_r05.log 2>&1
tail -3 gpurun_out/prof_r05.log
timeout 400 python scripts/bench_f64_family.py > gpurun_out/f64_family_rates.json 2> gpurun_out/f64_family.err; echo "f64 family rc=$?"
timeout 400 python scripts/bench_f64_family.py --dtype f32 > gpurun_out/f32_family_rates.json 2> gpurun_out/f32_family.err; echo "f32 family rc=$?"
timeout 300 python scripts/bench_tdl_family.py > gpurun_out/tdl_family_rates.json 2> gpurun_out/tdl_family.err; echo "tdl family rc=$?"
timeout 300 python scripts/bench_staged_c4.py > gpurun_out/staged_c4.json 2> gpurun_out/staged_c4.err; echo "staged rc=$?"
timeout 300 python scripts/bench_staged_c4.py --dtype f64 > gpurun_out/staged_c4_f64.json 2> gpurun_out/staged_c4_f64.err; echo "staged f64 rc=$?"
timeout 300 python scripts/bench_operators.py > gpurun_out/operators.json 2> gpurun_out/operators.err; echo "operators rc=$?"
timeout 300 python scripts/bench_simulators.py > gpurun_out/simulators.json 2> gpurun_out/simulators.err; echo "simulators rc=$?"
timeout 300 python scripts/bench_mimo_schemes.py > gpurun_out/mimo_schemes.json 2> gpurun_out/mimo_schemes.err; echo "schemes rc=$?"
timeout 300 python scripts/bench_ia_solvers.py > gpurun_out/ia_solvers.json 2> gpurun_out/ia_solvers.err; echo "ia rc=$?"
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
