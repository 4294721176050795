#!/bin/bash
# round 4, call 4: bench.py's native exchange / strong leg / staged f64 + byte counters
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_bench_contract.py tests/test_gpu_staged_c4.py -m gpu -q --timeout=900 2>&1 | tail -15
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --pmc off > gpurun_out/bench_torchrun_1rank_native.log 2>&1
echo "torchrun rc=$?"; tail -n 1 gpurun_out/bench_torchrun_1rank_native.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['rccl']['exchange_impl'][:60], json.dumps(d['strong'])[:700])"
for dt in f32 f64; do
  timeout 600 python scripts/bench_staged_c4.py --counters --dtype $dt > gpurun_out/staged_c4_$dt.json 2> gpurun_out/staged_c4_$dt.err
  python -c "
import json
d=json.load(open('gpurun_out/staged_c4_$dt.json'))
print('$dt', 'rate %.4g' % d['realizations_per_s'], 'frac %.3f' % d['frac'], 'min-rule %.3f' % d['frac_min_rule'], 'measured B %s' % d['measured_hbm_bytes_per_realization'], 'balg', d['b_alg_bytes_per_realization'], 'copy', d['copy_GBps_measured_this_run'], 'of copy', d['frac_of_copy_bw_measured_this_run'])"
done
