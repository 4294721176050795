#!/bin/bash
# round 4, call 12: the driver's default line (N = 1) and its launch line with one rank (native exchange + strong leg)
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "default bench rc=$? in $(( $(date +%s) - t0 )) s"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value %.4g' % d['value'], 'frac %.4f' % r['frac'], 'valu_busy', r['valu_busy_chip'], 'rates', d['rates'])
print('ser_err', d.get('ser_abs_err_vs_oracle'), 'cpu', d['cpu_baseline']['value'], 'allcores', d.get('cpu_baseline_all_cores',{}).get('value'), d.get('cpu_baseline_all_cores',{}).get('cores'))
ow=d['other_workloads']
for k in ('c4_staged','c4_staged_f64'):
    s=ow[k]; print(k, 'rate %.4g' % s['realizations_per_s'], 'frac %.3f' % s['frac'], 'minrule %.3f' % s['frac_min_rule'], 'measured', s['measured_hbm_bytes_per_realization'], 'copy', s['copy_GBps_measured_this_run'])
for c in ('c2','c3','c5','f1','f6'):
    print(c, {dt: '%.4g' % ow[c][dt]['realizations_per_s'] for dt in ('f64','f32')})
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --pmc off > gpurun_out/bench_torchrun_1rank_native.log 2>&1
echo "torchrun rc=$?"; tail -n 1 gpurun_out/bench_torchrun_1rank_native.log | cut -c1-300
