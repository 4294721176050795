#!/bin/bash
# round 4, call 26: the planar family is the complex64 default (option f32_mfma = 1 keeps the matrix-core kernel): whole suite,
# then the driver-style bench lines of the final build
export TMPDIR=/tmp
mkdir -p gpurun_out
true
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
echo "bench rc=$?"; tail -n 1 gpurun_out/bench_default.log > gpurun_out/bench_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read())
r=d['roofline']; print('value %.4g' % d['value'], 'frac %.4f' % r['frac'], 'valu_busy', r['valu_busy_chip'], 'rates', d['rates'])
r32=d['roofline_f32']; print('f32 kernel', r32['kernel'], 'frac %.4f' % r32['frac'], 'valu_busy', r32['valu_busy_chip'], 'insts', r32['valu_wave_insts_per_realization'])
print('ser_err', d.get('ser_abs_err_vs_oracle'), 'cpu', d['cpu_baseline']['value'], 'allcores', d.get('cpu_baseline_all_cores',{}).get('value'), d.get('cpu_baseline_all_cores',{}).get('cores'))
ow=d['other_workloads']
for k in ('c4_staged','c4_staged_f64'):
    s=ow[k]; print(k, 'rate %.4g' % s['realizations_per_s'], 'frac %.3f' % s['frac'], 'measured', s['measured_hbm_bytes_per_realization'], 'copy', s['copy_GBps_measured_this_run'])
for c in ('c2','c3','c5','f1','f6'):
    print(c, {dt: '%.4g' % ow[c][dt]['realizations_per_s'] for dt in ('f64','f32')})
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --pmc off > gpurun_out/bench_torchrun_1rank_native.log 2>&1
echo "torchrun rc=$?"; tail -n 1 gpurun_out/bench_torchrun_1rank_native.log | cut -c1-300
