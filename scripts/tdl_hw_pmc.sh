#!/bin/bash
# counters of config 3's kernels at 1024 / 2048 (one launch of 2^18 realizations each): gpurun_out/tdl_pmc_<group>/
mkdir -p gpurun_out; export TMPDIR=/tmp
specs=${@:-"1024:2:f32 2048:2:f32 2048:3:f32 1024:2:f64 2048:2:f64 2048:3:f64"}
i=0
for pmc in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/tdl_pmc_$i -o t -- python scripts/tdl_one_launch.py $specs > gpurun_out/tdl_pmc_$i.log 2>&1
done
python - <<'P'
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/tdl_pmc_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_run_ofdm_tdl' not in k: continue
        k = k.split('(')[0].replace('void mcle::', '')
        rows.setdefault(k, collections.OrderedDict())
        rows[k][r['Counter_Name']] = rows[k].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k, v in rows.items():
    print(k)
    print('   ', {a: '%.4g' % b for a, b in v.items()})
    n = 1 << 18
    if 'SQ_INSTS_VALU' in v:
        print('    valu/realization %.0f  salu %.0f  lds %.0f  busy(4 cyc) %.3f  wait_any %.3f  lds_conflict/idx_active %.3f' % (
            v['SQ_INSTS_VALU'] / n, v.get('SQ_INSTS_SALU', 0) / n, v.get('SQ_INSTS_LDS', 0) / n,
            4 * v.get('SQ_ACTIVE_INST_VALU', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1), v.get('SQ_WAIT_INST_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1),
            v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
P
