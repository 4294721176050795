#!/bin/bash
# counters of config 4's kernel family per geometry (one launch of 2^16 realizations each): instructions per realization, subcarrier and stream
mkdir -p gpurun_out; export TMPDIR=/tmp
specs=${@:-"256:2:2:f32 256:2:4:f32 256:4:4:f32 512:2:2:f32 512:4:4:f32 1024:2:2:f32 1024:2:4:f32 1024:4:4:f32 2048:2:2:f32 2048:4:4:f32 512:2:2:f64 1024:2:2:f64 2048:2:2:f64 2048:4:4:f64 1024:4:4:f64"}
i=0
for pmc in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/c4_pmc_$i -o t -- python scripts/c4_one_launch.py $specs > gpurun_out/c4_pmc_$i.log 2>&1
done
python - <<'P'
import csv, glob, collections, re
rows = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/c4_pmc_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_run_mimo_ofdm' not in k: continue
        k = k.split('(')[0].replace('void mcle::', '')
        rows.setdefault(k, collections.OrderedDict())
        rows[k][r['Counter_Name']] = rows[k].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
n = 1 << 16
for k, v in rows.items():
    print(k)
    print('    per realization: valu %.0f  salu %.0f  lds %.0f  vmem_rd %.1f | GRBM cycles %.1f | 4 x active_valu / wave_cycles %.3f  wait_any %.3f  lds conflict %.3f  waves resident %.1f' % (
        v['SQ_INSTS_VALU'] / n, v.get('SQ_INSTS_SALU', 0) / n, v.get('SQ_INSTS_LDS', 0) / n, v.get('SQ_INSTS_VMEM_RD', 0) / n, v['GRBM_GUI_ACTIVE'] / n,
        4 * v.get('SQ_ACTIVE_INST_VALU', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1), v.get('SQ_WAIT_INST_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1),
        v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1), v.get('SQ_WAVE_CYCLES', 0) / max(v['GRBM_GUI_ACTIVE'], 1)))
P
