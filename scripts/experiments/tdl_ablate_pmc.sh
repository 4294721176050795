#!/bin/bash
# needs the stage-ablation knob: git apply scripts/experiments/tdl_ablate.patch && make -C pyphysim_amd/csrc (results are
# meaningless with stages off; the knob is not in the product kernel).  Bits: 1 rays, 2 tap polynomials (both only after the
# first pass), 4 noise, 8 taps 1.., 16 equaliser taps 1.., 32 equalise + demodulate.
# dynamic VALU / LDS instruction counts of the config-3 matrix-core kernel with stages switched off (MCLE_TDL_ABLATE)
export TMPDIR=/tmp; mkdir -p gpurun_out
for ab in ${ABLATES:-0 1 2 4 8 16 32 63}; do
  rm -rf gpurun_out/abl_$ab
  MCLE_TDL_ABLATE=$ab timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/abl_$ab -o abl -- python bench.py --config c3 --steps 2 --warmup 1 --no-cpu --pmc off > /dev/null 2>&1
  python - $ab <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for p in glob.glob('gpurun_out/abl_%s/**/abl_counter_collection.csv' % sys.argv[1], recursive=True):
    for row in csv.DictReader(open(p)):
        if 'tdl' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
per = 32768.0 * 4
print('ablate', sys.argv[1], ' '.join('%s=%.0f' % (k.replace('SQ_', ''), sum(v) / len(v) / per) for k, v in sorted(agg.items())), '(per wave and pass)')
PY
done
