# VALU / LDS / MFMA instruction counts of the two f1 kernels (with f1_mfma_kernel.patch applied); see f1_ab.sh
export TMPDIR=/tmp
for v in "grid_oversub=0" "no_mfma=1"; do
rm -rf gpurun_out/f1pmc
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/f1pmc -o f1 -- python bench.py --opt $v --config f1 --steps 2 --warmup 1 --no-cpu --pmc off > /dev/null 2>&1
python - "$v" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for p in glob.glob('gpurun_out/f1pmc/**/f1_counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(p)):
        if 'mimo_ofdm_tdl' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
per = 98304.0 * 4
print(sys.argv[1], ' '.join('%s=%.0f' % (k.replace('SQ_', ''), sum(v) / len(v) / per) for k, v in sorted(agg.items())), '(per wave and realization)')
PY
done
