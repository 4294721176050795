#!/bin/bash
# round 6: section ablation of the part-wave config-4 kernel (pipeline_mimo_pw.hip, NW = 4; MCLE_EXPERIMENTS build, scripts/build_exp.sh): per variant the kernel
# time and the dynamic VALU / LDS / SALU instruction counts -> gpurun_out/pw_sections.json
export TMPDIR=/tmp
export MCLE_LIBRARY=$PWD/scripts/experiments/bin/libmcle_exp.so
mkdir -p gpurun_out
rm -rf /tmp/psec && mkdir -p /tmp/psec
ARGS="--no-cpu --pmc off --single-demod --demod mindist --dtype f64 --config c4 --batch 262144 --opt f64_threads=263"
for v in 0 32 64 128 256 512 1024 2016; do
  timeout 300 python bench.py --steps 8 --warmup 2 $ARGS --opt f64_variant=$v 2>/dev/null | tail -1 > /tmp/psec/time_$v.json
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/psec/pmc_$v -o pmc -- python bench.py --steps 3 --warmup 1 --preroll-ms 0 $ARGS --opt f64_variant=$v > /dev/null 2>&1
done
python - <<'PY'
import json, csv, glob
out = {}
for v in (0, 32, 64, 128, 256, 512, 1024, 2016):
    d = json.loads(open("/tmp/psec/time_%d.json" % v).read())
    row = {"kernel_ms_per_launch": d["roofline"]["kernel_ms_per_launch"], "realizations_per_s": d["value"]}
    agg = {}
    for p in glob.glob("/tmp/psec/pmc_%d/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(p)):
            if "k_run_mimo_ofdm_pw<" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    n = 4 * 262144.0
    row["per_realization"] = {k: val / n for k, val in agg.items()}
    out[str(v)] = row
    pr = row["per_realization"]
    print(v, "%.3f ms" % row["kernel_ms_per_launch"], "valu %.0f lds %.0f salu %.0f" % (pr.get("SQ_INSTS_VALU", 0), pr.get("SQ_INSTS_LDS", 0), pr.get("SQ_INSTS_SALU", 0)),
          "busy %.3f" % (4 * pr.get("SQ_ACTIVE_INST_VALU", 0) / (pr.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024)))
json.dump(out, open("gpurun_out/pw_sections.json", "w"), indent=1)
PY
