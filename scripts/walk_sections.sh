#!/bin/bash
# round 6: section ablation of the packed complex128 symbol walk k_link_walk_f64 (csrc/walk_f64.hpp; config 5 and f6) at the bench legs' batches
# (MCLE_EXPERIMENTS build, scripts/build_exp.sh; option f64_variant: 1 = no symbol draws, 2 = no noise Philox, 4 = no Box-Muller,
# 8 = no estimate arithmetic, 16 = no decisions, 31 = all of them) -> gpurun_out/walk_sections.json
export TMPDIR=/tmp
export MCLE_LIBRARY=$PWD/scripts/experiments/bin/libmcle_exp.so
mkdir -p gpurun_out; rm -rf /tmp/wsec && mkdir -p /tmp/wsec
for cfg in c5 f6; do
  ARGS="$(python bench.py --profile-spec ${cfg}_f64) --no-cpu --pmc off --single-demod"
  for v in 0 1 2 4 6 8 16 31; do
    timeout 300 python bench.py --steps 8 --warmup 2 $ARGS --opt f64_variant=$v 2>/dev/null | tail -1 > /tmp/wsec/time_${cfg}_$v.json
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/wsec/pmc_${cfg}_$v -o pmc -- python bench.py --steps 3 --warmup 1 --preroll-ms 0 $ARGS --opt f64_variant=$v > /dev/null 2>&1
  done
done
python - <<'PY'
import json, csv, glob, subprocess, sys
out = {}
for cfg, needle in (("c5", "k_link_walk<double"), ("f6", "k_link_walk<double")):
    batch = int(subprocess.run([sys.executable, "bench.py", "--profile-spec", cfg + "_f64"], capture_output=True, text=True).stdout.split("--batch")[1].split()[0])
    for v in (0, 1, 2, 4, 6, 8, 16, 31):
        d = json.loads(open("/tmp/wsec/time_%s_%d.json" % (cfg, v)).read())
        row = {"kernel_ms_per_launch": d["roofline"]["kernel_ms_per_launch"], "realizations_per_s": d["value"], "batch": batch}
        agg = {}
        for p in glob.glob("/tmp/wsec/pmc_%s_%d/**/*counter_collection.csv" % (cfg, v), recursive=True):
            for r in csv.DictReader(open(p)):
                if needle in r["Kernel_Name"]:
                    agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n = 4.0 * batch
        row["per_realization"] = {k: val / n for k, val in agg.items()}
        out["%s_%d" % (cfg, v)] = row
        pr = row["per_realization"]
        print(cfg, v, "%.3f ms" % row["kernel_ms_per_launch"], "valu %.0f lds %.0f salu %.0f" % (pr.get("SQ_INSTS_VALU", 0), pr.get("SQ_INSTS_LDS", 0), pr.get("SQ_INSTS_SALU", 0)),
              "busy %.3f" % (4 * pr.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, pr.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024)))
json.dump(out, open("gpurun_out/walk_sections.json", "w"), indent=1)
PY
