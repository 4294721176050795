#!/usr/bin/env python3
"""Condense gpurun_out/ (rocprofv3 csv + bench json) into the tracked profiles/<round>/ summaries.

usage: python scripts/collect_profiles.py r02
For every profiled config (c4, c3, c2, f1, c5, f6) writes <cfg>_kernel_stats.csv and <cfg>_pmc_summary.json
(round 5: counter SUMS over every dispatch of the dominant kernel in the counter run + the realizations those dispatches
cover (`_realizations_total` = (warm-up + timed steps) x batch) + the `_derived` fractions of bench.derive_pmc, the same
function bench.py applies to its live counters -- a step that the pipeline cuts into several dispatches is normalised by what
it processed, not by a per-launch mean; `_dispatches`, `_grid_sizes` say how a step was cut): hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, following
/opt/skills/guides/MI355X_MICROARCH.md (rocprofv3 counts KiB; on gfx950 FETCH_SIZE reports half of
a coalesced stream's bytes, WRITE_SIZE is uncalibrated); valu_busy_chip = 4 * SQ_ACTIVE_INST_VALU /
(GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); mfma_busy_chip = SQ_VALU_MFMA_BUSY_CYCLES / the same SIMD-cycles.
The `_dispatch` record is rocprofv3's own (its VGPR_Count is half the real allocation on gfx950 and its
LDS_Block_Size omits dynamic LDS -- the compiler's numbers are in kernel_resources.json, scripts/kernel_resources.py).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
sys.path.insert(0, REPO)
from bench import derive_pmc  # noqa: E402
src = os.path.join(REPO, "gpurun_out")
dst = os.path.join(REPO, "profiles", rnd)
os.makedirs(dst, exist_ok=True)

# tag -> (kernel-name needle of the dominant kernel, label); the bench arguments of a tag (config, dtype, demodulator, batch)
# are recorded by scripts/prof_r04.sh (prof_r03.sh in round 3) in gpurun_out/prof_<tag>_meta.json
CONFIGS = {
    "c4_f64": ("k_run_mimo_ofdm_planar<", "k_run_mimo_ofdm_planar<1024, 4, 4, 2, 4> (complex128, FFT 1024, 4x4, 512 threads), min-distance demodulator (margin certificate): the bench.py headline"),
    "c4_f64sl": ("k_run_mimo_ofdm_planar<", "k_run_mimo_ofdm_planar<1024, 4, 4, 2, 4> with the QAM slicer"),
    "c4": ("k_run_mimo_ofdm_planar<", "k_run_mimo_ofdm_planar<float, 1024, 4, 4, 4, 4, 4> (complex64, FFT 1024, 4x4: radix-16 passes, one transform per wavefront), QAM slicer"),
    "c4md": ("k_run_mimo_ofdm_planar<", "k_run_mimo_ofdm_planar<float, 1024, 4, 4, 4, 4, 4> with the min-distance demodulator (margin certificate)"),
    "c4_mfma": ("k_run_mimo_ofdm_mfma<", "k_run_mimo_ofdm_mfma<3,6> (complex64, FFT 1024, 4x4 on the matrix cores: option f32_mfma = 1, the default of rounds 2-3), QAM slicer"),
    "c4md_mfma": ("k_run_mimo_ofdm_mfma<", "k_run_mimo_ofdm_mfma<3,6> with the min-distance demodulator"),
    "f1": ("k_run_mimo_ofdm_tdl_wave<", "k_run_mimo_ofdm_tdl_wave<float, 1024, 4, 4, 2, 2, 3> (complex64, one receive antenna per wavefront; bench.py --config f1; k_mimo_tdl_symbol_polys<float, true> runs before it)"),
    "f1_coop": ("k_run_mimo_ofdm_tdl<", "k_run_mimo_ofdm_tdl<float,1024,4> (the workgroup-cooperative kernel of rounds 1-4: option mimo_tdl_kernel=1)"),
    "c3": ("k_run_ofdm_tdl_wave<", "k_run_ofdm_tdl_wave<float, 1024, 2, 3> (complex64, FFT 1024, one realization per wavefront; k_tdl_symbol_polys runs before it)"),
    "c3_f64": ("k_run_ofdm_tdl_wave<", "k_run_ofdm_tdl_wave<double, 1024, 5, 2> (complex128, one realization per wavefront)"),
    "c3_mfma": ("k_run_ofdm_tdl_mfma<", "k_run_ofdm_tdl_mfma<2> (complex64 on the matrix cores, 4 realizations per pass: option tdl_kernel=1, the default of rounds 2-3)"),
    "c3_f64_batch": ("k_run_ofdm_tdl_batch<", "k_run_ofdm_tdl_batch<double,1024,2> (complex128, two realizations per pass: option tdl_kernel=1)"),
    "f1_f64": ("k_run_mimo_ofdm_tdl_wave<", "k_run_mimo_ofdm_tdl_wave<double, 1024, 4, 4, 5, 2, 2> (complex128, one receive antenna per wavefront)"),
    "f1_f64_coop": ("k_run_mimo_ofdm_tdl<", "k_run_mimo_ofdm_tdl<double,1024,4> (cooperative kernel, option mimo_tdl_kernel=1)"),
    "c5_f64": ("k_ia_link<", "k_ia_link<double> (complex128 symbol walk)"),
    "f6_f64": ("k_bd_link<", "k_bd_link<double,2> (complex128 symbol walk)"),
    "c2": ("k_run_flat_mfma<", "k_run_flat_mfma<8,2> (complex64, 8 Jakes rays on the matrix cores)"),
    "c2_f64": ("k_run_flat<", "k_run_flat<double,8,0> (complex128, 16-symbol rotation recurrence of the ray phasors)"),
    "c5": ("k_ia_link<", "k_ia_link<float> (the symbol walk; k_ia_solve_links<float,false> runs before it, see c5_kernel_stats.csv)"),
    "f6": ("k_bd_link<", "k_bd_link<float,2> (the symbol walk; k_bd_solve_links_static<float,3,2> runs before it, see f6_kernel_stats.csv)"),
}


if rnd >= "r06":
    # round 6: ONE table for the profile driver and for this script (bench.py: profile_specs)
    import bench as _bench
    CONFIGS = {tag: (sp["kernel"], "%s; bench.py --config %s --dtype %s --demod %s --batch %d%s%s" % (
        sp["kernel"], sp["config"], sp["dtype"], sp["demod"], sp["batch"], "".join(" --opt " + o for o in sp["opts"]),
        ("; leg " + sp["leg"]) if sp["leg"] else "")) for tag, sp in _bench.profile_specs().items()}


def first(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


for cfg, (needle, label) in CONFIGS.items():
    meta_path = os.path.join(src, "prof_%s_meta.json" % cfg)
    if not os.path.exists(meta_path):
        continue
    bench_args = json.load(open(meta_path))["bench_args"]
    per_launch = int(bench_args.split("--batch")[1].split()[0])
    stats = first(os.path.join(src, "prof_%s_stats" % cfg, "**", "%s_kernel_stats.csv" % cfg))
    if stats:
        shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % cfg))
    summary, meta = {}, {}
    grids = collections.Counter()
    for path in sorted(glob.glob(os.path.join(src, "prof_%s_*" % cfg, "**", "%s_counter_collection.csv" % cfg),
                                 recursive=True)):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(path)):
            if needle in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
                meta = {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                            "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
                meta["Kernel_Name"] = row["Kernel_Name"].split("(")[0]
                if row["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE"):
                    grids[row["Grid_Size"]] += 1
        for name, vals in agg.items():
            summary[name] = {"launches": len(vals), "sum": sum(vals), "mean_per_launch": sum(vals) / len(vals), "min": min(vals),
                             "max": max(vals)}
    if not summary:
        continue
    # the counter runs are `--warmup 1 --steps 3` without pre-roll (scripts/prof_r05.sh): four steps of `per_launch` realizations
    steps = 4
    total = steps * per_launch
    derived = derive_pmc({k: v["sum"] for k, v in summary.items()}, total)
    dispatches = max(v["launches"] for v in summary.values())
    summary["_derived"] = derived
    summary["_kernel"] = meta.get("Kernel_Name", label)
    summary["_kernel_note"] = label
    summary["_bench_args"] = bench_args
    summary["_realizations_per_step"] = per_launch
    summary["_realizations_total"] = total
    summary["_dispatches"] = dispatches
    summary["_dispatches_per_step"] = dispatches / float(steps)
    summary["_grid_sizes"] = dict(grids)
    summary["_dispatch"] = meta
    summary["_dispatch_note"] = ("rocprofv3's record: VGPR_Count is HALF the allocation on gfx950, LDS_Block_Size omits "
                                 "dynamic LDS; see kernel_resources.json for the code-object values")
    json.dump(summary, open(os.path.join(dst, "%s_pmc_summary.json" % cfg), "w"), indent=1)
    print(cfg, json.dumps(derived), json.dumps(meta))

for name in ("f64_rates.txt", "staged_c4.json", "staged_c4_f64.json", "bench_default.json", "bench_torchrun_1rank.json",
             "f64_ablation.txt"):
    pth = os.path.join(src, name)
    if os.path.exists(pth):
        shutil.copy(pth, os.path.join(dst, name))
st = first(os.path.join(src, "prof_staged_c4", "**", "staged*_kernel_stats.csv"))
if st:
    shutil.copy(st, os.path.join(dst, "staged_c4_kernel_stats.csv"))

ops = first(os.path.join(src, "prof_operators_stats", "**", "operators_kernel_stats.csv"))
if ops:
    shutil.copy(ops, os.path.join(dst, "operators_kernel_stats.csv"))
staged = first(os.path.join(src, "prof_f1staged_stats", "**", "f1staged_kernel_stats.csv"))
if staged:
    shutil.copy(staged, os.path.join(dst, "staged_f1_kernel_stats.csv"))
sj = os.path.join(src, "staged_f1.json")
if os.path.exists(sj):
    lines = [l for l in open(sj).read().strip().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(dst, "staged_f1.json"), "w").write(lines[-1] + "\n")

for name, out_name in (("ia_solvers", "ia_solvers"), ("mimo_schemes", "mimo_schemes"), ("operators", "operators_hbm"),
                       ("simulators", "simulators")):
    pj = os.path.join(src, name + ".json")
    if os.path.exists(pj):
        text = open(pj).read()
        try:
            doc = json.loads(text[text.index("{"):])
        except ValueError:
            continue
        json.dump(doc, open(os.path.join(dst, out_name + ".json"), "w"), indent=1)

for path in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(dst, os.path.basename(path)), "w").write(lines[-1] + "\n")
print("wrote", sorted(os.listdir(dst)))
