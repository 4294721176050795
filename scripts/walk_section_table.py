#!/usr/bin/env python3
"""profiles/r06/{c5,f6}_section_table.md from the section ablations of the complex128 symbol walks.

Input: gpurun_out/walk_sections.json (scripts/walk_sections.sh: the MCLE_EXPERIMENTS build of k_link_walk_f64 with ONE section
compiled out per run -- its time from bench.py's HIP events, its instruction counts from a rocprofv3 --pmc pass) and, for the
round-5 kernels the packed walk replaced, profiles/r06/walk_sections_round5_kernels.json (the same ablations inside k_ia_link<double> /
k_bd_link<double>, taken before they were retired to option walk_legacy=1).

A section's cost = whole kernel minus the run without it (wave-instructions per realization, ms per launch); `units` = how many
of the section's unit of work a WAVEFRONT issues per realization (a lane pair = two columns; config 5: 100 pairs per realization =
1.5625 passes of 64 lanes; f6: 250 pairs = 3.906 passes), `per unit` = instructions per unit, `minimum` = what the ISA needs for
that unit (stated per row), `x min` their ratio.  The ablated run keeps every value alive with a few substitute instructions
(`sub`), added back here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# per configuration: pairs per realization, noise blocks / Box-Muller samples / decisions / complex MACs per lane pair, streams
SHAPE = {
    "c5": dict(pairs=100, blocks=6, bm=12, decisions=6, macs=30, streams=3, philox_sym=1,
               title="config 5 (K = 3, 2 x 2 closed-form IA, 16-QAM, 200 columns, 20 dB), complex128, slicer, 262 144 realizations per launch"),
    "f6": dict(pairs=250, blocks=6, bm=12, decisions=12, macs=36, streams=6, philox_sym=1.25,
               title="f6 (block diagonalisation, K = 3 cells of 2 x 2, the reference's PSK(4), 500 columns, 15 dB), complex128, 131 072 realizations per launch"),
}
# (key of the ablated run, section, unit name, units per lane pair, minimum instructions per unit, substitute instructions per lane pair, note)
ROWS = [
    ("1", "symbol draws", "pass", None, None, 0,
     "one Philox call serves the pass's 2 S runs of 9 DATA blocks (two calls for S = 6 when a pass straddles two realizations: a quarter "
     "of f6's passes); per stream an address (3), a 16-bit LDS read, two mask / shift pairs"),
    ("2", "noise Philox", "block", "blocks", 40, 4,
     "10 rounds x (2 v_mad_u64_u32 + 2 v_bitop3_b32); the round keys are scalar"),
    ("4", "Box-Muller", "sample", "bm", 59, 4,
     "bm_f64.hpp: 47 f64 + 12 integer instructions per complex normal (table + degree-7 log1p, rsq + Newton + two corrections, table + "
     "degree-7 / 6 sincos): the design count of round 3, reached"),
    ("8", "estimates", "complex MAC", "macs", 4, None,
     "four chained v_fma_f64 per complex multiply-add; the coefficient is one ds_read_b128 per pair of columns.  (f6: the f64 count of the "
     "pass IS the minimum -- 708 static f64 instructions against 12 x 47 + 144 + 24; the surplus is v_mov_b32 pairs that line the "
     "128-bit LDS results up as operand pairs, 1.9 per MAC)"),
    ("16", "decisions + counts", "decision", "decisions", None, None, ""),
]
DEC_MIN = {"c5": (24, "slicer: 2 x (2 fma, max, min, floor, cvt) + pack + 6 (both Gray decodes in one register) + 3 (label) + xor, "
                      "compare, add-carry, bcnt"),
           "f6": (14, "axis certificate: 2 adds, 4 compares (|.| as modifiers), 2 selects, or, bfe + xor, compare, add-carry, bcnt")}


def table(cfg, data, legacy):
    sh = SHAPE[cfg]
    whole = data[cfg + "_0"]
    pr0 = whole["per_realization"]
    passes = sh["pairs"] / 64.0
    out = ["# Section table: %s" % sh["title"], "",
           "`k_link_walk<double, ...>` (csrc/walk_f64.hpp; workgroups of four wavefronts), MCLE_EXPERIMENTS build, `scripts/walk_sections.sh` -> `scripts/walk_section_table.py`.",
           "Whole kernel (solve + walk, HIP events): **%.3f ms per launch = %.3e realizations/s**; %d VALU, %d LDS, %d SALU wave-instructions "
           "per realization (%.0f VALU per pass of 64 lane pairs, %.4g passes per realization)." % (
               whole["kernel_ms_per_launch"], whole["realizations_per_s"], round(pr0["SQ_INSTS_VALU"]), round(pr0["SQ_INSTS_LDS"]),
               round(pr0["SQ_INSTS_SALU"]), pr0["SQ_INSTS_VALU"] / passes, passes), ""]
    out += ["| section | VALU / realization | share | ms | units / realization | VALU per unit | minimum | x min | what the unit is |",
            "|---|---|---|---|---|---|---|---|---|"]
    total = 0.0
    for key, name, unit, ukey, umin, sub, note in ROWS:
        run = data["%s_%s" % (cfg, key)]
        d_valu = pr0["SQ_INSTS_VALU"] - run["per_realization"]["SQ_INSTS_VALU"]
        d_ms = whole["kernel_ms_per_launch"] - run["kernel_ms_per_launch"]
        if key == "16":
            umin, note = DEC_MIN[cfg]
            sub = 2.0                                                           # two adds, a compare and an add stand in per PAIR of decisions
        if key == "8":
            sub = 4.0 * sh["decisions"] / sh["macs"]                            # two complex adds per estimate stand in
        if ukey is None:
            units = passes
            per_unit = d_valu / units
            umin_s, ratio = "see note", ""
        else:
            units = sh[ukey] * passes
            per_unit = d_valu / units + (sub or 0)
            umin_s, ratio = str(umin), "%.2f" % (per_unit / umin)
        total += d_valu
        out.append("| %s | %d | %.0f %% | %.3f | %.1f %s | %.1f | %s | %s | %s |" % (
            name, round(d_valu), 100.0 * d_valu / pr0["SQ_INSTS_VALU"], d_ms, units, unit + ("es" if unit.endswith("s") else "s"), per_unit,
            umin_s, ratio, note))
    sk = data[cfg + "_31"]
    out.append("| everything above compiled out | %d | %.0f %% | %.3f (the run itself) | | | | | chunk set-up (records to LDS), lane -> pair "
               "bookkeeping, two masked DPP wave sums and the LDS-atomic hand-over per pass, accounting, ONE flush of the counters per four wavefronts (with one per wavefront this run took 0.60 ms: `walk_grid_sweep.log`); includes the substitutes of all five ablations |" % (
                   round(sk["per_realization"]["SQ_INSTS_VALU"]), 100.0 * sk["per_realization"]["SQ_INSTS_VALU"] / pr0["SQ_INSTS_VALU"],
                   sk["kernel_ms_per_launch"]))
    busy = 4.0 * pr0["SQ_ACTIVE_INST_VALU"] / max(1.0, pr0["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    out += ["", "Sections sum to %d of the %d VALU instructions (the rest: the skeleton).  `4 x SQ_ACTIVE_INST_VALU / SIMD-cycles` of the whole "
            "kernel: **%.2f** -- the walk issues on more than nine cycles in ten, so its time IS its instruction count; the sum of the "
            "sections' times exceeds the kernel's because a compiled-out section also frees issue slots the other wavefronts were waiting for."
            % (round(total), round(pr0["SQ_INSTS_VALU"]), busy), ""]
    if legacy is not None and cfg + "_0" in legacy:
        lw = legacy[cfg + "_0"]
        lp = lw["per_realization"]
        out += ["## The round-5 kernel it replaced (`%s<double>`, option walk_legacy = 1)" % ("k_ia_link" if cfg == "c5" else "k_bd_link"), "",
                "%.3f ms per launch, %d VALU / %d SALU / %d LDS wave-instructions per realization (ablations of the same five sections inside "
                "that kernel: `profiles/r06/walk_sections_round5_kernels.json`).  Where the %d instructions went:" % (
                    lw["kernel_ms_per_launch"], round(lp["SQ_INSTS_VALU"]), round(lp["SQ_INSTS_SALU"]), round(lp["SQ_INSTS_LDS"]),
                    round(lp["SQ_INSTS_VALU"] - pr0["SQ_INSTS_VALU"])), ""]
        out += LEGACY_NOTES[cfg]
    return "\n".join(out) + "\n"


LEGACY_NOTES = {
    "c5": ["* **22 % idle lanes**: 200 columns = a pass of 64 lane pairs and a pass of 36 per realization; the packed walk runs 25 full "
           "passes per 16 realizations instead of 32.",
           "* **17 % `v_readlane_b32`** (static count 1 490 of 8 600 vector instructions): the record G[3][3], U[3][2] as sixty scalar registers "
           "next to the Philox keys and the Box-Muller constants -- 178 spilled SGPRs, every use behind a read-back.  The packed walk reads "
           "a coefficient from the chunk's LDS copy at its use (18 spilled SGPRs, 21 read-backs in the whole kernel).",
           "* **decisions at ~75 instructions each**: a run-time `demod_one` per decision -- method switch, certificate switch, candidate-grid "
           "search and sweep inlined six times, the spilled modem parameters re-read around every branch.  Compile-time decision form: 24.",
           "* symbol draws: four `ds_bpermute_b32` and a select tree per stream -> the pass's DATA blocks through LDS, one 16-bit read per stream."],
    "f6": ["* **decisions at ~85 instructions each, 47 % of the kernel**: the reference's PSK(4) = exp(j 2 pi m / 4) sits ON the axes, so the "
           "quadrant certificate of round 5 never applied to it and every decision went through the generic path (candidate-grid cell, "
           "1 - 2 table entries, the literal metric) behind the run-time switches of `demod_one`; 2 322 SALU instructions per realization "
           "were its loop control.  The packed walk compiles the on-axis certificate (`demod_axis4_cert`: signs of re - im and re + im, "
           "margin 2^-30 a): 16 per decision.",
           "* **20 % `v_readlane_b32`** (2 426 of 11 847 static): the record d[6], W[6][2] = 72 scalar registers.  -> LDS, as for config 5.",
           "* symbol draws 12.6 % (six streams x four `ds_bpermute_b32` + selects per pass) -> 8 %.",
           "* 250 lane pairs per realization fill 3.9 of 4 passes: the packing itself buys f6 2 %."],
}


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "walk_sections.json")
    data = json.load(open(src))
    lpath = os.path.join(ROOT, "profiles", "r06", "walk_sections_round5_kernels.json")
    legacy = json.load(open(lpath)) if os.path.exists(lpath) else None
    for cfg in ("c5", "f6"):
        path = os.path.join(ROOT, "profiles", "r06", "%s_section_table.md" % cfg)
        open(path, "w").write(table(cfg, data, legacy))
        print("wrote", path)
    json.dump(data, open(os.path.join(ROOT, "profiles", "r06", "walk_sections.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
