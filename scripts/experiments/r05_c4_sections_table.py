#!/usr/bin/env python3
"""round 5: the instruction-level account of the complex128 headline kernel (k_run_mimo_ofdm_planar<double, 1024, 4, 4, 4, 2, 12>)
from its section ablation (scripts/experiments/r05_c4_sections.sh -> profiles/r05/c4_f64_sections.json: kernel time and DYNAMIC
instruction counters of the kernel with one section compiled out at a time, MCLE_EXPERIMENTS build).
Per section: instructions it issues, the cycles it costs, cycles per instruction against the issue cost of its instruction mix
(profiles/r03/f64_rates.txt, profiles/r04/f32_rates.txt at two wavefronts per SIMD), and the instructions its arithmetic needs at
least.  Writes profiles/r05/c4_f64_section_table.md."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = json.load(open(os.path.join(REPO, "profiles", "r05", "c4_f64_sections.json")))
N = 262144
CYC_PER_MS = 2.4e9 * 256 / N / 1e3          # cycles per realization and CU for one millisecond of launch time
base = d["0"]
b_ms, b_valu = base["kernel_ms_per_launch"], base["per_realization"]["SQ_INSTS_VALU"]
# issue cost of the dominant instruction classes at two wavefronts per SIMD: f64 add / mul / fma 2.13-2.26 ns = 5.1-5.4 cycles;
# v_mad_u64_u32 4.96, v_bitop3 / 32-bit logic 2.7-2.9, three-operand integer forms and conversions 4.5-4.7
SECTIONS = [
    # key, name, issue cost of the mix (cycles / instruction), minimum instructions per realization, how the minimum is counted
    ("32", "scatter: symbol draws + table look-ups", 4.6,
     256 * 62 / 64.0 + 4096 * 1 / 64.0, "256 Philox blocks x 62 instructions (10 rounds x (2 v_mad_u64_u32 + 2 v_bitop3) + counter set-up) + one byte extraction per symbol"),
    ("64", "transmit transform, passes A + B (4 antennas)", 5.2,
     4 * 256 * 4 * 28 / 64.0, "per antenna 4 radix-4 layers x 256 butterflies x (16 real adds + 3 complex products of 4) = 28 672 real operations"),
    ("128", "noise: 2 048 Philox blocks + 4 096 Box-Muller samples (measured with the H x products out: variants 256 -> 384)", 4.35,
     2048 * 62 / 64.0 + 4096 * 59 / 64.0, "2 048 blocks x 62 + 4 096 samples x (47 f64 + 12 integer instructions, bm_f64.hpp)"),
    ("256", "H x products (16 complex multiply-adds per position)", 5.2,
     1024 * 16 * 4 / 64.0, "1 024 positions x 4 x 4 complex multiply-adds x 4 FMAs"),
    ("512", "receive transform, passes B' + A' (4 antennas)", 5.2,
     4 * 256 * 4 * 28 / 64.0, "as the transmit transform"),
    ("1024", "decode: G y, certificate, labels, count (4 096 symbols)", 4.7,
     4096 * (16 + 22 + 6) / 64.0, "per symbol 4 complex multiply-adds (16 FMAs) + the margin certificate (22: 2 x (fma, max, min, rint, sub, compare, convert) + 8 packed Gray decode) + xor / compare / popcount / adds (6)"),
]
rows, tot_ms, tot_valu, tot_min = [], 0.0, 0.0, 0.0
for key, name, cost, vmin, how in SECTIONS:
    v = d[key]
    ref_ms, ref_valu = b_ms, b_valu
    if key == "128":
        # the noise is measured next to the H x products compiled out (variants 256 and 384 = 128 + 256): with the noise ALONE
        # compiled out the H x loop no longer has its draws to hide the LDS reads behind and the kernel gets slower, not faster
        ref_ms, ref_valu = d["256"]["kernel_ms_per_launch"], d["256"]["per_realization"]["SQ_INSTS_VALU"]
        v = d["384"]
    dms = ref_ms - v["kernel_ms_per_launch"]
    dvalu = ref_valu - v["per_realization"]["SQ_INSTS_VALU"]
    cyc = dms * CYC_PER_MS                                  # cycles per realization on one CU (each of its 4 SIMDs)
    per_simd = dvalu / 4.0
    rows.append(dict(section=name, ms=dms, valu=dvalu, cycles=cyc, cycles_per_inst=cyc / per_simd, issue_cost=cost,
                     predicted_cycles=per_simd * cost, min_valu=vmin, inst_over_min=dvalu / vmin, time_over_issue=cyc / (per_simd * cost), how=how))
    tot_ms += dms
    tot_valu += dvalu
    tot_min += vmin
rest = d["2016"]
r_ms, r_valu = rest["kernel_ms_per_launch"], rest["per_realization"]["SQ_INSTS_VALU"]
lines = ["# The complex128 headline kernel by section (round 5)", "",
         "`k_run_mimo_ofdm_planar<double, 1024, 4, 4, 4, 2, 12>`, 262 144 realizations per launch, one MI355X; measured by compiling ONE",
         "section out at a time (`-DMCLE_EXPERIMENTS`, option `f64_variant`; `scripts/experiments/r05_c4_sections.sh`, data in",
         "`c4_f64_sections.json`).  Whole kernel: **%.3f ms**, %.0f VALU wave-instructions per realization = %.0f cycles per realization" % (
             b_ms, b_valu, b_ms * CYC_PER_MS),
         "and CU (every SIMD of the CU runs one of the workgroup's four wavefronts, two workgroups resident).", "",
         "| section | ms | VALU instr. | cycles | cycles / instr. (per SIMD) | issue cost of its mix | time / issue | minimum instr. | instr. / minimum |",
         "|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    lines.append("| %s | %.3f | %.0f | %.0f | %.2f | %.1f | **%.2f** | %.0f | **%.2f** |" % (
        r["section"], r["ms"], r["valu"], r["cycles"], r["cycles_per_inst"], r["issue_cost"], r["time_over_issue"], r["min_valu"], r["inst_over_min"]))
lines.append("| skeleton (all six compiled out): scatter stores, the span-1 butterflies of passes C / C', record hand-over, 5 barriers per symbol, loop | %.3f | %.0f | %.0f | %.2f | ~5 | %.2f | — | — |" % (
    r_ms, r_valu, r_ms * CYC_PER_MS, r_ms * CYC_PER_MS / (r_valu / 4.0), r_ms * CYC_PER_MS / (r_valu / 4.0 * 5.0)))
lines.append("| sum of the rows | %.3f | %.0f | | | | | | |" % (tot_ms + r_ms, tot_valu + r_valu))
pred = sum(r["predicted_cycles"] for r in rows) + r_valu / 4.0 * 5.0
lines += ["", "How to read it.",
          "* **instructions x issue cost = %.0f cycles** against %.0f measured: the kernel runs at **%.2f** of what its own instruction" % (pred, b_ms * CYC_PER_MS, pred / (b_ms * CYC_PER_MS)),
          "  stream costs to issue at two wavefronts per SIMD (the counter's `valu_busy` 0.73 books 4 cycles per instruction; an f64",
          "  operation takes 5.1).  The remaining %.0f %% are the skeleton's waits (five workgroup barriers and the first LDS round trip of" % (100 * (1 - pred / (b_ms * CYC_PER_MS))),
          "  every stage per symbol, with one other workgroup per CU to cover them) and the transforms' LDS round trips (time / issue 1.25-1.33).",
          "* **instructions / minimum**: the draw ledger (noise + scatter: %.0f %% of the instructions) is AT its minimum -- it is the" % (100 * (rows[0]["valu"] + rows[2]["valu"]) / b_valu),
          "  mcle-philox-v1 contract (Philox4x32-10 + an f64 Box-Muller within 3e-16 of NumPy's).  The first edition of this table (same",
          "  script, the build of commit c07f5ff: 10.854 ms, 17 838 instructions) read 1.38 for the H x products and 1.52 for the decode, and",
          "  THAT is what the table was for: the ISA of those two sections showed every complex128 multiply-add as v_mul + v_fma + v_add per",
          "  component (the generic `cfma` keeps the product-then-add association) where two chained FMAs do -- `cfma4`, common.hpp:",
          "  10.85 -> %.2f ms, %.0f instructions (profiles/r05/f64_chained_fma_ab.log).  What is left: the H x products (%.2f) carry" % (b_ms, b_valu, rows[3]["inst_over_min"]),
          "  the lane swap of the paired draws (its ablation leaves the LDS reads and a move per product in, hence below 1); the decode at %.2f the LDS address arithmetic of four planes" % rows[5]["inst_over_min"],
          "  and the byte packing; the transforms at %.2f the radix-16 root products (14 %% of a pass: they buy two LDS round trips per" % rows[1]["inst_over_min"],
          "  transform, measured +9 % in round 4) and the swizzled addresses.",
          "* What would still move it: a third wavefront per SIMD (the 64 KiB of complex128 planes per realization allow two workgroups",
          "  per CU: a LDS-capacity bound, not a tuning choice) would cover most of the waits, i.e. <= %.0f %%; nothing in the table is a" % (100 * (1 - pred / (b_ms * CYC_PER_MS))),
          "  factor any more."]
open(os.path.join(REPO, "profiles", "r05", "c4_f64_section_table.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
