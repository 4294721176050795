#!/usr/bin/env python3
"""Mean per launch of every rocprofv3 --pmc counter for kernels whose name contains NEEDLE, plus the derived fractions.
usage: python scripts/pmc_summary.py DIR NEEDLE REALIZATIONS_PER_LAUNCH"""
import collections
import csv
import glob
import sys


def main():
    root, needle, per = sys.argv[1], sys.argv[2], float(sys.argv[3])
    agg = collections.defaultdict(list)
    info = None
    for p in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            if needle in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
                info = (row["Kernel_Name"][:90], row.get("VGPR_Count"), row.get("LDS_Block_Size"), row.get("Scratch_Size"),
                        row.get("Grid_Size"))
    print(info)
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    for k, v in sorted(m.items()):
        print("%-28s %16.0f  per realization %12.1f" % (k, v, v / per))
    if "GRBM_GUI_ACTIVE" in m:
        simd = m["GRBM_GUI_ACTIVE"] / 8 * 1024
        if "SQ_ACTIVE_INST_VALU" in m:
            print("valu_busy_chip %.3f" % (4 * m["SQ_ACTIVE_INST_VALU"] / simd))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            print("mfma_busy_chip %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd))
        print("kernel cycles (GRBM/8) %.0f = %.3f ms at 2.4 GHz" % (m["GRBM_GUI_ACTIVE"] / 8, m["GRBM_GUI_ACTIVE"] / 8 / 2.4e6))
    if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
        print("wait_inst_any_frac %.3f" % (m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        print("lds_bank_conflict_frac %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
    if "SQ_INSTS_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
        print("cycles per VALU wave-inst %.2f" % (4 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]))


if __name__ == "__main__":
    main()
