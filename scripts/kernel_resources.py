#!/usr/bin/env python3
"""Register / LDS / scratch usage of the fused kernels as the COMPILER recorded it (code-object metadata), for
profiles/<round>/kernel_resources.json.  Runs on the build container (no GPU needed).

rocprofv3's per-dispatch VGPR_Count on gfx950 reports HALF the allocation (granule of 4 assumed, the hardware's is
8: 84 for a kernel whose .vgpr_count is 168) and its LDS_Block_Size omits dynamic LDS, so the trace's `_dispatch`
record cannot corroborate occupancy claims; this file can.  Dynamic LDS is added by the host code at launch
(run_*_impl in csrc/*.hip) and is listed in DESIGN.md.

usage: python scripts/kernel_resources.py r03
"""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "pyphysim_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
         "-ffp-contract=fast", "--cuda-device-only", "-c"]
SOURCES = {"pipelines.hip": ("k_run_mimo_ofdm", "k_run_flat", "k_run_flat_mfma", "k_run_ofdm_tdl"),
           "pipeline_mimo_mfma.hip": ("k_run_mimo_ofdm_mfma", "k_mimo_filters"),
           "pipeline_mimo_planar.hip": ("k_run_mimo_ofdm_planar", "k_mimo_filters_planar"),
           "pipeline_siso_tdl.hip": ("k_run_ofdm_tdl_batch", "k_run_ofdm_tdl_mfma", "k_tdl_symbol_polys"),
           "pipeline_siso_tdl_wave_f32.hip": ("k_run_ofdm_tdl_wave",),
           "pipeline_siso_tdl_wave_f64.hip": ("k_run_ofdm_tdl_wave",),
           "pipeline_mimo_tdl.hip": ("k_run_mimo_ofdm_tdl", "k_mimo_tdl_symbol_polys"),
           "pipeline_mimo_flat.hip": ("k_mimo_flat_setup", "k_mimo_flat_link"),
           "kernels_ia.hip": ("k_ia_solve_links", "k_ia_link"),
           "kernels_bd.hip": ("k_bd_solve_links", "k_bd_link"),
           "kernels_ofdm_mfma.hip": ("k_ofdm_mod_1024_mfma", "k_ofdm_demod_1024_mfma"),
           "kernels_channel.hip": ("k_jakes_blocks", "k_jakes_mfma")}
KEYS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
        ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size")


def waves_per_simd(vgpr, agpr):
    alloc = -(-(vgpr + agpr) // 8) * 8          # unified file, granule 8 (MI355X_MICROARCH.md, register files)
    return min(8, 512 // max(alloc, 8))


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for src, kernels in SOURCES.items():
            bundle, elf = os.path.join(tmp, "a.bundle"), os.path.join(tmp, "a.elf")
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, src), "-o", bundle], check=True)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + bundle, "--output=" + elf,
                            "--unbundle"], check=True)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], check=True,
                                   capture_output=True, text=True).stdout
            for block in notes.split("  - .agpr_count:")[1:]:
                block = ".agpr_count:" + block
                name = re.search(r"\.name:\s+(\S+)", block).group(1)
                demangled = subprocess.run(["c++filt", name], capture_output=True,
                                           text=True).stdout.strip()
                short = demangled.split("(")[0].replace("void mcle::", "")
                if not any(short.startswith(k + "<") or short == k for k in kernels):
                    continue
                rec = {}
                for k in KEYS:
                    m = re.search(re.escape(k) + r":\s+(\d+)", block)
                    if m:
                        rec[k.lstrip(".")] = int(m.group(1))
                rec["waves_per_simd_by_registers"] = waves_per_simd(rec.get("vgpr_count", 0), rec.get("agpr_count", 0))
                out[short] = rec
    dst = os.path.join(REPO, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    doc = {"_source": "code-object metadata (hipcc --cuda-device-only + llvm-readelf --notes), gfx950, flags of "
                      "pyphysim_amd/csrc/Makefile", "kernels": dict(sorted(out.items()))}
    json.dump(doc, open(os.path.join(dst, "kernel_resources.json"), "w"), indent=1)
    for k, v in sorted(out.items()):
        if "mfma" in k or "<float>" in k or "float, 2>" in k:
            print(k, v["vgpr_count"], v.get("agpr_count"), "spilled", v.get("vgpr_spill_count"), "scratch", v.get("private_segment_fixed_size"),
                  "waves/SIMD", v["waves_per_simd_by_registers"])


if __name__ == "__main__":
    main()
