#!/usr/bin/env python3
"""Register / LDS / scratch usage of the fused kernels as the COMPILER recorded it (code-object metadata), for
profiles/<round>/kernel_resources.json.  Runs on the build container (no GPU needed).

rocprofv3's per-dispatch VGPR_Count on gfx950 reports HALF the allocation (granule of 4 assumed, the hardware's is
8: 84 for a kernel whose .vgpr_count is 168) and its LDS_Block_Size omits dynamic LDS, so the trace's `_dispatch`
record cannot corroborate occupancy claims; this file can.  Dynamic LDS is added by the host code at launch
(run_*_impl in csrc/*.hip) and is listed in DESIGN.md.

usage: python scripts/kernel_resources.py r05      (after `make -C pyphysim_amd/csrc`: reads the built *.o files)
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
# kernels listed per translation unit (every *.o of the in-tree build is read: `make -C pyphysim_amd/csrc` first)
PREFIXES = ("k_run_", "k_link_walk", "k_mimo_filters", "k_tdl_symbol_polys", "k_mimo_tdl_symbol_polys", "k_mimo_flat_", "k_ia_solve_links",
            "k_ia_link", "k_bd_solve_links", "k_bd_link", "k_ofdm_mod_1024_mfma", "k_ofdm_demod_1024_mfma", "k_jakes_blocks",
            "k_jakes_mfma")
KEYS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
        ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size")

# What the dispatchers pick with every option at 0 ("default": true): regular expressions over the demangled kernel names.
# Everything else in the library is reachable through an option (A/B runs, kernel-vs-kernel tests) or outside a default's envelope.
DEFAULT = [
    # config 4 family (pipeline_mimo_planar.hip: run_mimo_ofdm_planar_t): 1024 with four receive antennas = radix-16 passes
    # (complex128 fused, two per SIMD; complex64 unfused, four), every other shape the radix-4 table
    # (round 6: complex128 4 x 4 inside the full-band / even-prefix / certificate envelope = the part-wave kernels -- pw<NW, decision form,
    #  wavefronts per SIMD, 0>: 512 / 1024 at three, 2048 at two -- and at 256 the full-wave kernel fw<antennas, decision form, 3, 0>,
    #  also for 2 x 2; the planar kernels below serve everything outside that envelope)
    r"k_run_mimo_ofdm_pw<[24], [1-4], 3, 0>", r"k_run_mimo_ofdm_pw<8, [1-4], 2, 0>", r"k_run_mimo_ofdm_fw<[24], [1-4], 3, 0>",
    r"k_run_mimo_ofdm_planar<double, 1024, [1-4], 4, 4, 2, 12>", r"k_run_mimo_ofdm_planar<float, 1024, [1-4], 4, 4, 4, 4>",
    # (last day of round 6: complex64 at 256 points, and at 512 with two receive antennas, two more wavefronts per SIMD than the table)
    r"k_run_mimo_ofdm_planar<double, 256, [12], 2, 2, 2, 0>", r"k_run_mimo_ofdm_planar<double, 256, [1-3], 3, 3, 2, 0>",
    r"k_run_mimo_ofdm_planar<double, 256, [1-4], 4, 2, 3, 0>",
    r"k_run_mimo_ofdm_planar<float, 256, [12], 2, 2, 4, 0>", r"k_run_mimo_ofdm_planar<float, 256, [1-3], 3, 3, 4, 0>",
    r"k_run_mimo_ofdm_planar<float, 256, [1-3], 4, 2, 5, 0>", r"k_run_mimo_ofdm_planar<float, 256, 4, 4, 2, 3, 0>",
    r"k_run_mimo_ofdm_planar<double, 512, [12], 2, 2, 3, 0>", r"k_run_mimo_ofdm_planar<float, 512, [12], 2, 2, 5, 0>",
    r"k_run_mimo_ofdm_planar<(double|float), 512, [1-3], 3, 3, 2, 0>",
    r"k_run_mimo_ofdm_planar<(double|float), 512, [1-4], 4, 2, 3, 0>",
    r"k_run_mimo_ofdm_planar<(double|float), 1024, [12], 2, 2, 3, 0>", r"k_run_mimo_ofdm_planar<(double|float), 1024, [1-3], 3, 3, 2, 0>",
    r"k_run_mimo_ofdm_planar<(double|float), 2048, [12], 2, 2, 4, 0>", r"k_run_mimo_ofdm_planar<(double|float), 2048, [1-3], 3, 3, 2, 0>",
    r"k_run_mimo_ofdm_planar<(double|float), 2048, [12], 4, 2, 4, 0>", r"k_run_mimo_ofdm_planar<double, 2048, 3, 4, 2, 4, 0>",
    r"k_run_mimo_ofdm_planar<(double|float), 2048, 4, 4, 4, 2, 0>", r"k_run_mimo_ofdm_planar<float, 2048, 3, 4, 4, 2, 0>", r"k_mimo_filters_planar<",
    # config 3 (pipeline_siso_tdl.hip: the wavefront kernel where it is the faster one, the batched kernels otherwise)
    # (2048 points since the last day of round 6: two wavefronts per realization, k_run_ofdm_tdl_hw<T, order, wavefronts per SIMD,
    #  realizations per workgroup>; the one-wavefront kernels there serve a delay beyond the prefix and large constellation tables)
    r"k_run_ofdm_tdl_hw<float, \d, 3, 1>", r"k_run_ofdm_tdl_hw<double, \d, 2, 2>",
    r"k_run_ofdm_tdl_wave<float, 1024, \d, 3>", r"k_run_ofdm_tdl_wave<float, 2048, \d, 2>", r"k_run_ofdm_tdl_wave<float, 256, \d, 6>",
    r"k_run_ofdm_tdl_wave<float, 512, \d, 4>", r"k_run_ofdm_tdl_wave<double, 256, \d, 5>", r"k_run_ofdm_tdl_wave<double, 512, \d, 3>",
    r"k_run_ofdm_tdl_wave<double, 1024, \d, 2>", r"k_run_ofdm_tdl_wave<double, 2048, \d, 1>",
    r"k_run_ofdm_tdl_batch<(float|double), (64|128), \d>", r"k_tdl_symbol_polys<",
    # f1 (pipeline_mimo_tdl.hip): one receive antenna per wavefront inside its envelope, the cooperative kernel for the rest
    r"k_run_mimo_ofdm_tdl_wave<", r"k_mimo_tdl_symbol_polys<", r"k_run_mimo_ofdm_tdl<(float|double), (64|128), [24]>",
    # configs 1 / 2, config 5, f6, the flat MIMO application
    r"k_run_flat_mfma<", r"k_run_flat<double", r"k_run_flat<float, \d+, 0>", r"k_ia_solve_links<", r"k_ia_link<", r"k_bd_solve_links",
    r"k_bd_link<", r"k_mimo_flat_", r"k_link_walk<",     # (round 6: the packed walk for an even number of columns >= 128; k_ia_link / k_bd_link otherwise)
]


def waves_per_simd(vgpr, agpr):
    alloc = -(-(vgpr + agpr) // 8) * 8          # unified file, granule 8 (MI355X_MICROARCH.md, register files)
    return min(8, 512 // max(alloc, 8))


def main():
    import glob
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(CSRC, "*.o"))):
            fat, elf = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "a.elf")
            for f in (fat, elf):
                if os.path.exists(f):
                    os.remove(f)
            r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True)
            if r.returncode != 0 or not os.path.exists(fat):
                continue                       # a translation unit without device code
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + elf,
                            "--unbundle"], check=True)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], check=True,
                                   capture_output=True, text=True).stdout
            for block in notes.split("  - .agpr_count:")[1:]:
                block = ".agpr_count:" + block
                name = re.search(r"\.name:\s+(\S+)", block).group(1)
                demangled = subprocess.run(["c++filt", name], capture_output=True,
                                           text=True).stdout.strip()
                short = demangled.split("(")[0].replace("void mcle::", "")
                if not short.startswith(PREFIXES):
                    continue
                rec = {}
                for k in KEYS:
                    m = re.search(re.escape(k) + r":\s+(\d+)", block)
                    if m:
                        rec[k.lstrip(".")] = int(m.group(1))
                rec["waves_per_simd_by_registers"] = waves_per_simd(rec.get("vgpr_count", 0), rec.get("agpr_count", 0))
                rec["default"] = any(re.match(rx, short) for rx in DEFAULT)
                rec["object"] = os.path.basename(obj)
                out[short] = rec
    dst = os.path.join(REPO, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    spilled = {k: v["vgpr_spill_count"] for k, v in sorted(out.items()) if v["default"] and v.get("vgpr_spill_count")}
    doc = {"_source": "code-object metadata of the in-tree build (llvm-objcopy .hip_fatbin + clang-offload-bundler + llvm-readelf "
                      "--notes), gfx950, flags of pyphysim_amd/csrc/Makefile",
           "_default": "\"default\": true = what the dispatchers pick with every option at 0 (scripts/kernel_resources.py DEFAULT)",
           "_default_kernels_with_spilled_vgprs": spilled,
           "kernels": dict(sorted(out.items()))}
    json.dump(doc, open(os.path.join(dst, "kernel_resources.json"), "w"), indent=1)
    print(len(out), "kernels;", sum(1 for v in out.values() if v["default"]), "default;", len(spilled), "default kernels spill:")
    for k, n in spilled.items():
        v = out[k]
        print("  %-70s vgpr %3d spilled %3d scratch %4d B" % (k, v["vgpr_count"], n, v.get("private_segment_fixed_size", 0)))


if __name__ == "__main__":
    main()
