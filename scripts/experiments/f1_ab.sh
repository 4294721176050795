# needs scripts/experiments/f1_mfma_kernel.patch (git apply, make -C pyphysim_amd/csrc).  Measured on one MI355X (98 304 realizations per
# launch): matrix-core kernel at 2 workgroups per CU 9.40 ms, at 3 per CU (105 spilled registers) 12.3 ms, the VALU kernel 8.18 ms;
# SER identical to 8 digits, the patch's 8 parity tests pass.  VALU instructions per wave and realization (f1_pmc.sh): 9 946 vs 9 808
# -- in this kernel the transforms are a small share next to the tap polynomials, the per-bin frequency response and the
# per-bin 4x4 solves, the MFMA glue costs what the radix-4 butterflies did, and the third workgroup per CU is lost.  Not adopted.
# frequency-selective MIMO-OFDM, one box: matrix-core kernel (2 workgroups per CU, default) / 3 per CU / the VALU kernel
for rep in 1 2; do
for v in "grid_oversub=0" "no_mfma=1"; do      # (the patch's own 3-waves knob, MCLE_F1_MFMA_WAVES, is an environment variable of that patch)
python bench.py --opt $v --config f1 --steps 10 --warmup 2 --no-cpu --pmc off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.4g' % d['value'], '%.3f' % d['roofline']['kernel_ms_per_launch'], d['ser'])"
done; done
