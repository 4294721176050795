# needs the stage-ablation knob: git apply scripts/experiments/tdl_ablate.patch && make -C pyphysim_amd/csrc (results are
# meaningless with stages off; the knob is not in the product kernel).  Bits: 1 rays, 2 tap polynomials (both only after the
# first pass), 4 noise, 8 taps 1.., 16 equaliser taps 1.., 32 equalise + demodulate.
for ab in 0 1 2 3 4 8 16 32 63; do MCLE_TDL_ABLATE=$ab python bench.py --config c3 --steps 10 --warmup 2 --no-cpu --pmc off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate $ab', '%.3g' % d['value'], '%.3f' % d['roofline']['kernel_ms_per_launch'])"; done
