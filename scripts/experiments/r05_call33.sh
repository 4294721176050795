#!/bin/bash
# round 5, call 33: smoke() and the f1 suite with the delay-class edge cases on the final build
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_mimo_tdl_wave.py -q --timeout=900 2>&1 | tail -3
