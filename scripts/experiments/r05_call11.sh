#!/bin/bash
# round 5, call 11: oracle-depth tests of every default kernel
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_oracle_depth.py tests/test_gpu_mimo_tdl_wave.py -m gpu -q --timeout=900 --durations=8 2>&1 | tail -22
