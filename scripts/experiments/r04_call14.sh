#!/bin/bash
# round 4, call 14: general IA on the 6x6 capacity (Nr = 5, Nt = 3, Ns = 2 ...), then the whole suite
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ia_general.py tests/test_gpu_ia_base.py tests/test_gpu_bd.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|Error|error" | tail -12
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -12
