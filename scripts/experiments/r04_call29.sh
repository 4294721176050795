#!/bin/bash
# round 4, call 29: seed hunt over the whole fuzz file (four offsets x 40 trials) on the final build, then the whole suite
export TMPDIR=/tmp
for off in 101 202 303 404; do
  MCLE_FUZZ_OFFSET=$off MCLE_FUZZ_TRIALS=40 MCLE_FUZZ_TRIALS_BASE=20 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -6
done
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -8
