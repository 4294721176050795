#!/bin/bash
# round 4, call 40: seed hunt over the fuzz file on the final build (wavefront config-3 kernel and planar families by default)
export TMPDIR=/tmp
for off in 511 622 733 844 955; do
  MCLE_FUZZ_OFFSET=$off MCLE_FUZZ_TRIALS=50 MCLE_FUZZ_TRIALS_BASE=40 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -6
done
