#!/bin/bash
# round 5, call 44: the whole GPU suite and smoke() on the final build
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head
