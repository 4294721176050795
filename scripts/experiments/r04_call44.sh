#!/bin/bash
# round 4, call 44: whole GPU suite on the final build
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
