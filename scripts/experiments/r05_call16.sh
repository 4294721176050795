#!/bin/bash
# round 5, call 16: the whole GPU suite after the sector certificate, the walks, config 2 and the planar defaults
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -x 2>&1 | tail -12
