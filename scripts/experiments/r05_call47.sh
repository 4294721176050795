#!/bin/bash
# round 5, call 47: f1 per size and geometry on the final build (scripts/experiments/r05_f1_shapes.py)
export TMPDIR=/tmp
python scripts/experiments/r05_f1_shapes.py final
