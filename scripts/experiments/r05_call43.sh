#!/bin/bash
# round 5, call 43: the new (2048, 2, 4) shape of the complex64 family tests
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_planar_f32.py -q --timeout=600 -k "2048x2x4" 2>&1 | tail -3
