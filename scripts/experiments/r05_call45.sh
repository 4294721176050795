#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_planar_f32.py tests/test_gpu_f64_kernel.py -q --timeout=600 2>&1 | tail -3
