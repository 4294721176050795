#!/bin/bash
# round 5, call 17: the select-free sector certificate: parity on adversarial points, every PSK pipeline test, then the rates
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_demod_cert.py tests/test_gpu_pipelines.py tests/test_gpu_tdl_wave.py tests/test_gpu_fuzz.py tests/test_gpu_mimo_tdl_wave.py tests/test_gpu_bd.py -m gpu -q --timeout=900 2>&1 | tail -4
python scripts/experiments/r05_psk_rates.py
