#!/bin/bash
# round 4, call 45: config-3 profiles under the final kernel names (the wavefront kernel is a template over fft_size now)
export TMPDIR=/tmp
bash scripts/prof_r04.sh c3 c3_f64 > gpurun_out/prof_r04.log 2>&1; tail -2 gpurun_out/prof_r04.log
python scripts/collect_profiles.py r04 2>&1 | grep -E "^c3 |^c3_f64 " | cut -c1-300
