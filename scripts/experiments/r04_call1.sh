#!/bin/bash
# round 4, call 1: the margin certificate through the whole GPU suite, then the complex128 config-4 kernel variants
# (MCLE_OPT_F64_VARIANT) side by side on one box: ms per 262 144 realizations, min-distance and slicer.
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x 2>&1 | tail -4
one() {  # name, extra args
  python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c4 --batch 262144 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
}
for dm in mindist slicer; do
  one "512thr $dm nocert" --demod $dm --opt demod_nocert=1
  one "512thr $dm var0" --demod $dm
  one "512thr $dm var1" --demod $dm --opt f64_variant=1
  one "512thr $dm var9" --demod $dm --opt f64_variant=9
  one "512thr $dm var16" --demod $dm --opt f64_variant=16
  one "256thr $dm var0" --demod $dm --opt f64_threads=256
  one "256thr $dm var4" --demod $dm --opt f64_threads=256 --opt f64_variant=4
  one "256thr $dm var1" --demod $dm --opt f64_threads=256 --opt f64_variant=1
  one "256thr $dm var5" --demod $dm --opt f64_threads=256 --opt f64_variant=5
  one "256thr $dm var9" --demod $dm --opt f64_threads=256 --opt f64_variant=9
  one "256thr $dm var13" --demod $dm --opt f64_threads=256 --opt f64_variant=13
done
# the other min-distance pipelines, certificate on / off (complex128 and complex64)
for dt in f64 f32; do
for cfg in c2 c3 c5 f1 f6; do
  b=131072; [ $cfg = c2 ] && b=16384; [ $cfg = f1 ] && b=98304; [ $cfg = c5 ] && b=262144
  for nc in 0 1; do
  python bench.py --steps 5 --warmup 1 --no-cpu --pmc off --single-demod --demod mindist --dtype $dt --config $cfg --batch $b --opt demod_nocert=$nc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $dt mindist nocert=$nc', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
  done
done; done
python bench.py --steps 5 --warmup 1 --no-cpu --pmc off --single-demod --demod mindist --dtype f32 --config c4 --batch 262144 --opt demod_nocert=1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 f32 mindist nocert=1', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"
python bench.py --steps 5 --warmup 1 --no-cpu --pmc off --single-demod --demod mindist --dtype f32 --config c4 --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 f32 mindist cert', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"
