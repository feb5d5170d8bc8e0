#!/bin/bash
# A/B of two builds of the library on one box: alternates them, prints device ms per step and two stage times.
# usage: bash tools/ab.sh himut_amd/libhimut_hip_a.so himut_amd/libhimut_hip_b.so [rounds]
a=$1; b=$2; n=${3:-3}
for i in $(seq $n); do
  for v in $a $b; do
    HIMUT_HIP_LIB_OVERRIDE=$PWD/$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 > /tmp/ab.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); s=d['stage_ms']; print('$v', round(d['device_ms_per_step'],4), 'parse', round(s['ms_parse'],4), 'emit', round(s['ms_emit'],4), 'capture', round(s['ms_capture'],4), 'eval', round(s['ms_eval'],4))"
  done
done
