#!/bin/bash
# A/B/... of several builds of the library on one box: alternates them, prints device ms per step and the stage times.
# usage: bash tools/abn.sh <rounds> lib_a.so lib_b.so ...
n=$1; shift
for i in $(seq $n); do
  for v in "$@"; do
    HIMUT_HIP_LIB_OVERRIDE=$PWD/$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 > /tmp/ab.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); s=d['stage_ms']; print('%-40s' % '$v', round(d['device_ms_per_step'],4), ' '.join('%s %.4f' % (k[3:], s[k]) for k in ('ms_parse','ms_index','ms_capture','ms_emit','ms_eval','ms_finalize')))"
  done
done
