#!/bin/bash
# A/B of two settings of one environment variable on one box: alternates them, prints ms per step and the stage times.
# usage: bash tools/ab_env.sh <rounds> VAR value_a value_b ...      ("-" = unset)
n=$1; var=$2; shift; shift
for i in $(seq $n); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 > /tmp/ab.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); s=d['stage_ms']; print('%-24s' % '$var=$v', round(d['ms_per_step'],4), round(d['device_ms_per_step'],4), ' '.join('%s %.4f' % (k[3:], s[k]) for k in ('ms_parse','ms_index','ms_capture','ms_emit','ms_eval','ms_finalize')), int(d['candidate_sites_per_step']), int(d['records_per_step']))"
  done
done
