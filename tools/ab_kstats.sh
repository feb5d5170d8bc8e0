#!/bin/bash
# rocprofv3 kernel averages of the call step for several builds of the library on one box.  usage: bash tools/ab_kstats.sh <kernel regex> lib_a.so lib_b.so ...
export TMPDIR=/tmp
pat=$1; shift
for r in 1 2; do
for v in "$@"; do
  rm -rf gpurun_out/abk_stats
  HIMUT_HIP_LIB_OVERRIDE=$PWD/$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/abk.log 2>&1
  echo "== $v"; python tools/kstats.py gpurun_out/abk_stats | grep -E "$pat" | cut -c1-44,100-
done
done
rm -rf gpurun_out/abk_stats
