#!/bin/bash
# Kernel times of the normcounts sweep for several builds of the library on one box.  usage: [NORM_ARGS="--error-rate 5e-4"] bash tools/ab_norm.sh lib_a.so lib_b.so ...
export TMPDIR=/tmp
for v in "$@"; do
  rm -rf gpurun_out/abn_stats
  HIMUT_HIP_LIB_OVERRIDE=$PWD/$v timeout -k 5 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abn_stats -- python3 tools/bench_normcounts.py --steps 3 --warmup 1 --no-cpu-baseline $NORM_ARGS > gpurun_out/abn.log 2>&1
  echo "== $v"; python tools/kstats.py gpurun_out/abn_stats | grep -E "k_norm_|k_callable|k_parse|k_read" | cut -c1-40,100-
done
rm -rf gpurun_out/abn_stats
