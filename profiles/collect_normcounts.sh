#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 data of the normcounts sweep (tools/bench_normcounts.py = bench.py's
# normcounts leg by itself), and the calibration of FETCH_SIZE for the sweep's access pattern (tools/ubench_rows.hip: rows of
# 256 qualities, a dword per lane at any alignment, every byte read once -> a known byte count).
# usage: bash profiles/collect_normcounts.sh <tag>
set -e
tag=${1:-r03}
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_norm_stats -- python3 tools/bench_normcounts.py --steps 5 --warmup 1 --no-cpu-baseline > $out/${tag}_norm_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_norm_fetch -- python3 tools/bench_normcounts.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_norm_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_norm_write -- python3 tools/bench_normcounts.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_norm_write.log 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_rows tools/ubench_rows.hip
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_rows_fetch -- /tmp/ubench_rows > $out/${tag}_rows_fetch.log 2>&1
timeout -k 10 300 python3 tools/bench_normcounts.py --steps 5 --warmup 1 > $out/${tag}_normcounts_bench.json 2> $out/${tag}_normcounts_bench.log
python3 profiles/summarize.py stats $out/${tag}_norm_stats $out/${tag}_normcounts_kernel_stats.csv
python3 profiles/summarize.py normpmc $out/${tag}_norm_fetch $out/${tag}_norm_write $out/${tag}_rows_fetch $out/${tag}_pmc_traffic_normcounts.json $tag
tail -c 900 $out/${tag}_normcounts_bench.json
