#!/bin/bash
# Run on the GPU box (through gpurun): collects the rocprofv3 data the summaries in profiles/ are made from.
# usage: bash profiles/collect.sh <tag>      -> gpurun_out/<tag>_{stats,fetch,write}/, gpurun_out/<tag>_bench.json
set -e
tag=${1:-r03}
export HIMUT_PROFILE_TAG=$tag
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs '' > $out/${tag}_stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs '' > $out/${tag}_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs '' > $out/${tag}_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_edges_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs edges > $out/${tag}_edges_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_edges_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs edges > $out/${tag}_edges_write.log 2>&1
python3 profiles/summarize.py edgespmc $out/${tag}_edges_fetch $out/${tag}_edges_write $out/${tag}_pmc_traffic_edges.json $tag
timeout -k 10 600 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.log
tail -c 600 $out/${tag}_bench.json
python3 profiles/summarize.py stats $out/${tag}_stats $out/${tag}_kernel_stats.csv
python3 profiles/summarize.py pmc $out/${tag}_fetch $out/${tag}_write $out/${tag}_pmc_traffic.json
