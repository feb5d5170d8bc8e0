#!/bin/bash
# On the GPU box: normcounts tests, kernel times and FETCH_SIZE of the normcounts bench.  usage: bash tools/norm_check.sh <tag>
tag=${1:-norm}
export TMPDIR=/tmp
python -m pytest tests/test_gpu_normcounts.py -x -q > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${tag}_pytest.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 tools/bench_normcounts.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1
python tools/kstats.py gpurun_out/${tag}_stats | grep -E "k_norm_|k_callable|k_parse_cs|k_read_live"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -- python3 tools/bench_normcounts.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_fetch.log 2>&1
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/${tag}_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if r["Counter_Name"]=="FETCH_SIZE" and ("k_norm_" in k or "k_callable" in k): acc[k].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, "FETCH_SIZE KiB avg %.0f -> %.2f GB raw, x2 = %.2f GB" % (sum(v)/len(v), sum(v)/len(v)*1024/1e9, 2*sum(v)/len(v)*1024/1e9))
PY
python tools/bench_normcounts.py --steps 5 --warmup 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('device_ms', d['device_ms'], 'ms_per_step', d['ms_per_step'])"
