#!/bin/bash
# Wait / issue counters of the call path's kernels (bench.py).  usage: bash tools/pmc_call_more.sh
export TMPDIR=/tmp
for c in "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/pmcc_$tag
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs '' > gpurun_out/pmcc.log 2>&1 || echo "failed $c"
done
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "himut::" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if v.get("SQ_WAVE_CYCLES",[0])[0] < 1e7: continue
    print(k+" "+" ".join("%s=%.4g"%(c,sum(x)/len(x)) for c,x in sorted(v.items())))
PY
rm -rf gpurun_out/pmcc_*
