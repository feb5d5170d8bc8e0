#!/bin/bash
# A few instruction / wait counters of the normcounts kernels for several builds.  usage: bash tools/pmc_norm_few.sh lib_a.so lib_b.so ...
export TMPDIR=/tmp
for v in "$@"; do
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/pmcf_$tag
  HIMUT_HIP_LIB_OVERRIDE=$PWD/$v timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcf_$tag -- python3 tools/bench_normcounts.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcf.log 2>&1 || echo "failed $c"
done
echo "== $v"
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcf_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "k_norm_" in k or "k_callable" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k+" "+" ".join("%s=%.4g"%(c.replace("SQ_",""),sum(x)/len(x)) for c,x in sorted(v.items())))
PY
rm -rf gpurun_out/pmcf_*
done
