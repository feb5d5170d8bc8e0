#!/bin/bash
# Wait / issue counters of the normcounts kernels for one build of the library.  usage: bash tools/pmc_norm_more.sh lib.so
export TMPDIR=/tmp
v=$1
for c in "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/pmcm_$tag
  HIMUT_HIP_LIB_OVERRIDE=$PWD/$v timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcm_$tag -- python3 tools/bench_normcounts.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcm.log 2>&1 || echo "failed $c"
done
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcm_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "k_norm_" in k or "k_callable" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k+" "+" ".join("%s=%.4g"%(c,sum(x)/len(x)) for c,x in sorted(v.items())))
PY
rm -rf gpurun_out/pmcm_*
