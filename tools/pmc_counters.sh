set -e
export TMPDIR=/tmp
out=gpurun_out
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmcP_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmcP_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcP_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "himut::" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmcP_summary.txt","w") as o:
    for k,v in acc.items():
        o.write(k+" "+" ".join("%s=%.3g"%(c,sum(x)/len(x)) for c,x in sorted(v.items()))+"\n")
print(open("gpurun_out/pmcP_summary.txt").read())
PY
