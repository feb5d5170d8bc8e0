#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 data of the normcounts sweep (tools/bench_normcounts.py).
# usage: bash profiles/collect_normcounts.sh <tag>
set -e
tag=${1:-r02}
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_norm_stats -- python3 tools/bench_normcounts.py --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_norm_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_norm_fetch -- python3 tools/bench_normcounts.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_norm_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_norm_write -- python3 tools/bench_normcounts.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_norm_write.log 2>&1
timeout -k 10 300 python3 tools/bench_normcounts.py --steps 5 --warmup 1 > $out/${tag}_normcounts_bench.json 2> $out/${tag}_normcounts_bench.log
python3 profiles/summarize.py stats $out/${tag}_norm_stats $out/${tag}_normcounts_kernel_stats.csv
python3 - <<PY
import csv, glob, json, collections
def avg(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "himut::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("::")[-1].split("<")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
f, w = avg("$out/${tag}_norm_fetch", "FETCH_SIZE"), avg("$out/${tag}_norm_write", "WRITE_SIZE")
doc = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/bench_normcounts.py, chr20-sized 30x contig; KiB per launch as counted. "
               "FETCH_SIZE reports half of the bytes of wide coalesced streaming reads on gfx950 (MI355X_MICROARCH.md): doubled for k_parse_cs and k_callable "
               "(16-byte streaming loads), not for k_norm_col (1- and 2-byte loads per lane: uncalibrated width, counted as is) or k_norm_tile (4-byte unaligned loads)",
       "collected": "profiles/collect_normcounts.sh $tag", "raw_kib_per_launch": {k: {"FETCH_SIZE": f.get(k, 0.0), "WRITE_SIZE": w.get(k, 0.0)} for k in sorted(set(f) | set(w))}}
for k in sorted(set(f) | set(w)):
    doc[k] = int(f.get(k, 0.0) * 1024 * (2.0 if k in ("k_parse_cs", "k_callable") else 1.0) + w.get(k, 0.0) * 1024)
json.dump(doc, open("$out/${tag}_pmc_traffic_normcounts.json", "w"), indent=1, sort_keys=True)
print({k: round(v / 1e9, 3) for k, v in doc.items() if isinstance(v, int)})
PY
tail -c 700 $out/${tag}_normcounts_bench.json
