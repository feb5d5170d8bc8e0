"""Turn rocprofv3 CSV output into the summaries kept in this directory.

usage: python profiles/summarize.py stats <dir> <out.csv>       (from --kernel-trace --stats --output-format csv)
       python profiles/summarize.py pmc <fetch_dir> <write_dir> <out.json>   (from the two --pmc passes)
"""
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *{} under {}".format(suffix, d))
    return hits[0]


def stats(d, out):
    rows = list(csv.DictReader(open(find(d, "kernel_stats.csv"))))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc_avg(d, counter):
    acc = {}
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        if not name.startswith("himut::") and "himut::" not in name:
            continue
        name = name.replace("void ", "")
        a = acc.setdefault(name, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


# kernels whose reads are wide coalesced streams (16 bytes per lane): FETCH_SIZE on gfx950 reports half of their bytes.
# (k_parse_cs streams the qualities only in its <true> form, which the call path does not use; there its reads are the cs
# text, 16 bytes per lane as well.)
STREAMING = ("k_parse_cs", "k_stream_capture")
# the kernels of one himut_run (bench.py's step): their sum is roofline_step's counter figure
STEP_KERNELS = ("k_parse_cs", "k_stream_capture", "k_mask_emit", "k_eval_columns", "k_block_sums", "k_block_table3", "k_scan_small",
                "k_finalize_flags", "k_compact", "k_run_totals", "k_window_index", "k_read_hap")


def pmc(fetch_dir, write_dir, out):
    f = pmc_avg(fetch_dir, "FETCH_SIZE")
    w = pmc_avg(write_dir, "WRITE_SIZE")
    doc = {"contig_len": 64444167, "depth": 30.0,      # bench.py's default workload (what collect.sh runs)
           "collected": "profiles/collect.sh " + (os.environ.get("HIMUT_PROFILE_TAG") or "r02") + ", rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes",
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 2 --warmup 1). "
                   "Counter unit = KiB. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports 1/2 of the bytes "
                   "of a wide coalesced streaming read, so fetch bytes are doubled for the streaming kernels ("
                   + ", ".join(STREAMING) + "); WRITE_SIZE is taken as is.",
           "raw_kib_per_launch": {}}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        doc["raw_kib_per_launch"][k] = {"FETCH_SIZE_kb_avg": fk, "WRITE_SIZE_kb_avg": wk}
        short = k.split("::")[-1].split("<")[0]
        mult = 2.0 if short in STREAMING else 1.0
        doc[short] = int(fk * 1024 * mult + wk * 1024)
    # the library's fills and copies (rocclr kernels) and the rocPRIM scans are not in the himut:: namespace: they move
    # tens of megabytes per step (the empty column store, the position bitmap) against the gigabytes above
    doc["step_total"] = int(sum(v for k, v in doc.items() if k in STEP_KERNELS))
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)
