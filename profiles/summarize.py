"""Turn rocprofv3 CSV output into the summaries kept in this directory.

usage: python profiles/summarize.py stats <dir> <out.csv>       (from --kernel-trace --stats --output-format csv)
       python profiles/summarize.py pmc <fetch_dir> <write_dir> <out.json>   (from the two --pmc passes)
       python profiles/summarize.py normpmc <fetch_dir> <write_dir> <rows_fetch_dir> <out.json> <tag>   (the normcounts pass)
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
STREAMING = ("k_parse_cs", "k_stream_capture", "k_flag_bases")
# the kernels of one himut_run (bench.py's step): their sum is roofline_step's counter figure
STEP_KERNELS = ("k_parse_cs", "k_stream_capture", "k_mask_emit", "k_eval_columns", "k_block_sums", "k_block_table3", "k_scan_small",
                "k_finalize_flags", "k_compact", "k_run_totals", "k_window_index", "k_read_hap")


def pmc(fetch_dir, write_dir, out):
    f = pmc_avg(fetch_dir, "FETCH_SIZE")
    w = pmc_avg(write_dir, "WRITE_SIZE")
    doc = {"contig_len": 64444167, "depth": 30.0,      # bench.py's default workload (what collect.sh runs)
           "collected": "profiles/collect.sh " + (os.environ.get("HIMUT_PROFILE_TAG") or "r03") + ", rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes",
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


def rows_calibration(d):
    """tools/ubench_rows.hip under --pmc FETCH_SIZE: bytes the kernels are known to read (every row once) over the bytes
    counted, per (rows in flight, arrays read, aligned) variant."""
    L, STEP, npos = 15008, 500, 64444167
    nreads, ntiles = npos // STEP, npos // 256
    rows = 0
    for t in range(ntiles):
        base = t * 256
        lo = max(0, -((-(base + 256 - L)) // STEP))
        hi = min(nreads, base // STEP + 1)
        rows += max(0, hi - lo)
    acc = {}
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        if r["Counter_Name"] != "FETCH_SIZE" or "k_rows" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc.setdefault(name, [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
    out = {}
    for name, (tot, n) in sorted(acc.items()):
        args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")
        mode = int(args[1])
        known = rows * ((256 if mode & 1 else 0) + (128 if mode & 2 else 0) + (32 if mode & 4 else 0))
        out[name] = {"known_bytes": known, "FETCH_SIZE_bytes": tot / n * 1024, "known_over_counted": known / (tot / n * 1024)}
    return out


def normpmc(fetch_dir, write_dir, rows_dir, out, tag):
    f = pmc_avg(fetch_dir, "FETCH_SIZE")
    w = pmc_avg(write_dir, "WRITE_SIZE")
    cal = rows_calibration(rows_dir)
    # the sweep's three arrays read with rows of four in flight at any alignment: k_norm_quad's own pattern
    key = [k for k in cal if k.replace(" ", "").endswith("<4,7,false>")]
    quad_mult = cal[key[0]]["known_over_counted"] if key else 1.0
    mult = {"k_parse_cs": 2.0, "k_callable": 2.0, "k_flag_bases": 2.0, "k_norm_quad": quad_mult}
    doc = {"collected": "profiles/collect_normcounts.sh " + tag,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/bench_normcounts.py, chr20-sized 30x contig; KiB per "
                   "launch as counted in raw_kib_per_launch.  FETCH_SIZE reports half of the bytes of a wide coalesced streaming read on "
                   "gfx950 (MI355X_MICROARCH.md): doubled for k_parse_cs and k_callable (16 bytes per lane).  k_norm_quad reads rows of "
                   "256 bases with a dword per lane at any alignment, a width the guide calls uncalibrated: its factor is measured on "
                   "tools/ubench_rows.hip, which reads the same three arrays in the same pattern and every byte once "
                   "(rows_calibration; the <4, 7, false> variant).  The other kernels are taken as counted.  WRITE_SIZE as is.",
           "rows_calibration": cal, "k_norm_quad_fetch_factor": quad_mult, "raw_kib_per_launch": {}}
    for k in sorted(set(f) | set(w)):
        short = k.split("::")[-1].split("<")[0]
        doc["raw_kib_per_launch"][k] = {"FETCH_SIZE": f.get(k, 0.0), "WRITE_SIZE": w.get(k, 0.0)}
        doc[short] = int(f.get(k, 0.0) * 1024 * mult.get(short, 1.0) + w.get(k, 0.0) * 1024)
    doc["sweep_total"] = int(sum(doc.get(k, 0) for k in ("k_norm_plan", "k_norm_quad", "k_norm_dirty", "k_norm_tile")))
    doc["pass_total"] = int(sum(v for k, v in doc.items() if k.startswith("k_") and isinstance(v, int)))
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    print({k: round(v / 1e9, 3) for k, v in doc.items() if isinstance(v, int)}, "quad factor", round(quad_mult, 3))


def edgespmc(fetch_dir, write_dir, out, tag):
    """The edge-count leg (bench.py --legs edges): k_parse_cs<false> + k_edges per pass."""
    f = pmc_avg(fetch_dir, "FETCH_SIZE")
    w = pmc_avg(write_dir, "WRITE_SIZE")
    doc = {"collected": "profiles/collect.sh " + tag + " (bench.py --legs edges under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
           "note": "KiB per launch as counted in raw_kib_per_launch.  k_parse_cs reads the cs text 16 bytes per lane: FETCH_SIZE doubled "
                   "(MI355X_MICROARCH.md).  k_edges reads scattered 64-byte sectors (a byte or a nibble per usable (read, hetSNP)): an "
                   "uncalibrated width, taken as counted.  The averages are over every launch of the process: k_parse_cs also runs in "
                   "the headline's steps, with the same input.",
           "raw_kib_per_launch": {}}
    for k in sorted(set(f) | set(w)):
        short = k.split("::")[-1].split("<")[0]
        if short not in ("k_parse_cs", "k_edges"):
            continue
        doc["raw_kib_per_launch"][k] = {"FETCH_SIZE": f.get(k, 0.0), "WRITE_SIZE": w.get(k, 0.0)}
        doc[short] = int(f.get(k, 0.0) * 1024 * (2.0 if short == "k_parse_cs" else 1.0) + w.get(k, 0.0) * 1024)
    doc["pass_total"] = int(doc.get("k_parse_cs", 0) + doc.get("k_edges", 0))
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    print({k: round(v / 1e9, 3) for k, v in doc.items() if isinstance(v, int)})


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "edgespmc":
        edgespmc(*sys.argv[2:6])
    elif sys.argv[1] == "normpmc":
        normpmc(*sys.argv[2:7])
    else:
        raise SystemExit(__doc__)
