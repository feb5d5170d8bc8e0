#!/usr/bin/env python3
"""Benchmark of the hot path: `himut call`'s per-chromosome CCS pileup scan on
MI355X.  One "step" = one full pass of the scan (all kernels of himut_run) over
one synthetic 30x contig that is already resident in HBM.  With N ranks each
rank owns its own contig (the reference's starmap axis, caller.py:766-810) and
every step ends with the RCCL gather of the record buffers to rank 0.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHR20_LEN = 64_444_167
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# algorithmic HBM bytes per unit of work, per kernel (DESIGN.md "Kernels and their rooflines")
STAGE_KERNEL = {"ms_parse": "k_parse_cs", "ms_emit": "k_mask_emit",
                "ms_capture": "k_stream_capture", "ms_eval": "k_eval_columns"}   # a stage's dominant kernel


def algorithmic_bytes(stage, st, cs_bytes):
    """Bytes the design has to move per launch, from the run's own counts."""
    rb, pos, cand, slots = st["read_bases"], st["positions"], st["n_candidates"], st["column_slots"]
    if stage == "ms_capture":    # every quality byte + every packed base once, one 2-byte slot per pile cell kept; the read's
        # proposals ride on the same wave: its mismatch list in, a mask word per candidate
        return rb * 1.5 + slots * 2.0 + cs_bytes / 4.0 * 8.0 + cand * 8.0
    if stage == "ms_parse":      # cs text in, ~16 B per cs operation out (segments + mismatch list), one bitmap word per mark;
        # the decode's waves also leave the column store empty for the capture (2 B per slot)
        return cs_bytes * 1.0 + cs_bytes / 4.0 * 16.0 + slots * 2.0
    if stage == "ms_emit":       # the sweep reads the position bitmap and the marked mask cells, writes candidate + key
        return pos / 8.0 + cand * (2.0 + 16.0)
    if stage == "ms_eval":       # column slots in, one 64-byte record out
        return slots * 2.0 + cand * 64.0
    if stage == "ms_index":      # position bitmap in (twice: totals, then the table), block table out
        return 2.0 * pos / 8.0 + pos / 256.0 * 24.0
    if stage == "ms_finalize":   # records in and out, keys in
        return cand * (64.0 * 2 + 8.0)
    return 0.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--contig-len", type=int, default=CHR20_LEN)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--error-rate", type=float, default=None,
                    help="NOT the headline workload: substitution, insertion and deletion rate each of the synthetic reads "
                         "(generator defaults 2e-4 / 1e-4 / 1e-4), for the sensitivity table of DESIGN.md")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mb", type=float, default=24.0)
    ap.add_argument("--legs", default="queued,normcounts,edges,e2e",
                    help="the SURVEY 8f rows measured behind the headline on the single-GPU run (comma list; '' for none)")
    ap.add_argument("--legs-limit-s", type=float, default=240.0, help="watchdog over all the legs")
    return ap.parse_args()


def make_side_sets(sample, seed):
    from himut_amd import genome
    return genome.side_sets(sample, seed)


def _sub_batch(batch, i0, i1):
    """Reads i0 .. i1 of a batch as a batch of their own (views: seq / bq / cs stay whole, offsets are absolute)."""
    import numpy as np
    from himut_amd.readbatch import ReadBatch
    return ReadBatch(name=batch.name, length=batch.length, tstart=batch.tstart[i0:i1], tend=batch.tend[i0:i1],
                     qstart=batch.qstart[i0:i1], qlen=batch.qlen[i0:i1], mapq=batch.mapq[i0:i1], flag=batch.flag[i0:i1],
                     qid=np.arange(i1 - i0, dtype=np.int32), qoff=batch.qoff[i0:i1], cs_off=batch.cs_off[i0:i1 + 1],
                     seq=batch.seq, bq=batch.bq, cs=batch.cs, tp=batch.tp[i0:i1])


def cpu_baseline(batch, chunks, params, pon, com, sample_mb):
    """The CPU oracle (a C restatement of the reference algorithm, "port") timed on the host: one thread over a bounded
    prefix of the workload, then one thread per chunk range on every core the process may use over the whole contig
    (SURVEY 8d(2)).  The reference's own Python figure travels as a constant: the reference cannot run here."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    assert np.array_equal(batch.qid, np.arange(batch.n))      # synthetic reads have unique names
    limit = int(sample_mb * 1e6)
    sub_chunks = [c for c in chunks if c[1] <= limit] or chunks[:1]
    n = int(np.searchsorted(batch.tstart, sub_chunks[-1][1], side="left"))
    t0 = time.perf_counter()
    recs, log = O.call(_sub_batch(batch, 0, n), sub_chunks, params, 1 / (10 ** 3), pon, com)
    dt = time.perf_counter() - t0
    span = sum(e - s + 1 for s, e in sub_chunks)
    out = {"value": span / 1e6 / dt, "unit": "Mbp/s", "cores": 1, "kind": "port",
           "sample": "first {} reference chunks ({:.1f} Mb, {} reads) of the same contig, oracle/himut_oracle.c "
                     "single thread, {:.1f} s; candidate sites/s {:.0f}".format(len(sub_chunks), span / 1e6, n, dt, log[1] / dt)}
    # all cores: the contig's chunk list cut into one range per core, one oracle call per thread (ctypes drops the GIL)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64, len(chunks)))
    cuts = [round(k * len(chunks) / cores) for k in range(cores + 1)]

    def work(k):
        cs = chunks[cuts[k]:cuts[k + 1]]
        if not cs:
            return 0
        i0 = int(np.searchsorted(batch.tstart, cs[0][0] - 30_000, side="left"))       # reads are <= 25 kb
        i1 = int(np.searchsorted(batch.tstart, cs[-1][1], side="left"))
        _r, lg = O.call(_sub_batch(batch, i0, i1), cs, params, 1 / (10 ** 3), pon, com)
        return lg[1]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        cand = sum(pool.map(work, range(cores)))
    dt = time.perf_counter() - t0
    span = sum(e - s + 1 for s, e in chunks)
    out["all_cores"] = {"value": span / 1e6 / dt, "unit": "Mbp/s", "cores": cores, "host_cpu_count": os.cpu_count(),
                        "kind": "port", "candidate_sites_per_sec": cand / dt,
                        "sample": "the whole contig ({:.1f} Mb, {} chunks), one oracle thread per chunk range on {} "
                                  "cores, {:.1f} s".format(span / 1e6, len(chunks), cores, dt)}
    ref = os.path.join(ROOT, "tests", "golden", "config1_reference.json")
    if os.path.exists(ref):
        try:
            t = json.load(open(ref))
            if t.get("reference_seconds"):
                out["reference_python"] = {"value": t["length"] / 1e6 / t["reference_seconds"], "unit": "Mbp/s",
                                           "cores": 1, "kind": "reference",
                                           "sample": "sjin09/himut caller.get_somatic_substitutions on BASELINE config 1 "
                                                     "(1 Mb, 30x), build container, {:.1f} s (tests/golden/"
                                                     "config1_reference.py); a constant carried here, the reference "
                                                     "cannot travel to the GPU box".format(t["reference_seconds"])}
        except Exception:
            pass
    return out


# ---------------------------------------------------------------------------------------
# The SURVEY 8f rows, measured behind the headline on the single-GPU run (their own keys of the line; the headline's
# keys stay what they are).  Each leg: inputs resident in HBM, device time from the library's hipEvents on its stream,
# a roofline against the bytes the leg's design has to move, the oracle timed on a bounded sample beside it.

NORM_ALT_ORDER = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
# algorithmic HBM bytes per read base of one normcounts pass: the qualities read by the read pass
# (mean-quality filter, callable bits) and again by the sweep, the packed bases read by the sweep, the callable bit
# written and read
NORM_BYTES_PER_BASE = {"prepass": 1.0 + 1.0 / 8.0, "sweep": 1.0 + 0.5 + 1.0 / 8.0}


def _prefix_batch(batch, end):
    """The reads that start in front of reference position `end`, with the arrays cut to them."""
    import numpy as np
    from himut_amd.readbatch import ReadBatch
    n = max(1, int(np.searchsorted(batch.tstart, end, side="left")))
    tot = int(batch.qoff[n - 1] + ((int(batch.qlen[n - 1]) + 31) & ~31))
    return n, ReadBatch(name=batch.name, length=batch.length, tstart=batch.tstart[:n], tend=batch.tend[:n],
                        qstart=batch.qstart[:n], qlen=batch.qlen[:n], mapq=batch.mapq[:n], flag=batch.flag[:n],
                        qid=batch.qid[:n], qoff=batch.qoff[:n], cs_off=batch.cs_off[:n + 1], seq=batch.seq[:tot // 2],
                        bq=batch.bq[:tot], cs=batch.cs[:int(batch.cs_off[n])], tp=batch.tp[:n])


def _pmc(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def leg_normcounts(ctx, sample, chunks, params, pon, com, steps, cpu=True, cpu_sample_mb=16.0):
    """himut normcounts' per-contig worker (normcounts.py:206-421) on the resident contig: K passes of
    himut_run_normcounts."""
    import numpy as np
    from himut_amd import normcounts
    batch = sample.batch
    refseq = bytes(sample.ref)
    chars, cls = normcounts.tri_classes(refseq)
    ctx.set_reference(refseq, cls, len(chars))
    tab = normcounts.alt_order_table(NORM_ALT_ORDER)
    ctx.set_stage_timing(2)
    ctx.run_normcounts(tab)
    ctx.run_normcounts(tab)
    ms, pre, sweep, quad = [], [], [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run_normcounts(tab)
        st = ctx.stats()
        ms.append(st["ms_total"]); pre.append(st["ms_parse"] + st["ms_index"]); sweep.append(st["ms_eval"]); quad.append(st["ms_capture"])
    wall = (time.perf_counter() - t0) / steps
    ctx.set_stage_timing(1)
    st = ctx.stats()
    ccs, ref, log = ctx.normcounts()
    positions = sum(e - s for s, e in chunks)
    dev_s, sweep_s, quad_s = float(np.mean(ms)) * 1e-3, float(np.mean(sweep)) * 1e-3, float(np.mean(quad)) * 1e-3
    rb = st["read_bases"]
    by_pass = rb * (NORM_BYTES_PER_BASE["prepass"] + NORM_BYTES_PER_BASE["sweep"])
    by_sweep = rb * NORM_BYTES_PER_BASE["sweep"]
    t = _pmc("pmc_traffic_normcounts.json")
    out = {"metric": "Mbp swept/sec at 30x CCS (himut normcounts callable-tricount sweep)", "unit": "Mbp/s",
           "value": positions / 1e6 / wall, "ms_per_contig": wall * 1e3, "device_ms": dev_s * 1e3, "steps": steps,
           "stage_ms": {"decode_and_read_pass": float(np.mean(pre)), "sweep": float(np.mean(sweep)), "k_norm_quad": quad_s * 1e3},
           "callable_bases": log[13], "num_bases": log[1], "positions": positions,
           "roofline": {"bound": "hbm", "kernel": "k_norm_quad", "achieved": by_sweep / quad_s / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by_sweep / quad_s / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": by_sweep, "avg_launch_ms": quad_s * 1e3,
                        "traffic": (t or {}).get("k_norm_quad"),
                        "traffic_source": ("profiles/pmc_traffic_normcounts.json (" + str(t.get("collected")) + ")") if t else None},
           "roofline_pass": {"bound": "hbm", "bytes_per_pass": by_pass, "achieved": by_pass / dev_s / 1e9,
                             "frac": by_pass / dev_s / 1e9 / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "survey_convention_frac": rb * 5.5 / dev_s / 1e9 / HBM_PEAK_GBS,
                             "note": "bytes_per_pass = read bases x ({prepass} + {sweep}) B (qualities twice, packed bases once, the "
                                     "callable bit written and read); survey_convention = 5.5 B per read base (SURVEY 8d: "
                                     "1.5 in, a 2-byte cell written and read), a throughput in the survey's unit".format(**NORM_BYTES_PER_BASE)}}
    if cpu:
        from oracle import oracle as O
        nch = max(1, int(cpu_sample_mb * 1e6 / 200000))
        sub_chunks = chunks[:nch]
        n, sub = _prefix_batch(batch, sub_chunks[-1][1])
        t1 = time.perf_counter()
        O.normcounts(sub, sub_chunks, params, refseq, 1 / (10 ** 3), pon, com, alt_order=NORM_ALT_ORDER)
        dt = time.perf_counter() - t1
        span = sum(e - s for s, e in sub_chunks)
        out["cpu_baseline"] = {"value": span / 1e6 / dt, "unit": "Mbp/s", "cores": 1, "kind": "port",
                               "sample": "first {} reference chunks ({:.1f} Mb, {} reads), oracle/himut_oracle.c "
                                         "orc_normcounts single thread, {:.1f} s".format(nch, span / 1e6, n, dt)}
    return out


def leg_queued(ctx, sample, chunks, params, pon, com, steps, device):
    """The headline's step with the host out of the way: two contexts hold the same contig and their runs are queued back
    to back (himut_run_begin / himut_run_end), the way the genome driver queues its contigs -- the decode and the small
    kernels of one run fill the dispatch gaps and the tail of the other.  Throughput of the same work; the headline keeps
    its one-run-at-a-time figure."""
    import torch
    from himut_amd import caller
    w2 = caller.Worker(device)
    try:
        w2.configure(germline_snv_prior=1 / (10 ** 3), phase=False, **params)
        c2 = w2.ctx
        c2.set_chunks(chunks)
        c2.set_site_set(0, pon)
        c2.set_site_set(1, com)
        c2.push_reads(sample.batch)
        ctxs = [ctx, c2]
        for c in ctxs:
            c.run(); c.run()                                   # capacities kept from here on
        n_ref = ctx.stats()["n_records"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctxs[0].run_begin()
        for k in range(1, steps):
            ctxs[k & 1].run_begin()
            ctxs[(k - 1) & 1].run_end()
        ctxs[(steps - 1) & 1].run_end()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        same = c2.stats()["n_records"] == n_ref and ctx.stats()["n_records"] == n_ref
        reran = int(ctx.stats()["reran"]) + int(c2.stats()["reran"])
    finally:
        w2.close()
    positions = sum(e - s for s, e in chunks)
    return {"metric": "Mbp scanned/sec at 30x CCS, runs of two contexts on the same contig queued back to back", "unit": "Mbp/s",
            "value": positions / 1e6 / dt, "ms_per_step": dt * 1e3, "steps": steps, "records_equal": bool(same), "reran": reran,
            "note": "every run complete (decode, capture, evaluation, records) inside the timed region; what differs from the "
                    "headline is that run k + 1 is queued before run k is waited for (DESIGN.md section 7)"}


def leg_edges(ctx, sample, steps, cpu=True):
    """himut phase's pair counting (phaselib.get_edges, phaselib.py:16-67) on the resident contig."""
    import numpy as np
    from himut_amd import phaselib
    b = sample.batch
    hets = sorted(set((int(p) + 1, chr(r), chr(al)) for p, r, al, g in zip(sample.snp_pos, sample.snp_ref, sample.snp_alt,
                                                                             sample.snp_gt) if g in (1, 2)))
    hpos = np.array([h[0] for h in hets], np.int32)
    href = np.array([ord(h[1]) for h in hets], np.uint8)
    band = phaselib.edge_band(b, hpos)
    ctx.run_edges(hpos, href, 20, 20, band)
    ms = []
    for _ in range(steps):
        counts = ctx.run_edges(hpos, href, 20, 20, band)
        ms.append(ctx.stats()["ms_total"])
    dev_s = float(np.mean(ms)) * 1e-3
    pairs = int(counts.sum())
    # bytes the design has to move: the cs text in and ~16 B per cs operation out (the decode), per usable (read, hetSNP)
    # a 64-byte sector of bases and one of qualities, one 4-byte atomic per pair
    nhits = int(np.sum(np.searchsorted(hpos, b.tend, side="right") - np.searchsorted(hpos, b.tstart, side="right")))
    alg = float(b.cs.shape[0]) * 5.0 + nhits * 128.0 + pairs * 4.0
    out = {"metric": "Mbp pair-counted/sec at 30x CCS (himut phase get_edges)", "unit": "Mbp/s",
           "value": b.length / 1e6 / dev_s, "device_ms": dev_s * 1e3, "steps": steps, "hetsnps": len(hets), "band": int(band),
           "pair_counts": pairs, "read_hetsnp_lookups": nhits,
           "roofline": {"bound": "hbm", "kernel": "k_parse_cs + k_edges", "achieved": alg / dev_s / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": alg / dev_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg,
                        "avg_launch_ms": dev_s * 1e3, "traffic": (_pmc("pmc_traffic_edges.json") or {}).get("pass_total"),
                        "traffic_source": "profiles/pmc_traffic_edges.json" if _pmc("pmc_traffic_edges.json") else None,
                        "note": "scattered 64-byte sectors and atomics: bound by the random-access rate, not by bytes"}}
    if cpu:
        from oracle import oracle as O
        span = min(b.length, 64_000_000)                      # (about 2 s of one core: the whole of a chr20-sized contig)
        n, sub = _prefix_batch(b, span)
        sh = [h for h in hets if h[0] < span + 100_000]
        t0 = time.perf_counter()
        O.edges(sub, sh, 20, 20)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": span / 1e6 / dt, "unit": "Mbp/s", "cores": 1, "kind": "port",
                               "sample": "reads of the first {:.0f} Mb ({}), oracle orc_edges single thread, {:.1f} s".format(span / 1e6, n, dt)}
    return out


def leg_e2e(sample, device, repeat=2):
    """BAM on disk -> VCF on disk for the same contig (the ingest in front of the path): host inflate into
    pinned windows || H2D || device-side record parse, thresholds, the scan, records back, VCF text.  PCIe-inclusive: never
    `value` of the headline."""
    import tempfile
    import threading
    from himut_amd import bamio, bamlib, caller, synth, util as hutil, vcflib
    name = sample.batch.name
    runs = []
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.bam")
        t0 = time.perf_counter()
        bamio.write_bam(path, [sample.batch], sample="SMP")
        com, pon = os.path.join(d, "c.vcf"), os.path.join(d, "p.vcf")
        synth.write_common_snps_vcf(com, sample, seed=1)
        synth.write_pon_vcf(pon, sample, seed=1)
        t_write = time.perf_counter() - t0
        for rep in range(repeat):
            t = {}
            w = caller.Worker(device)
            ctx = w.ctx
            t_all = time.perf_counter()
            st = bamio.BamStream(path, 0)
            t["open_s"] = time.perf_counter() - t_all
            side = {}

            def parse_side():
                side["pk"] = caller.site_keys(vcflib.load_pon(name, pon))
                side["ck"] = caller.site_keys(vcflib.load_common_snp(name, com))
            th = threading.Thread(target=parse_side)
            th.start()
            t0 = time.perf_counter()
            res = st.ingest_contig(ctx, name)
            t["ingest_inflate_h2d_parse_s"] = time.perf_counter() - t0
            th.join()
            t0 = time.perf_counter()
            chrom_lst, c2c = hutil.load_loci(None, None, st.tname2tsize)
            ts, te, ql_, mq_, tp_ = ctx.ingest_read_meta(res["n_reads"])
            starts = bamlib.sample_starts(chrom_lst, st.tname2tsize)
            ql, qu, md = bamlib.thresholds_from_samples({name: bamlib.sample_qlens(ts, te, ql_, mq_, tp_, starts[name])}, chrom_lst)
            t["thresholds_s"] = time.perf_counter() - t0
            w.configure(30, 60, ql, qu, 0.99, 20, 93, 0.01, 0, 20, md, 3, 1, 3, 1e-3, False)
            t0 = time.perf_counter()
            ctx.set_chunks([(x[1], x[2]) for x in c2c[name]]); ctx.set_site_set(0, side["pk"]); ctx.set_site_set(1, side["ck"])
            ctx.run()
            t["first_run_s"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            recs = ctx.records()
            t["d2h_s"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            vcflib.dump_records(os.path.join(d, "o.vcf"), "#HEADER", [name], {name: recs}, False)
            t["format_write_s"] = time.perf_counter() - t0
            t["end_to_end_s"] = time.perf_counter() - t_all
            t["records"] = int(len(recs))
            t["inflated_GB_per_s"] = (res["bases_padded"] * 1.5 + res["cs_bytes"]) / 1e9 / t["ingest_inflate_h2d_parse_s"]
            w.close()
            st.close()
            runs.append(t)
        bam_mb = os.path.getsize(path) / 1e6
    best = min(runs, key=lambda r: r["end_to_end_s"])
    return {"metric": "Mbp/sec, BAM on disk -> VCF on disk (himut call, one contig)", "unit": "Mbp/s",
            "value": sample.batch.length / 1e6 / best["end_to_end_s"], "best_of": repeat, "stages_s": best,
            "first_in_process": runs[0]["end_to_end_s"], "bam_MB": bam_mb, "bam_write_s_untimed": t_write,
            "host_threads": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
            "note": "PCIe- and inflate-inclusive; the host's BGZF inflate is the floor (DESIGN.md section 11)"}


def run_legs(out, legs, a, ctx, w, sample, chunks, params, pon, com, device):
    """The 8f legs under one watchdog: a leg that raises is reported in its key; a leg that hangs ends the process with the
    line printed and a non-zero exit code."""
    import threading

    def give_up():
        out.setdefault("legs_error", "the legs did not finish within {:.0f} s".format(a.legs_limit_s))
        emit_line(out)
        os._exit(3)
    timer = threading.Timer(a.legs_limit_s, give_up)
    timer.daemon = True
    timer.start()
    cpu = not a.no_cpu_baseline
    steps = max(3, min(a.steps, 10))
    for leg in legs:
        t0 = time.perf_counter()
        try:
            if leg == "queued":
                r = leg_queued(ctx, sample, chunks, params, pon, com, max(20, a.steps), device)
            elif leg == "normcounts":
                r = leg_normcounts(ctx, sample, chunks, params, pon, com, steps, cpu)
            elif leg == "edges":
                r = leg_edges(ctx, sample, steps, cpu)
            elif leg == "e2e":
                w.close()               # (the leg makes its own contexts: a real call starts from the file)
                r = leg_e2e(sample, device)
            else:
                r = {"error": "unknown leg"}
        except Exception as e:          # noqa: BLE001 -- reported in the line, the headline stands
            r = {"error": "{}: {}".format(type(e).__name__, e)}
        r["leg_wall_s"] = time.perf_counter() - t0
        out[leg] = r
    timer.cancel()


def emit_line(out):
    """The one JSON line, on the process's real stdout (see main)."""
    os.write(_REAL_STDOUT, (json.dumps(out) + "\n").encode())


_REAL_STDOUT = 1


def main():
    global _REAL_STDOUT
    a = parse()
    # Exactly one line goes to stdout: libraries that greet on file descriptor 1 (RCCL's version banner, gloo's
    # "connected to peer ranks") write to stderr from here on, the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch  # before libhimut_hip.so: both must bind to one HIP runtime
    import torch.distributed as dist
    import numpy as np
    # rehearsal knobs (not used by the driver): HIMUT_BENCH_BACKEND=gloo runs the N > 1 code path
    # on a box with one GPU, every rank on cuda:0
    backend = os.environ.get("HIMUT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from himut_amd import bamlib, caller, synth, util as hutil
    from himut_amd import dist as hdist

    # ---- synthetic workload (BASELINE.json configs[1] shape; one contig per rank)
    t_gen = time.perf_counter()
    names = ["chr{}".format(20 + k) for k in range(world)]   # one chr20-sized contig per rank
    cfg = synth.SynthConfig(seed=2 + rank, contig_len=a.contig_len, depth=a.depth, name=names[rank])
    if a.error_rate is not None:
        cfg.sub_rate = cfg.ins_rate = cfg.del_rate = a.error_rate
    legs = [x for x in a.legs.split(",") if x] if world == 1 else []
    sample = synth.generate(cfg, want_ref="normcounts" in legs)
    batch = batch_keep = sample.batch
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((batch.name, 0, batch.length))]
    ql, qu, md = bamlib.get_thresholds({batch.name: batch}, [batch.name], {batch.name: batch.length})
    pon, com = make_side_sets(sample, 100 + rank)
    cs_bytes = int(batch.cs.shape[0])
    params = dict(min_qv=30, min_mapq=60, qlen_lower_limit=ql, qlen_upper_limit=qu, min_sequence_identity=0.99,
                  min_gq=20, min_bq=93, min_trim=0.01, max_mismatch_count=0, mismatch_window_size=20, md_threshold=md,
                  min_ref_count=3, min_alt_count=1, min_hap_count=3)
    t_gen = time.perf_counter() - t_gen

    w = caller.Worker(local_rank)
    w.configure(germline_snv_prior=1 / (10 ** 3), phase=False, **params)
    ctx = w.ctx
    ctx.set_chunks(chunks)
    ctx.set_site_set(0, pon)
    ctx.set_site_set(1, com)
    t_h2d = time.perf_counter()
    ctx.push_reads(batch)                     # inputs resident in HBM before the timed region
    t_h2d = time.perf_counter() - t_h2d

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: the final exchange of every step (record buffers of the step's contigs -> rank 0, RCCL over xGMI)
    # is pipelined: it runs while the next step scans.  All K exchanges complete inside the timed region.
    ex = None
    use_ex = world > 1 or os.environ.get("HIMUT_BENCH_FORCE_EXCHANGE") == "1"   # rehearsal of the N > 1 path on one rank
    if use_ex:
        if world == 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend, rank=0, world_size=1)
        ctx.run()                                           # untimed: sizes the exchange buffers
        cap = hdist.RecordExchange.plan(ctx.records_device()[1])
        ex = hdist.RecordExchange(rank, world, cap, depth=2)

    def step():
        ctx.run()
        if use_ex:
            _, n = ctx.records_device()
            if backend == "nccl":
                ex.submit(n, ctx.log(), ctx=ctx)
            else:
                ex.submit(n, ctx.log(), records=ctx.records())

    for _ in range(a.warmup):
        step()
    if ex is not None:
        ex.drain()
        ex.meta = []
    stage_ms = {}
    for attempt in range(3):
        barrier()
        t0 = time.perf_counter()
        # timed region: the library records only the run's start / end and the events around the column capture
        # (the dominant kernel, whose launch time the roofline needs); an event costs a barrier packet on the queue
        stage_ms = {"ms_total": [], "ms_capture": [], "reran": []}
        for _ in range(a.steps):
            step()
            tot_, cap_, rr_ = ctx.stats_brief()
            stage_ms["ms_total"].append(tot_); stage_ms["ms_capture"].append(cap_); stage_ms["reran"].append(rr_)
        try:
            gathered = ex.drain() if ex is not None else None
        except hdist.RecordExchangeOverflow as e:      # (every rank alike) more records than the warm-up pass had: once more
            if attempt == 2:
                raise
            ex = hdist.RecordExchange(rank, world, e.caps, depth=2)
            continue
        barrier()
        elapsed = time.perf_counter() - t0
        break
    # untimed: three more passes with every stage event on, for the per-stage breakdown
    ctx.set_stage_timing(2)
    detail = {}
    for _ in range(3):
        ctx.run()
        st = ctx.stats()
        for k in ("ms_total", "ms_parse", "ms_hap", "ms_emit", "ms_index", "ms_capture", "ms_eval", "ms_finalize"):
            detail.setdefault(k, []).append(st[k])
    ctx.set_stage_timing(1)
    if gathered is not None:                                # untimed: host copy of the last step's gathered records
        gathered = (gathered[0], ex.last_records(gathered[0]))
    red_dev = "cuda" if backend == "nccl" else "cpu"
    if use_ex:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    st = ctx.stats()
    log = ctx.log()
    reran_steps = int(sum(stage_ms.get("reran", [0])))
    totals = torch.tensor([st["positions"], log[1], st["read_bases"], st["n_records"]], dtype=torch.float64,
                          device=red_dev)
    if use_ex:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    positions, cand_sites, read_bases, n_records = [float(x) for x in totals.tolist()]

    want_genome = world > 1 or os.environ.get("HIMUT_BENCH_GENOME") == "1"
    out = None
    if rank == 0:
        if gathered is not None:   # every step's exchange delivered every rank's records
            counts, last = gathered
            assert len(counts) == world and all(len(c) == a.steps for c in counts)
            assert sum(int(c[-1][0]) for c in counts) == int(n_records)
            assert sum(len(v) for v in last.values()) == int(n_records)
        ms_per_step = elapsed / a.steps * 1e3
        mbp_s = positions / 1e6 / (elapsed / a.steps)
        avg = {k: float(np.mean(v)) for k, v in detail.items()}
        timed = {k: float(np.mean(v)) for k, v in stage_ms.items()}
        dom = max(STAGE_KERNEL, key=lambda k: avg[k])           # the dominant kernel of the step
        dom_ms = timed[dom] if dom in timed else avg[dom]       # the column capture: measured inside the timed region
        alg_bytes = algorithmic_bytes(dom, st, cs_bytes)
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        per_kernel = {STAGE_KERNEL[k]: {"ms": avg[k], "alg_GBps": algorithmic_bytes(k, st, cs_bytes) / (avg[k] * 1e-3) / 1e9}
                      for k in STAGE_KERNEL}
        # HBM traffic from the PMC counters: not collected in this run (counters need their own rocprofv3 passes) but
        # read from the committed summary of the same workload, profiles/pmc_traffic.json (profiles/collect.sh)
        traffic, traffic_step, traffic_source = None, None, None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                t = json.load(open(tp))
                if t.get("contig_len") == a.contig_len and t.get("depth") == a.depth:
                    traffic = t.get(STAGE_KERNEL[dom])
                    traffic_step = t.get("step_total")
                    traffic_source = "profiles/pmc_traffic.json (" + str(t.get("collected", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes")) + ")"
            except Exception:
                traffic = None
        all_stages = ("ms_parse", "ms_index", "ms_capture", "ms_emit", "ms_eval", "ms_finalize")
        moved = sum(algorithmic_bytes(k, st, cs_bytes) for k in all_stages)
        step_s = timed["ms_total"] * 1e-3
        out = {
            "metric": "Mbp scanned/sec at 30x CCS (himut call pileup scan)", "value": mbp_s, "unit": "Mbp/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32 + f64 genotype tail",
            "data": "synthetic",
            "config": {"workload": "chr20-sized contig ({} bp) {:.0f}x synthetic CCS per GPU, common-SNP + PoN "
                                   "filtering, reference chunking ({} chunks)".format(a.contig_len, a.depth,
                                                                                        len(chunks)) +
                                   ("" if a.error_rate is None else "; NOT the default reads: error rates {:g} each".format(a.error_rate)),
                       "reads_per_gpu": st["n_reads"], "read_bases_per_gpu": st["read_bases"],
                       "parallelism": "contig-per-gpu x{} + RCCL gather".format(world)},
            "candidate_sites_per_sec": cand_sites / (elapsed / a.steps),
            "candidate_sites_per_step": cand_sites, "records_per_step": n_records,
            "device_ms_per_step": timed["ms_total"],
            "stage_ms": avg, "stage_ms_note": "three untimed passes after the timed region with every stage event "
                                              "recorded (himut_set_stage_timing 2); the timed steps record run "
                                              "start / end and the events around k_stream_capture only",
            "kernels": per_kernel,
            "setup_s": {"generate": t_gen, "h2d": t_h2d},
            # The whole step, two ways.  "moved": the bytes this design has to move (sum of the per-kernel algorithmic
            # bytes of DESIGN.md section 4) over the step's device time -- the achieved-bandwidth figure; beside it the
            # same with the PMC counter bytes of the committed profile.  "survey_convention": the SURVEY's own count
            # (3.5 B per read base + 2 B per position + (1.5 D + 48) B per candidate site), which prices
            # a dense two-pass pile this design does not build: a throughput in the survey's unit, not a bandwidth.
            "roofline_step": {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "device_ms": timed["ms_total"],
                "moved": {"bytes_per_step": moved, "achieved": moved / step_s / 1e9, "frac": moved / step_s / 1e9 / HBM_PEAK_GBS,
                          "counter_bytes_per_step": traffic_step,
                          "counter_frac": (traffic_step / step_s / 1e9 / HBM_PEAK_GBS) if traffic_step else None},
                "survey_convention": (lambda by: {"bytes_per_step": by, "achieved": by / step_s / 1e9,
                                                  "frac": by / step_s / 1e9 / HBM_PEAK_GBS})(
                    (read_bases * 3.5 + positions * 2.0 + cand_sites * (1.5 * a.depth + 48.0)) / world)},
            "roofline": {"bound": "hbm", "kernel": STAGE_KERNEL[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms},
            # steps of the timed region in which himut_run repeated itself with exact buffer sizes (a count exceeded the
            # capacities kept from the run before): 0 expected after the warm-up
            "reran_steps": reran_steps,
        }
        if not a.no_cpu_baseline and world == 1:           # the CPU leg runs on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(batch, chunks, params, pon, com, a.cpu_sample_mb)
        if legs:
            run_legs(out, legs, a, ctx, w, sample, chunks, params, pon, com, local_rank)
            w = None if "e2e" in legs else w

    # ---- BASELINE configs[2]: the whole synthetic GRCh38, strong scaling (N > 1; HIMUT_BENCH_GENOME=1 forces it on
    # one rank, HIMUT_BENCH_GENOME_SCALE divides the contig lengths for rehearsals).  It runs behind the headline
    # measurement and under a watchdog: whatever happens in it, the headline line is printed.
    if want_genome:
        import threading
        from himut_amd import genome
        w.close()
        w = None
        del sample, batch_keep
        batch = None
        limit = float(os.environ.get("HIMUT_BENCH_GENOME_LIMIT_S", "420"))

        def give_up(why):
            if rank == 0:
                out["genome_strong"] = {"error": why}
                emit_line(out)
            os._exit(3)         # (the other ranks may sit in a collective: no teardown; non-zero: the run record shows it)
        timer = threading.Timer(limit, give_up, args=("the genome leg did not finish within {:.0f} s".format(limit),))
        timer.daemon = True
        timer.start()
        try:
            genome_strong = genome.run_genome(rank, world, local_rank, scale=float(os.environ.get("HIMUT_BENCH_GENOME_SCALE", "1")),
                                              depth=a.depth, steps=max(1, min(a.steps, 3)), backend=backend)
        except Exception as e:      # noqa: BLE001 -- reported in the line, the headline stands
            timer.cancel()
            give_up("{}: {}".format(type(e).__name__, e))
        timer.cancel()
        if rank == 0:
            out["genome_strong"] = genome_strong
    if rank == 0:
        emit_line(out)
    if w is not None:
        w.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
