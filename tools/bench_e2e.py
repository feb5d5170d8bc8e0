#!/usr/bin/env python3
"""End-to-end `himut call` from a BAM file on disk to the VCF, stage by stage: open (header, BGZF block table, index),
ingest (host inflate into pinned windows || H2D || device-side record parse), thresholds, side VCFs, the scan (the
FIRST run on the fresh reads, allocations included: a real call makes one run per contig), record download, VCF text.
Writes a synthetic 30x BAM (+ .bai) for a contig of --contig-len first (untimed).  --host-ingest times the round-1 path
(host parse into arrays, pageable H2D) beside it."""
import argparse
import json
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contig-len", type=int, default=64_444_167)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--host-ingest", action="store_true")
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    import numpy as np
    from himut_amd import bamio, bamlib, caller, synth, util as hutil, vcflib
    s = synth.generate(synth.SynthConfig(seed=3, contig_len=a.contig_len, name="chr20"))
    out_lines = []
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.bam")
        bamio.write_bam(path, [s.batch], sample="SMP")
        com = os.path.join(d, "c.vcf")
        pon = os.path.join(d, "p.vcf")
        synth.write_common_snps_vcf(com, s, seed=1)
        synth.write_pon_vcf(pon, s, seed=1)
        bases = s.batch.total_read_bases()
        del s
        for rep in range(a.repeat):
            t = {"mode": "host-ingest" if a.host_ingest else "device-ingest", "repeat": rep}
            w = caller.Worker(0)
            ctx = w.ctx
            t_all = time.perf_counter()
            t0 = time.perf_counter()
            if a.host_ingest:
                bam = bamio.BamFile(path, a.threads)
                t["ingest_host_parse_s"] = time.perf_counter() - t0
                b = bam.batches["chr20"]
                tname2tsize = bam.tname2tsize
                t0 = time.perf_counter()
                ctx.push_reads(b)
                t["h2d_pageable_s"] = time.perf_counter() - t0
                t0 = time.perf_counter()
                chrom_lst, c2c = hutil.load_loci(None, None, tname2tsize)
                ql, qu, md = bamlib.get_thresholds(bam.batches, chrom_lst, tname2tsize)
                t["thresholds_s"] = time.perf_counter() - t0
            else:
                st = bamio.BamStream(path, a.threads)
                t["open_s"] = time.perf_counter() - t0
                t["indexed"] = st.indexed
                tname2tsize = st.tname2tsize
                # the side VCFs are parsed by a second host thread while the ingest (ctypes calls, no GIL) runs
                side = {}

                def parse_side():
                    t1 = time.perf_counter()
                    side["pk"] = caller.site_keys(vcflib.load_pon("chr20", pon))
                    side["ck"] = caller.site_keys(vcflib.load_common_snp("chr20", com))
                    side["s"] = time.perf_counter() - t1
                th = threading.Thread(target=parse_side)
                th.start()
                t0 = time.perf_counter()
                res = st.ingest_contig(ctx, "chr20")
                t["ingest_inflate_h2d_parse_s"] = time.perf_counter() - t0
                th.join()
                t["side_vcfs_beside_ingest_s"] = side["s"]
                t0 = time.perf_counter()
                chrom_lst, c2c = hutil.load_loci(None, None, tname2tsize)
                ts, te, ql_, mq_, tp_ = ctx.ingest_read_meta(res["n_reads"])
                starts = bamlib.sample_starts(chrom_lst, tname2tsize)
                ql, qu, md = bamlib.thresholds_from_samples({"chr20": bamlib.sample_qlens(ts, te, ql_, mq_, tp_, starts["chr20"])}, chrom_lst)
                t["thresholds_s"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            if a.host_ingest:
                pk = caller.site_keys(vcflib.load_pon("chr20", pon))
                ck = caller.site_keys(vcflib.load_common_snp("chr20", com))
            else:
                pk, ck = side["pk"], side["ck"]
            t["side_vcfs_s"] = time.perf_counter() - t0
            w.configure(30, 60, ql, qu, 0.99, 20, 93, 0.01, 0, 20, md, 3, 1, 3, 1e-3, False)
            chunks = [(x[1], x[2]) for x in c2c["chr20"]]
            t0 = time.perf_counter()
            ctx.set_chunks(chunks); ctx.set_site_set(0, pk); ctx.set_site_set(1, ck)
            ctx.run()
            t["first_run_s"] = time.perf_counter() - t0
            t["first_run_device_ms"] = ctx.stats()["ms_total"]
            t0 = time.perf_counter()
            recs = ctx.records()
            t["d2h_s"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            out = os.path.join(d, "o.vcf")
            vcflib.dump_records(out, "#HEADER", ["chr20"], {"chr20": recs}, False)
            t["format_write_s"] = time.perf_counter() - t0
            tot = time.perf_counter() - t_all
            ctx.run()
            t["steady_run_device_ms"] = ctx.stats()["ms_total"]
            t["records"] = len(recs)
            t["bam_MB"] = os.path.getsize(path) / 1e6
            t["read_Gbases"] = bases / 1e9
            t["contig_Mb"] = a.contig_len / 1e6
            t["end_to_end_s"] = tot
            t["end_to_end_Mbp_per_s"] = a.contig_len / 1e6 / tot
            w.close()
            if not a.host_ingest:
                st.close()
            print(json.dumps(t), flush=True)


if __name__ == "__main__":
    main()
