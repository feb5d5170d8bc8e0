#!/usr/bin/env python3
"""End-to-end `himut call` from a BAM file on disk to the VCF: ingest, H2D, kernels, D2H, formatting, broken out.
Writes a synthetic 30x BAM for a contig of --contig-len first (untimed)."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contig-len", type=int, default=32_000_000)
    a = ap.parse_args()
    from himut_amd import bamio, bamlib, caller, synth, util as hutil, vcflib
    s = synth.generate(synth.SynthConfig(seed=3, contig_len=a.contig_len, name="chr20"))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.bam")
        bamio.write_bam(path, [s.batch], sample="SMP")
        com = os.path.join(d, "c.vcf")
        pon = os.path.join(d, "p.vcf")
        synth.write_common_snps_vcf(com, s, seed=1)
        synth.write_pon_vcf(pon, s, seed=1)
        del s
        t = {}
        t0 = time.perf_counter()
        bam = bamio.BamFile(path)
        t["ingest_s"] = time.perf_counter() - t0
        b = bam.batches["chr20"]
        t0 = time.perf_counter()
        chrom_lst, c2c = hutil.load_loci(None, None, bam.tname2tsize)
        ql, qu, md = bamlib.get_thresholds(bam.batches, chrom_lst, bam.tname2tsize)
        pk = caller.site_keys(vcflib.load_pon("chr20", pon))
        ck = caller.site_keys(vcflib.load_common_snp("chr20", com))
        t["host_prep_s"] = time.perf_counter() - t0
        w = caller.Worker(0)
        w.configure(30, 60, ql, qu, 0.99, 20, 93, 0.01, 0, 20, md, 3, 1, 3, 1e-3, False)
        chunks = [(x[1], x[2]) for x in c2c["chr20"]]
        ctx = w.ctx
        ctx.set_chunks(chunks); ctx.set_site_set(0, pk); ctx.set_site_set(1, ck)
        t0 = time.perf_counter()
        ctx.push_reads(b)
        t["h2d_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        ctx.run()
        t["first_run_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        ctx.run()
        t["run_s"] = time.perf_counter() - t0
        t["device_ms"] = ctx.stats()["ms_total"]
        t0 = time.perf_counter()
        recs = ctx.records()
        t["d2h_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = os.path.join(d, "o.vcf")
        vcflib.dump_records(out, "#HEADER", ["chr20"], {"chr20": recs}, False)
        t["format_write_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        tup = caller.records_to_tuples("chr20", recs)
        vcflib.dump_sbs(os.path.join(d, "o2.vcf"), "#HEADER", ["chr20"], {"chr20": tup})
        t["format_write_python_s"] = time.perf_counter() - t0
        assert open(out).read() == open(os.path.join(d, "o2.vcf")).read()
        t["records"] = len(recs)
        t["bam_MB"] = os.path.getsize(path) / 1e6
        t["read_Gbases"] = b.total_read_bases() / 1e9
        t["contig_Mb"] = a.contig_len / 1e6
        tot = t["ingest_s"] + t["host_prep_s"] + t["h2d_s"] + t["run_s"] + t["d2h_s"] + t["format_write_s"]
        t["end_to_end_s"] = tot
        t["end_to_end_Mbp_per_s"] = a.contig_len / 1e6 / tot
    print(json.dumps(t))


if __name__ == "__main__":
    main()
