#!/usr/bin/env python3
"""The normcounts sweep under --phase on one GPU: a chr20-sized 30x contig with a phased germline VCF (blocks of 40 hetSNPs),
K timed passes of himut_run_normcounts.  Prints one JSON line (device ms per contig, stages).

    python tools/bench_normcounts_phase.py [--contig-len N] [--steps K]
"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--contig-len", type=int, default=64_444_167)
    ap.add_argument("--call", action="store_true", help="time himut_run (the call path) on the same contig and chunks instead")
    ap.add_argument("--no-phase", action="store_true", help="the same chunks (the phase blocks) without --phase: what the chunking costs by itself")
    a = ap.parse_args()
    import numpy as np
    from himut_amd import bamlib, caller, normcounts, synth, vcflib
    import bench as B
    s = synth.generate(synth.SynthConfig(seed=2, contig_len=a.contig_len, depth=30.0, snp_rate=1e-3, name="chr20"), want_ref=True)
    b = s.batch
    with tempfile.TemporaryDirectory() as d:
        pv = os.path.join(d, "p.vcf")
        synth.write_phased_vcf(pv, s, block=40)
        hb, hp, hs, c2c = vcflib.load_phased_hetsnps(pv, [b.name], {b.name: b.length})
    chunks = [(c[1], c[2]) for c in c2c[b.name]]
    ql, qu, md = bamlib.get_thresholds({b.name: b}, [b.name], {b.name: b.length})
    pon, com = B.make_side_sets(s, 100)
    params = dict(min_qv=30, min_mapq=60, qlen_lower_limit=ql, qlen_upper_limit=qu, min_sequence_identity=0.99,
                  min_gq=20, min_bq=93, min_trim=0.01, max_mismatch_count=0, mismatch_window_size=20, md_threshold=md,
                  min_ref_count=3, min_alt_count=1, min_hap_count=3)
    w = caller.Worker(0)
    w.configure(germline_snv_prior=1 / (10 ** 3), phase=not a.no_phase, **params)
    ctx = w.ctx
    ctx.set_chunks(chunks)
    ctx.set_site_set(0, pon)
    ctx.set_site_set(1, com)
    ctx.push_reads(b)
    if not a.no_phase:
        ctx.set_phase(*caller.pack_phase_sets(chunks, dict(hb[b.name]), dict(hp[b.name]), dict(hs[b.name])))
    if a.call:
        ctx.set_stage_timing(2)
        rows = []
        for k in range(a.steps + 2):
            ctx.run()
            if k >= 2:
                rows.append(ctx.stats())
        keys = [k for k in rows[0] if k.startswith("ms_")]
        print(json.dumps({"chunks": len(chunks), "phase": not a.no_phase, "records": int(rows[-1]["n_records"]),
                          **{k: float(np.mean([r[k] for r in rows])) for k in keys}}))
        return
    refseq = bytes(s.ref)
    chars, cls = normcounts.tri_classes(refseq)
    ctx.set_reference(refseq, cls, len(chars))
    tab = normcounts.alt_order_table(B.NORM_ALT_ORDER)
    ctx.set_stage_timing(2)
    ms, quad = [], []
    for k in range(a.steps + 2):
        ctx.run_normcounts(tab)
        st = ctx.stats()
        if k >= 2:
            ms.append(st["ms_total"]); quad.append(st["ms_capture"])
    ccs, ref, log = ctx.normcounts()
    span = [e - s_ for s_, e in chunks]
    print(json.dumps({"chunks": len(chunks), "positions": int(sum(span)), "longest_chunk": int(max(span)), "phase": not a.no_phase, "device_ms": float(np.mean(ms)), "k_norm_quad_ms": float(np.mean(quad)),
                      "callable_bases": int(log[13]), "reran": int(st["reran"]), "tiles_left_to_k_norm_tile": int(st["column_slots"])}))


if __name__ == "__main__":
    main()
