#!/usr/bin/env python3
"""The normcounts sweep (k_norm_plan + k_norm_quad) against k_norm_tile on contigs of growing size: the same counts.
usage: python tools/norm_sizes.py 2e6 8e6 ...   (each size in a process of its own is the caller's business)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from himut_amd import bamlib, caller, normcounts, synth, util as hutil
import bench as B

for arg in sys.argv[1:]:
    L = int(float(arg))
    s = synth.generate(synth.SynthConfig(seed=2, contig_len=L, name="chr20"), want_ref=True)
    b = s.batch
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
    ql, qu, md = bamlib.get_thresholds({b.name: b}, [b.name], {b.name: b.length})
    w = caller.Worker(0)
    w.configure(30, 60, ql, qu, 0.99, 20, 93, 0.01, 0, 20, md, 3, 1, 3, 1e-3, False)
    refseq = bytes(s.ref)
    chars, cls = normcounts.tri_classes(refseq)
    w.ctx.set_chunks(chunks); w.ctx.set_reference(refseq, cls, len(chars)); w.ctx.push_reads(b)
    tab = normcounts.alt_order_table(B.NORM_ALT_ORDER)
    out = {}
    for sweep in (1, 0):
        w.ctx.debug_normcounts(sweep=sweep)
        t0 = time.time()
        w.ctx.run_normcounts(tab)
        st = w.ctx.stats()
        ccs, ref, log = w.ctx.normcounts()
        out[sweep] = (ccs.tolist(), ref.tolist(), log)
        print(L, "sweep", sweep, "ms", round(st["ms_total"], 3), "eval", round(st["ms_eval"], 3), "redo", st["column_slots"], "reran", st["reran"], log[:4], flush=True)
    print(L, "equal" if out[0] == out[1] else "DIFFERENT", flush=True)
    w.close()
