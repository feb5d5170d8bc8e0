#!/usr/bin/env python3
"""Randomised parity of the HIP paths with the oracle (test infrastructure: the oracle is the checker).  Every round draws a
synthetic contig (length, depth, error rates, read lengths, soft clips, noisy / low-quality / low-mapq reads, qualities up
to 255 in some), thresholds, a chunking (the reference's 200 kb tiles or random regions) and --phase or not, runs himut call's
worker and the normcounts sweep on the GPU and the oracle on the CPU, and compares records, counters and counts bit for bit.
Prints one line per round and stops at the first difference with the seed that reproduces it.

    python tools/fuzz_parity.py --rounds 40 --seed 1 [--minutes 8]
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=8.0)
    ap.add_argument("--scale", type=float, default=1.0, help="multiplies the contig lengths drawn (30 .. 260 kb at 1)")
    ap.add_argument("--start", type=int, default=0, help="skip the rounds in front of this one (the seeds stay what they are)")
    ap.add_argument("--verbose", action="store_true", help="say where a round differs")
    ap.add_argument("--only", type=int, default=None, help="run the one round with this seed and say where it differs")
    a = ap.parse_args()
    import numpy as np
    from himut_amd import caller, normcounts, synth, util as hutil, vcflib
    from himut_amd.readbatch import ReadBatch
    from oracle import oracle as O
    from tests import util
    w = caller.Worker(0)
    order = {"A": ["T", "G", "C"], "T": ["C", "A", "G"], "G": ["A", "C", "T"], "C": ["G", "T", "A"]}
    t_end = time.time() + a.minutes * 60
    for rnd in range(a.start, a.rounds) if a.only is None else [0]:
        if time.time() > t_end:
            break
        seed = a.seed * 1000 + rnd if a.only is None else a.only
        rs = np.random.RandomState(seed)
        lg = lambda lo, hi: float(np.exp(rs.uniform(np.log(lo), np.log(hi))))
        phase = bool(rs.rand() < 0.3)
        rl = float(rs.choice([6000, 10000, 15000, 20000]))
        cfg = synth.SynthConfig(seed=seed, contig_len=int(rs.randint(30_000, 260_000) * a.scale), depth=float(rs.choice([6, 15, 30, 45, 80])),
                                sub_rate=lg(1e-5, 3e-3), ins_rate=lg(1e-5, 2e-3), del_rate=lg(1e-5, 2e-3), som_rate=lg(1e-5, 5e-4),
                                snp_rate=lg(2e-4, 3e-3), read_len_mean=rl, read_len_sd=rl / 6, read_len_min=int(rl / 3), read_len_max=int(rl * 1.7),
                                frac_noisy=float(rs.choice([0, 0.05, 0.3])), frac_lowbq=float(rs.choice([0, 0.1])),
                                frac_lowmapq=float(rs.choice([0, 0.1])), frac_softclip=float(rs.choice([0, 0.3])), softclip_max=3000,
                                hetalt_frac=float(rs.choice([0, 0.05])), cs_long=bool(rs.rand() < 0.15),
                                pile_frac=float(rs.choice([0, 0, 0.05])), pile_mult=float(rs.choice([3, 6])), name="chrF")
        if rs.rand() < 0.1:
            cfg.depth = 150.0
            cfg.contig_len = min(cfg.contig_len, 90_000)
        s = synth.generate(cfg, want_ref=True)
        b = s.batch
        if rs.rand() < 0.3:                                    # qualities of 94 .. 255 in some reads
            bq = b.bq.copy()
            idx = rs.randint(0, len(bq), 5000)
            bq[idx] = rs.randint(94, 256, 5000)
            b = ReadBatch(name=b.name, length=b.length, tstart=b.tstart, tend=b.tend, qstart=b.qstart, qlen=b.qlen, mapq=b.mapq,
                          flag=b.flag, qid=b.qid, qoff=b.qoff, cs_off=b.cs_off, seq=b.seq, bq=bq, cs=b.cs, tp=b.tp)
        zq = rs.rand() < 0.12                                    # a few zero qualities (ValueError in a candidate column)
        nb = rs.rand() < 0.08 and not cfg.cs_long                # a few bases outside ATGC (KeyError where aligned and fetched)
        if zq or nb:
            bq, seq = b.bq.copy(), b.seq.copy()
            if zq:
                bq[rs.randint(0, len(bq), int(rs.choice([1, 20, 400])))] = 0
            if nb:
                k = rs.randint(0, len(seq), int(rs.choice([1, 5])))
                seq[k] = (seq[k] & 0x0f) | 0xf0
            b = ReadBatch(name=b.name, length=b.length, tstart=b.tstart, tend=b.tend, qstart=b.qstart, qlen=b.qlen, mapq=b.mapq,
                          flag=b.flag, qid=b.qid, qoff=b.qoff, cs_off=b.cs_off, seq=seq, bq=bq, cs=b.cs, tp=b.tp)
        p = dict(util.CALL_DEFAULTS)
        p.update(qlen_lower_limit=int(rl / 2.5), qlen_upper_limit=int(rl * 1.6), md_threshold=int(rs.choice([20, 60, 400])),
                 min_bq=int(rs.choice([0, 20, 60, 93, 93, 93, 150])), min_qv=int(rs.choice([0, 20, 30])),
                 min_sequence_identity=float(rs.choice([0.0, 0.9, 0.99])), min_trim=float(rs.choice([0.0, 0.01, 0.1])),
                 max_mismatch_count=int(rs.choice([0, 0, 1, 4])), mismatch_window_size=int(rs.choice([0, 10, 20, 60])),
                 min_gq=int(rs.choice([0, 20, 60])), min_ref_count=int(rs.choice([0, 3, 10])), min_alt_count=int(rs.choice([1, 2])),
                 min_hap_count=int(rs.choice([0, 3])))
        phase_sets = None
        if phase:
            with tempfile.TemporaryDirectory() as d:
                pv = os.path.join(d, "p.vcf")
                synth.write_phased_vcf(pv, s, block=int(rs.choice([8, 40])))
                hb, hp, hs, c2c = vcflib.load_phased_hetsnps(pv, [b.name], {b.name: b.length})
            if b.name not in c2c or not c2c[b.name]:
                phase = False
            else:
                phase_sets = (dict(hb[b.name]), dict(hp[b.name]), dict(hs[b.name]))
                chunks = [(c[1], c[2]) for c in c2c[b.name]]
        if not phase:
            if rs.rand() < 0.5:
                chunks = [(c[1], c[2]) for c in hutil.chunkloci((b.name, 0, b.length))]
            else:                                              # random regions in order (they may touch or overlap)
                cuts = np.sort(rs.randint(0, b.length, 2 * int(rs.randint(1, 6))))
                chunks = [(int(cuts[i]), int(max(cuts[i + 1], cuts[i] + 1))) for i in range(0, len(cuts), 2)]
        sites = [(int(x) + 1, chr(r), chr(al)) for x, r, al in zip(s.snp_pos, s.snp_ref, s.snp_alt)]
        pon = O.site_keys(sites[::3])
        com = O.site_keys(sites[1::3])
        what = "seed {} len {} depth {:.0f} sub {:.1e} ins {:.1e} del {:.1e} readlen {:.0f} chunks {} phase {} min_bq {} win {}/{}".format(
            seed, b.length, cfg.depth, cfg.sub_rate, cfg.ins_rate, cfg.del_rate, rl, len(chunks), phase, p["min_bq"],
            p["max_mismatch_count"], p["mismatch_window_size"])
        t0 = time.time()
        # ---- the call path
        orecs = oerr = None
        try:
            orecs, olog = O.call(b, chunks, p, p["germline_snv_prior"], pon, com, phase_sets)
        except Exception as e:                                  # noqa: BLE001 -- the reference raises: so must the library
            oerr = type(e).__name__ + ": " + str(e)[:120]
        herr = None
        try:
            w.configure(p["min_qv"], p["min_mapq"], p["qlen_lower_limit"], p["qlen_upper_limit"], p["min_sequence_identity"],
                        p["min_gq"], p["min_bq"], p["min_trim"], p["max_mismatch_count"], p["mismatch_window_size"],
                        p["md_threshold"], p["min_ref_count"], p["min_alt_count"], p["min_hap_count"], p["germline_snv_prior"], phase)
            hrecs, hlog = w.call_contig(b, chunks, pon, com, phase_sets)
        except Exception as e:                                  # noqa: BLE001
            herr = type(e).__name__ + ": " + str(e)[:120]
        if nb and oerr is None and herr is not None and ("error 3" in herr or "error 4" in herr):
            # the N landed on a substitution: cs still names the old base, which the reference would pile (it takes a
            # substitution's base from cs) and the library refuses (it takes every base from SEQ and checks the two agree:
            # DESIGN section 5) -- an input no aligner writes
            print("skip (N on a substitution: cs and SEQ disagree)  " + what, flush=True)
            continue
        if (oerr is None) != (herr is None):
            print("DIFFERENT (call: oracle raised {}, library raised {}): {} [zero qualities {} N bases {} chunks {}]".format(oerr, herr, what, zq, nb, chunks)); return 1
        if oerr is None:
            ok = hlog == olog and len(hrecs) == len(orecs) and all(np.array_equal(hrecs[k], orecs[k]) for k in (
                "tpos", "chunk", "phase_set", "gq", "ref", "alt", "gt0", "gt1", "status", "gt_state", "counts", "bqsum"))
            if not ok:
                print("DIFFERENT (call): " + what)
                if a.only is not None or a.verbose:
                    print("chunks", chunks)
                    print("log oracle ", olog); print("log library", hlog, " records", len(orecs), len(hrecs))
                    ko = set((int(r["tpos"]), int(r["chunk"]), int(r["ref"]), int(r["alt"])) for r in orecs)
                    kh = set((int(r["tpos"]), int(r["chunk"]), int(r["ref"]), int(r["alt"])) for r in hrecs)
                    print("library only:", sorted(kh - ko)[:40]); print("oracle only:", sorted(ko - kh)[:40])
                    print("reads", len(b.tstart), "last read starts", b.tstart[-5:], "ends", b.tend[-5:])
                    n = min(len(orecs), len(hrecs))
                    for k in ("tpos", "chunk", "phase_set", "gq", "ref", "alt", "gt0", "gt1", "status", "gt_state", "counts", "bqsum"):
                        d = np.nonzero(np.any(np.atleast_2d((hrecs[k][:n] != orecs[k][:n]).reshape(n, -1)), axis=1) if n else np.zeros(0))[0]
                        if len(d):
                            i = int(d[0])
                            print("field", k, "first difference at record", i, "of", len(d), ": oracle", orecs[i], "| library", hrecs[i]); break
                return 1
        # ---- the normcounts sweep
        refseq = bytearray(bytes(s.ref))
        if rs.rand() < 0.3:                                     # soft-masked stretches and runs of N in the FASTA (case matters: normcounts.py:320)
            for _ in range(int(rs.randint(1, 6))):
                a0 = int(rs.randint(0, len(refseq))); n0 = int(rs.choice([1, 3, 50, 3000]))
                if rs.rand() < 0.5:
                    refseq[a0:a0 + n0] = bytes(refseq[a0:a0 + n0]).lower()
                else:
                    refseq[a0:a0 + n0] = b"N" * len(refseq[a0:a0 + n0])
        refseq = bytes(refseq)
        nhs = bool(rs.rand() < 0.2)
        perm = lambda x: [x[i] for i in rs.permutation(3)]
        order_r = {r_: perm([c_ for c_ in "ATGC" if c_ != r_]) for r_ in "ATGC"} if rs.rand() < 0.5 else order
        oerr = herr = None
        try:
            o_ccs, o_ref, o_log = O.normcounts(b, chunks, p, refseq, p["germline_snv_prior"], pon, com, alt_order=order_r, non_human_sample=nhs, phase=phase_sets)
        except Exception as e:                                  # noqa: BLE001
            oerr = type(e).__name__
        try:
            ccs, rf, log = normcounts.norm_contig(w, b, chunks, refseq, pon, com, nhs, order_r, phase_sets=phase_sets)
        except Exception as e:                                  # noqa: BLE001
            herr = type(e).__name__
        if (oerr is None) != (herr is None):
            print("DIFFERENT (normcounts: oracle raised {}, library raised {}): {}".format(oerr, herr, what)); return 1
        if oerr is None and not (log == o_log and ccs == o_ccs and rf == o_ref):
            print("DIFFERENT (normcounts): " + what); return 1
        # ---- the BAM ingest: the batch written to a file, parsed on the device and by the host, compared byte for byte
        if rs.rand() < 0.15 and b.length < 160_000 * max(1.0, a.scale / 4):
            from himut_amd import bamio
            with tempfile.TemporaryDirectory() as d:
                path = os.path.join(d, "f.bam")
                bamio.write_bam(path, [b], sample="S")
                host = bamio.BamFile(path, threads=2).batches[b.name]
                st = bamio.BamStream(path, threads=3)
                res = st.ingest_contig(w.ctx, b.name, window_bytes=int(rs.choice([96, 300, 1024])) << 10)
                got = w.ctx.download_reads(res, b.name, st.tname2tsize[b.name])
                st.close()
            for k in ("tstart", "tend", "qstart", "qlen", "mapq", "flag", "qid", "qoff", "cs_off", "seq", "bq", "cs", "tp"):
                x, y = getattr(got, k), getattr(host, k)
                if x.shape != y.shape or not np.array_equal(x, y):
                    print("DIFFERENT (ingest, field {}): {}".format(k, what)); return 1
        # ---- the edge counts of himut phase, through the same context
        if rs.rand() < 0.5:
            hets = sorted(set((int(pp) + 1, chr(r), chr(al)) for pp, r, al, g in zip(s.snp_pos, s.snp_ref, s.snp_alt, s.snp_gt) if g in (1, 2)))
            if len(hets) >= 2:
                ebq, emq = int(rs.choice([0, 20, 93])), int(rs.choice([0, 20, 60]))
                o_lst, o_e2c = O.edges(b, hets, ebq, emq)
                hpos = np.array([h[0] for h in hets], np.int32)
                href = np.array([ord(h[1]) for h in hets], np.uint8)
                band = O.edge_band(b, hpos)
                w.ctx.push_reads(b)
                cnt = w.ctx.run_edges(hpos, href, ebq, emq, band).reshape(-1, 4)
                h_e2c = {}
                for e in np.flatnonzero(cnt.sum(1)):
                    i, d = int(e) // band, int(e) % band
                    h_e2c[(i, i + 1 + d)] = [float(x) for x in cnt[e]]
                if sorted(h_e2c) != o_lst or h_e2c != o_e2c:
                    print("DIFFERENT (edges): " + what); return 1
        print("ok  {:5.1f} s  records {}  callable {}  {}{}".format(time.time() - t0, -1 if orecs is None else len(orecs),
                                                                    -1 if oerr else o_log[13], what,
                                                                    "" if orecs is not None and not oerr else "  [raised: call {} normcounts {}]".format(
                                                                        "yes" if orecs is None else "no", oerr)), flush=True)
    print("fuzz ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())
