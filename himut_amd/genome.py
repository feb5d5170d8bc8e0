"""BASELINE.json configs[2]: a whole synthetic GRCh38 (24 contigs with the primary-assembly lengths, 3.09 Gb) at
30x through `himut call`'s scan -- the reference's ``Pool.starmap`` axis (caller.py:766-810) spread over the GPUs of a
node.  The contigs are packed onto the ranks by longest-processing-time (dist.lpt_assign); every rank keeps the
reads of its contigs resident in HBM (one context per contig: a whole share fits 288 GB many times over) and scans
them in turn; the record buffer of each finished contig travels to rank 0 while the next contig is scanned
(dist.RecordExchange, point-to-point over xGMI).  Strong scaling: the genome is fixed, N ranks share it.

Used by bench.py (--gpus N > 1), tools/bench_genome.py and tests/test_gpu_genome.py.  Test and bench
infrastructure: the reads are synthetic (synth.py); nothing here is on the product path of `himut call`."""
import time

import numpy as np

GRCH38 = {"chr1": 248956422, "chr2": 242193529, "chr3": 198295559, "chr4": 190214555, "chr5": 181538259,
          "chr6": 170805979, "chr7": 159345973, "chr8": 145138636, "chr9": 138394717, "chr10": 133797422,
          "chr11": 135086622, "chr12": 133275309, "chr13": 114364328, "chr14": 107043718, "chr15": 101991189,
          "chr16": 90338345, "chr17": 83257441, "chr18": 80373285, "chr19": 58617616, "chr20": 64444167,
          "chr21": 46709983, "chr22": 50818468, "chrX": 156040895, "chrY": 57227415}

CALL_PARAMS = dict(min_qv=30, min_mapq=60, min_sequence_identity=0.99, min_gq=20, min_bq=93, min_trim=0.01,
                   max_mismatch_count=0, mismatch_window_size=20, min_ref_count=3, min_alt_count=1, min_hap_count=3)
GERMLINE_SNV_PRIOR = 1 / (10 ** 3)


def side_sets(sample, seed):
    """PoN / common-SNP key arrays of the size BASELINE configs[1] describes
    (common = half the germline SNPs + 1e-4/bp decoys, PoN = 1e-4/bp random)."""
    from . import caller
    rs = np.random.RandomState(seed)
    L = sample.batch.length
    keep = rs.rand(sample.snp_pos.shape[0]) < 0.5
    common = [(int(p) + 1, chr(r), chr(a)) for p, r, a in zip(sample.snp_pos[keep], sample.snp_ref[keep], sample.snp_alt[keep])]
    n_decoy = int(1e-4 * L)
    pos = rs.randint(1, L + 1, size=2 * n_decoy)
    ra = rs.randint(0, 4, size=(2 * n_decoy, 2))
    decoys = [(int(p), "ACGT"[i], "ACGT"[j]) for p, (i, j) in zip(pos, ra) if i != j]
    h = len(decoys) // 2
    return caller.site_keys(decoys[:h]), caller.site_keys(common + decoys[h:])


def genome_sizes(scale=1.0):
    """Contig lengths of the synthetic genome (GRCh38 primary assembly divided by ``scale``), natural order."""
    from .util import natsorted
    sizes = {c: max(int(L / scale), 250_000) for c, L in GRCH38.items()}
    return sizes, natsorted(list(sizes))


def contig_inputs(name, length, depth, names):
    """Seeded synthetic inputs of one contig: read batch, reference chunks, side sets."""
    from . import synth, util as hutil
    k = names.index(name)
    s = synth.generate(synth.SynthConfig(seed=300 + k, contig_len=length, depth=depth, name=name))
    chunks = [(c[1], c[2]) for c in hutil.chunkloci((name, 0, length))]
    pon, com = side_sets(s, 500 + k)
    return s.batch, chunks, pon, com


class ResidentContig:
    """One contig of this rank's share: a context of its own with the reads in HBM."""

    def __init__(self, name, worker, span, n_reads, read_bases):
        self.name, self.worker, self.span, self.n_reads, self.read_bases = name, worker, span, n_reads, read_bases
        self.ctx = worker.ctx


def thresholds_for(batch):
    from . import bamlib
    return bamlib.get_thresholds({batch.name: batch}, [batch.name], {batch.name: batch.length})


def load_share(mine, sizes, names, depth, device, thresholds=None, broadcast=None):
    """Generates and uploads this rank's contigs (``mine``, in scan order).  ``thresholds``: (qlen_lower, qlen_upper,
    md) or None = take them from the first contig loaded and pass them through ``broadcast`` (rank 0's go to
    everyone, as the reference's one global get_thresholds does).  Returns (list of ResidentContig, thresholds,
    seconds spent generating / uploading)."""
    from . import caller
    out = []
    t_gen = t_h2d = 0.0
    pending = []
    first = None
    for c in mine:
        t0 = time.perf_counter()
        batch, chunks, pon, com = contig_inputs(c, sizes[c], depth, names)
        t_gen += time.perf_counter() - t0
        if first is None:
            first = batch
            if thresholds is None:
                thresholds = thresholds_for(batch)
                if broadcast is not None:
                    thresholds = tuple(broadcast(list(thresholds)))
        w = caller.Worker(device)
        ql, qu, md = thresholds
        w.configure(qlen_lower_limit=ql, qlen_upper_limit=qu, md_threshold=md, germline_snv_prior=GERMLINE_SNV_PRIOR,
                    phase=False, **CALL_PARAMS)
        t0 = time.perf_counter()
        w.ctx.set_chunks(chunks)
        w.ctx.set_site_set(0, pon)
        w.ctx.set_site_set(1, com)
        w.ctx.push_reads(batch)
        t_h2d += time.perf_counter() - t0
        out.append(ResidentContig(c, w, sum(e - s + 1 for s, e in chunks), batch.n, batch.total_read_bases()))
        del batch
    if first is None and thresholds is None and broadcast is not None:      # a rank without contigs still joins
        thresholds = tuple(broadcast([0, 0, 0]))
    return out, thresholds, dict(generate=t_gen, h2d=t_h2d)


def run_genome(rank, world, device, scale=1.0, depth=30.0, steps=3, backend="nccl", keep_records=False, overlap=None):
    """The strong-scaling run.  Collective over the (already initialised when world > 1) process group; with
    world == 1 and no group it runs without any exchange unless ``keep_records`` asks for the gathered buffers (a
    one-rank group must then exist).  Returns on rank 0 a dict (see bench.py ``genome_strong``), with
    ``"records_by_contig"`` / ``"logs_by_contig"`` of the last pass when ``keep_records``; None on the other ranks."""
    import os
    import torch
    import torch.distributed as dist
    from . import dist as hdist
    if overlap is None:        # HIMUT_GENOME_OVERLAP=0: one contig's run at a time (per-contig device times are then meaningful)
        overlap = os.environ.get("HIMUT_GENOME_OVERLAP", "1") != "0"
    sizes, names = genome_sizes(scale)
    plan = hdist.lpt_assign(sizes, world)
    mine = plan[rank]
    grouped = dist.is_available() and dist.is_initialized()
    bc = (lambda v: hdist.broadcast_ints(v)) if grouped else None
    share, thresholds, setup = load_share(mine, sizes, names, depth, device, broadcast=bc)
    rounds = max(len(p) for p in plan)

    def one_pass(ex):
        # every contig's run is queued before any is waited for (a context each, streams of their own): the GPU goes from
        # one contig's last kernels into the next one's first without the host in between
        if overlap:
            for rc in share:
                rc.ctx.run_begin()
        dev_ms = 0.0
        for k in range(rounds):
            if k < len(share):
                ctx = share[k].ctx
                if overlap:
                    ctx.run_end()
                else:
                    ctx.run()
                dev_ms += ctx.stats()["ms_total"]
                if ex is not None:
                    n = ctx.records_device()[1]
                    if backend == "nccl":
                        ex.submit(n, ctx.log(), ctx=ctx)
                    else:
                        ex.submit(n, ctx.log(), records=ctx.records())
            elif ex is not None:
                ex.submit(0, [0] * 15)
        return dev_ms

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # rehearsal pass (untimed): sizes every work buffer and tells the exchange how much each round carries
    one_pass(None)
    one_pass(None)
    ex = None
    if grouped:
        caps = hdist.RecordExchange.plan([rc.ctx.records_device()[1] for rc in share])
        ex = hdist.RecordExchange(rank, world, caps, depth=2)
        one_pass(ex)
        ex.drain()
        ex.meta = []
    replanned = 0
    while True:
        barrier()
        t0 = time.perf_counter()
        dev_ms = 0.0
        for _ in range(steps):
            dev_ms += one_pass(ex)
        try:
            gathered = ex.drain() if ex is not None else None
        except hdist.RecordExchangeOverflow as e:      # (every rank alike) a pass held more records than the rehearsal: once more
            if replanned >= 2:
                raise
            replanned += 1
            ex = hdist.RecordExchange(rank, world, e.caps, depth=2)
            continue
        barrier()
        elapsed = time.perf_counter() - t0
        break

    st = [rc.ctx.stats() for rc in share]
    logs = [rc.ctx.log() for rc in share]
    tot = dict(span=sum(rc.span for rc in share), bases=sum(rc.read_bases for rc in share),
               reads=sum(rc.n_reads for rc in share), cand=sum(l[1] for l in logs),
               recs=sum(s["n_records"] for s in st), dev_s=dev_ms / 1e3 / steps, elapsed=elapsed,
               gen=setup["generate"], h2d=setup["h2d"], reran=sum(s["reran"] for s in st))
    keys = list(tot)
    if grouped:
        red = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        v = torch.tensor([float(tot[k]) for k in keys], dtype=torch.float64, device=red)
        vmax = v.clone()
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        sums, maxs = dict(zip(keys, v.tolist())), dict(zip(keys, vmax.tolist()))
    else:
        sums = maxs = tot
    if rank != 0:
        for rc in share:
            rc.worker.close()
        return None
    per_step = maxs["elapsed"] / steps
    out = {
        "workload": "whole synthetic GRCh38 (24 contigs, {:.0f} bp) {:.0f}x CCS, common-SNP + PoN filtering; contigs LPT-packed "
                    "onto {} rank(s), reads resident in HBM, records sent to rank 0 per contig while the next one is scanned"
                    .format(sums["span"], depth, world),
        "scaling": "strong", "n_gpus": world, "steps": steps, "scale": scale,
        "Mbp_per_s": sums["span"] / 1e6 / per_step, "candidate_sites_per_s": sums["cand"] / per_step,
        "s_per_genome": per_step, "genome_bp": int(sums["span"]), "reads": int(sums["reads"]),
        "read_bases": int(sums["bases"]), "candidate_sites": int(sums["cand"]), "records": int(sums["recs"]),
        # (with the contigs' runs overlapped a contig's own device time includes its waiting behind the others: the sum
        # says nothing, and what of the exchange is not hidden cannot be told apart)
        "runs_overlapped": bool(overlap),
        "slowest_rank_device_s": None if overlap else maxs["dev_s"],
        "exchange_exposed_s": None if overlap else max(per_step - maxs["dev_s"], 0.0),
        "setup_s_max_rank": {"generate": maxs["gen"], "h2d_pageable": maxs["h2d"]},
        "reran": int(sums["reran"]), "thresholds": list(thresholds) if thresholds else None,
        "contigs_per_rank": plan,
    }
    if gathered is not None:
        counts, _ = gathered
        # every pass delivered every contig's records: counts of the last pass against the ranks' own totals
        got = sum(int(counts[r][-(rounds - k)][0]) for r in range(world) for k in range(rounds))
        assert got == int(sums["recs"]), (got, sums["recs"])
        if keep_records:
            recs = {}
            nslot = ex.rounds * ex.depth
            for k in range(rounds):
                slot = (ex.k - rounds + k) % nslot
                for r in range(world):
                    if k < len(plan[r]):
                        n = int(counts[r][-(rounds - k)][0])
                        raw = ex.recv[slot][r][:n * hdist.REC].cpu().numpy()
                        recs[plan[r][k]] = raw.view(hdist.RECORD_DTYPE).copy()
            out["records_by_contig"] = recs
            out["logs_by_contig"] = {plan[r][k]: [int(x) for x in counts[r][-(rounds - k)][1:]]
                           for r in range(world) for k in range(len(plan[r]))}
    elif keep_records:
        out["records_by_contig"] = {rc.name: rc.ctx.records() for rc in share}
        out["logs_by_contig"] = {rc.name: rc.ctx.log() for rc in share}
    for rc in share:
        rc.worker.close()
    return out
