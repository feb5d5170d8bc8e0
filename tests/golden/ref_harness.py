"""Harness that runs the *reference* himut (read-only checkout under
/root/reference) on in-memory read batches, so golden vectors can be captured.

ONLY used by tests/golden/make_golden.py inside the build container; nothing on
the GPU box imports this file's reference-dependent parts (the reference does
not travel).  The reference's arithmetic on this path is pure Python + numpy;
its six third-party imports that are absent here (pysam, tabix, cyvcf2,
pyfastx, natsort, plotnine) are replaced by placeholders, of which only two are
ever exercised on the path:

* ``pysam.AlignmentFile`` -> FakeAlignmentFile: serves the reads of a
  ReadBatch with htslib's documented ``fetch`` overlap rule (0-based half-open,
  ``reference_start < end and reference_end > start``, file order).
* ``natsort.natsorted`` -> digit-aware sort stand-in.

Because those two boundaries are stand-ins, parity at the pysam/natsort
boundary is "unpinned" (SURVEY.md §8c); everything behind them is the real
reference code.
"""
import array
import os
import re
import sys
import types

REFERENCE_SRC = "/root/reference/src"

_FAKE_BAMS = {}


def register_bam(path, batches, sample="syn"):
    """batches: dict contig name -> ReadBatch (insertion order = @SQ order)."""
    _FAKE_BAMS[path] = (batches, sample)


class FakeRead:
    __slots__ = ("is_secondary", "reference_name", "reference_start", "reference_end", "query_name",
                 "query_alignment_start", "query_alignment_end", "query_sequence", "mapping_quality",
                 "query_qualities", "_tags")

    def get_tag(self, k):
        return self._tags[k]

    def has_tag(self, k):
        return k in self._tags


def _reads_of(batch):
    cache = getattr(batch, "_fake_reads", None)
    if cache is not None:
        return cache
    out = []
    for i in range(batch.n):
        r = FakeRead()
        r.is_secondary = bool(int(batch.flag[i]) & 0x100)
        r.reference_name = batch.name
        r.reference_start = int(batch.tstart[i])
        r.reference_end = int(batch.tend[i])
        r.query_name = batch.query_name(i)
        r.query_alignment_start = int(batch.qstart[i])
        r.query_alignment_end = int(batch.qlen[i])
        r.query_sequence = batch.query_sequence(i)
        r.mapping_quality = int(batch.mapq[i])
        r.query_qualities = array.array("B", bytes(batch.query_qualities(i)))
        tags = {"cs": batch.cs_tag(i)}
        if int(batch.tp[i]):
            tags["tp"] = chr(int(batch.tp[i]))
        r._tags = tags
        out.append(r)
    batch._fake_reads = out
    return out


class FakeHeader:
    def __init__(self, batches, sample):
        lines = ["@HD\tVN:1.6\tSO:coordinate"]
        for name, b in batches.items():
            lines.append("@SQ\tSN:{}\tLN:{}".format(name, b.length))
        lines.append("@RG\tID:1\tSM:{}".format(sample))
        self._text = "\n".join(lines) + "\n"

    def __str__(self):
        return self._text


class FakeAlignmentFile:
    def __init__(self, path, mode="rb", **kw):
        self._batches, sample = _FAKE_BAMS[path]
        self.header = FakeHeader(self._batches, sample)

    def fetch(self, contig=None, start=None, stop=None, **kw):
        b = self._batches[contig]
        if start is not None and stop is not None and start > stop:
            raise ValueError("invalid coordinates: start ({}) > stop ({})".format(start, stop))
        import numpy as np
        if start is None:
            start = 0
        if stop is None:
            stop = b.length
        idx = np.nonzero((b.tstart < stop) & (b.tend > start))[0]
        reads = _reads_of(b)
        for i in idx:
            yield reads[int(i)]

    def close(self):
        pass


def _natural_key(x):
    if isinstance(x, str):
        parts = re.split(r"(\d+)", x)
        return tuple(int(p) if p.isdigit() else p for p in parts)
    if isinstance(x, (tuple, list)):
        return tuple(_natural_key(y) for y in x)
    return ("", x)


def natsorted(seq, key=None, reverse=False, alg=None):
    if key is None:
        return sorted(seq, key=_natural_key, reverse=reverse)
    return sorted(seq, key=lambda v: _natural_key(key(v)), reverse=reverse)


_loaded = None


def load_reference():
    """Imports the reference's path modules; returns a namespace of them."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not os.path.isdir(REFERENCE_SRC):
        raise RuntimeError("reference checkout not present; goldens can only be regenerated in the build container")
    # numpy's SIMD argsort breaks ties differently from the scalar insertion
    # sort of the numpy the reference pins: this must be set
    # before numpy is imported to have an effect, so make_golden.py re-executes
    # itself with it; here we only record whether it is on.
    for name in ("pysam", "tabix", "cyvcf2", "pyfastx", "natsort", "plotnine"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["pysam"].AlignmentFile = FakeAlignmentFile
    sys.modules["natsort"].natsorted = natsorted
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import himut.caller
    import himut.bamlib
    import himut.cslib
    import himut.gtlib
    import himut.haplib
    import himut.util
    import himut.vcflib
    ns = types.SimpleNamespace(caller=himut.caller, bamlib=himut.bamlib, cslib=himut.cslib, gtlib=himut.gtlib,
                               haplib=himut.haplib, util=himut.util, vcflib=himut.vcflib)
    _loaded = ns
    return ns


CALL_DEFAULTS = dict(min_qv=30, min_mapq=60, min_sequence_identity=0.99, min_gq=20, min_bq=93, min_trim=0.01,
                     max_mismatch_count=0, mismatch_window_size=20, min_ref_count=3, min_alt_count=1,
                     min_hap_count=3, somatic_snv_prior=1 / (10 ** 6), germline_snv_prior=1 / (10 ** 3),
                     germline_indel_prior=1 / (10 ** 4))


def run_reference_worker(bam_path, chrom, chunks, qlen_lower, qlen_upper, md_threshold, common_snps=None,
                         panel_of_normals=None, phase=False, phase_sets=None, non_human_sample=False,
                         create_panel_of_normals=False, **overrides):
    """Calls the reference's per-chromosome worker (caller.py:208).  Returns
    (records, log) exactly as it assigns them."""
    ref = load_reference()
    p = dict(CALL_DEFAULTS)
    p.update(overrides)
    hbit, hpos, hetsnp = ({}, {}, {})
    if phase_sets is not None:
        hbit, hpos, hetsnp = phase_sets
    out_lst, out_log = {}, {}
    ref.caller.get_somatic_substitutions(
        chrom, bam_path, common_snps, panel_of_normals, [(chrom, s, e) for (s, e) in chunks], hbit, hpos, hetsnp,
        p["min_qv"], p["min_mapq"], qlen_lower, qlen_upper, p["min_sequence_identity"], p["min_gq"], p["min_bq"],
        p["min_trim"], p["max_mismatch_count"], p["mismatch_window_size"], md_threshold, p["min_ref_count"],
        p["min_alt_count"], p["min_hap_count"], p["somatic_snv_prior"], p["germline_snv_prior"],
        p["germline_indel_prior"], phase, non_human_sample, create_panel_of_normals, out_lst, out_log)
    return out_lst[chrom], out_log[chrom]


def run_reference_normcounts(bam_path, chrom, seq, chunks, qlen_lower, qlen_upper, md_threshold, common_snps=None,
                             panel_of_normals=None, non_human_sample=False, phase_sets=None, **overrides):
    """Calls the reference's normcounts worker (normcounts.py:206) without phasing.  Returns
    (ccs_tri2count, ref_tri2count, log, alt_order): alt_order[ref] = list(base_set.difference(ref)) as THIS
    interpreter orders it -- the worker's PoN/common precedence and its tie rule depend on it."""
    ref = load_reference()
    import himut.normcounts as N
    p = dict(CALL_DEFAULTS)
    p.update(overrides)
    ccs, rf, log = {}, {}, {}
    hbit, hpos, hetsnp = phase_sets if phase_sets is not None else ({}, {}, {})
    N.get_callable_tricounts(
        chrom, seq, bam_path, common_snps, panel_of_normals, [(chrom, s, e) for (s, e) in chunks], hbit, hpos, hetsnp,
        p["min_qv"], p["min_mapq"], p["min_trim"], qlen_lower, qlen_upper, p["min_sequence_identity"], p["min_gq"],
        p["min_bq"], p["mismatch_window_size"], p["max_mismatch_count"], p["min_ref_count"], p["min_alt_count"],
        p["min_hap_count"], md_threshold, p["somatic_snv_prior"], p["germline_snv_prior"], p["germline_indel_prior"],
        phase_sets is not None, non_human_sample, ccs, rf, log)
    order = {b: list(ref.util.base_set.difference(b)) for b in "ATGC"}
    return ccs[chrom], rf[chrom], log[chrom], order


class FakeFasta:
    """pyfastx.Fasta stand-in: fasta[chrom] behaves like the sequence string (indexing, slicing, str())."""

    def __init__(self, path, **kw):
        self._seqs = {}
        name, parts = None, []
        for line in open(path):
            if line.startswith(">"):
                if name is not None:
                    self._seqs[name] = "".join(parts)
                name, parts = line[1:].split()[0], []
            else:
                parts.append(line.strip())
        if name is not None:
            self._seqs[name] = "".join(parts)

    def __getitem__(self, chrom):
        return self._seqs[chrom]


def load_reference_norm_host():
    """mutlib / reflib / vcflib of the reference with pyfastx.Fasta served by FakeFasta."""
    load_reference()
    sys.modules["pyfastx"].Fasta = FakeFasta
    import himut.mutlib
    import himut.reflib
    import himut.vcflib
    return types.SimpleNamespace(mutlib=himut.mutlib, reflib=himut.reflib, vcflib=himut.vcflib)


def run_reference_edges(bam_path, chrom, hetsnp_lst, min_bq, min_mapq):
    """phaselib.get_edges of the reference (phaselib.py:16-67).  Returns (edge_lst, {edge: [4 counts]})."""
    load_reference()
    import scipy.stats
    if not hasattr(scipy.stats, "binom_test"):      # removed from scipy >= 1.12; phaselib imports it by name (get_edges never calls it)
        scipy.stats.binom_test = lambda x, n, p=0.5, alternative="two-sided": scipy.stats.binomtest(int(x), int(n), p, alternative).pvalue
    import himut.phaselib as PL
    hpos = [h[0] for h in hetsnp_lst]
    hidx = {h: i for i, h in enumerate(hetsnp_lst)}
    edge_lst, edge2counts = PL.get_edges(chrom, bam_path, min_bq, min_mapq, hpos, hetsnp_lst, hidx)
    return [list(e) for e in edge_lst], {"{},{}".format(*k): [float(x) for x in v] for k, v in edge2counts.items()}


def _phaselib():
    load_reference()
    import scipy.stats
    if not hasattr(scipy.stats, "binom_test"):      # removed from scipy >= 1.12; phaselib imports it by name
        scipy.stats.binom_test = lambda x, n, p=0.5, alternative="two-sided": scipy.stats.binomtest(int(x), int(n), p, alternative).pvalue
    import himut.phaselib as PL
    return PL


def run_reference_phase(bam_path, chrom, chrom_len, vcf_path, min_bq, min_mapq, min_p_value, min_phase_proportion,
                        out_path, tname2tsize):
    """The reference's `himut phase` for one contig without its process pool and input checks: get_hblock
    (phaselib.py:235-250), get_hblock_statistics and vcflib.dump_phased_hetsnps.  Returns
    (hblock_lst, statistics, phased VCF text)."""
    import numpy as np
    PL = _phaselib()
    ref = load_reference()
    out = {}
    PL.get_hblock(chrom, chrom_len, bam_path, vcf_path, min_bq, min_mapq, min_p_value, min_phase_proportion, out)
    hetsnp_lst, _, _ = ref.vcflib.load_hetsnps(vcf_path, chrom, chrom_len)
    stats = PL.get_hblock_statistics(out[chrom], hetsnp_lst)
    ref.vcflib.dump_phased_hetsnps(bam_path, vcf_path, chrom, None, tname2tsize, min_bq, min_mapq, min_p_value,
                                   min_phase_proportion, 1, [chrom], out, "1.0.4", out_path)
    blocks = [[[int(h), str(st)] for h, st in blk] for blk in out[chrom]]
    return blocks, [int(x) for x in stats], open(out_path).read()
