"""Side-VCF loaders and the himut VCF / log writers (reference:
src/himut/vcflib.py).  These stay in Python by design: the
device returns integers, and dividing / formatting them with the same Python
float formatting the reference uses makes the text identical by construction.

Behaviours reproduced on purpose (they change the output):
* load_common_snp keeps records of every contig EXCEPT the requested one,
  keyed by position only (vcflib.py:434).
* load_pon indexes column 10 of every record, so PoN files need a sample column
  (vcflib.py:26).
* the header prints max_mismatch_count under --mismatch_window and vice versa,
  because the driver passes them swapped (caller.py:742-743 vs vcflib.py:167-168).
"""
from collections import defaultdict
from datetime import datetime

import numpy as np

from .util import natsorted

LOG_ROWS = ["num_ccs", "num_sbs", "num_het_sbs", "num_hetalt_sbs", "num_homalt_sbs", "num_somrev_sbs",
            "num_homref_sbs", "num_uncallable_sbs", "num_low_gq_sbs", "num_low_bq_sbs", "num_pon_filtered_sbs",
            "num_pop_filtered_sbs", "num_md_filtered_sbs", "num_ab_filtered_sbs", "num_som"]


class VcfRecord:
    """Fields of one VCF data line that the path looks at (vcflib.py:13-52)."""

    __slots__ = ("chrom", "pos", "ref", "alt_lst", "alt", "qual", "is_pass", "is_biallelic", "is_snp", "is_indel",
                 "sample_gt", "sample_phase_set")

    def __init__(self, line):
        f = line.strip().split()
        self.chrom = f[0]
        self.pos = int(f[1])
        self.ref = f[3]
        self.alt_lst = f[4].split(",")
        self.qual = float(f[5]) if f[5] != "." else f[5]       # printed back with str() by the phased-hetSNP writer
        self.is_pass = f[6] == "PASS"
        fmt = dict(zip(f[8].split(":"), f[9].split(":")))   # IndexError without a sample column, as the reference
        self.sample_gt = fmt.get("GT")
        self.sample_phase_set = fmt.get("PS")
        self.is_biallelic = len(self.alt_lst) == 1
        self.alt = self.alt_lst[0] if self.is_biallelic else None
        self.is_snp = self.is_biallelic and len(self.ref) == 1 and len(self.alt) == 1
        # vcflib.py:41-50: a doublet substitution is neither; anything else of unequal length is an indel
        self.is_indel = (self.is_biallelic and not self.is_snp and not (len(self.ref) == len(self.alt) == 2)
                         and len(self.ref) != len(self.alt))


def _data_lines(path):
    with open(path) as fh:
        for line in fh:
            if not line.startswith("#"):
                yield line


_SIDE_CACHE = {}


def _by_contig(vcf_file, kind, parse):
    """The records of a side VCF that pass ``parse`` (a line -> (chrom, key) or None), grouped by contig.  The reference
    reads the whole file again for every contig (its workers are separate processes); here one driver process asks for
    contig after contig, and the file is parsed once per (path, size, time of modification)."""
    import os
    st = os.stat(vcf_file)
    key = (os.path.abspath(vcf_file), kind, st.st_size, st.st_mtime_ns)
    hit = _SIDE_CACHE.get(key)
    if hit is None:
        hit = defaultdict(set)
        for line in _data_lines(vcf_file):
            r = parse(line)
            if r is not None:
                hit[r[0]].add(r[1])
        for k in [k for k in _SIDE_CACHE if k[:2] == key[:2]]:
            del _SIDE_CACHE[k]                      # an older version of the same file
        _SIDE_CACHE[key] = hit
    return hit


def _pon_record(line):
    v = VcfRecord(line)
    return (v.chrom, (v.pos, v.ref, v.alt)) if v.is_snp and v.is_pass else None


def load_pon(chrom, vcf_file):
    """(pos, ref, alt) of PASS bi-allelic SNVs on ``chrom`` (vcflib.py:396-409)."""
    return set(_by_contig(vcf_file, "pon", _pon_record).get(chrom, ()))


def _bgz_lines(path, chrom):
    """Data lines of a bgzip-compressed VCF on ``chrom``.  A BGZF file is a series of gzip members, so python's
    gzip reads it; the reference goes through a tabix index (vcflib.py:381,417,449), which only selects the same
    lines faster.  The region arguments of the reference's per-chunk queries (chunk +- qlen_upper_limit,
    caller.py:269-289) always contain the chunk, so one pass per contig gives the same membership answers."""
    import gzip
    with gzip.open(path, "rt") as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            if line.split("\t", 1)[0] == chrom:
                yield line


def load_bgz_pon(chrom, vcf_file):
    """.bgz panel of normals on ``chrom`` (vcflib.py:412-423)."""
    out = set()
    for line in _bgz_lines(vcf_file, chrom):
        v = VcfRecord(line)
        if v.is_snp and v.is_pass:
            out.add((v.pos, v.ref, v.alt))
    return out


def load_bgz_common_snp(chrom, vcf_file):
    """.bgz common SNPs on ``chrom`` -- the tabix variant queries the RIGHT contig, unlike load_common_snp
    (vcflib.py:443-459)."""
    out = set()
    for line in _bgz_lines(vcf_file, chrom):
        f = line.strip().split("\t")
        alts = f[4].split(",")
        if f[6] == "PASS" and len(alts) == 1 and len(f[3]) == 1 and len(alts[0]) == 1:
            out.add((int(f[1]), f[3], alts[0]))
    return out


def _common_record(line):
    f = line.strip().split()
    alts = f[4].split(",")
    if f[6] == "PASS" and len(alts) == 1 and len(f[3]) == 1 and len(alts[0]) == 1:
        return f[0], (int(f[1]), f[3], alts[0])
    return None


def load_common_snp(chrom, vcf_file):
    """(pos, ref, alt) of PASS bi-allelic SNVs -- from the OTHER contigs
    (vcflib.py:426-440: ``if chrom != arr[0] and ...``)."""
    out = set()
    for c, keys in _by_contig(vcf_file, "common", _common_record).items():
        if c != chrom:
            out |= keys
    return out


def load_phased_hetsnps(vcf_file, chrom_lst, tname2tsize):
    """Phase sets of 0|1 / 1|0 records (vcflib.py:617-663).  Returns
    (chrom2ps2hbit, chrom2ps2hpos, chrom2ps2hetsnp, chrom2chunkloci); the chunk
    list becomes one (chrom, first_pos, last_pos) per phase set."""
    hbit = {t: defaultdict(list) for t in tname2tsize}
    hpos = {t: defaultdict(list) for t in tname2tsize}
    hsnp = {t: defaultdict(list) for t in tname2tsize}
    # a plain .vcf is read whole, a .bgz contig by contig in the order asked for (the reference's tabix queries)
    for _, v in _hetsnp_lines(vcf_file, chrom_lst):
        if v.sample_gt in ("0|1", "1|0"):
            if v.alt is None:
                raise AttributeError("multi-allelic phased record (the reference fails here too)")
            hpos[v.chrom][v.sample_phase_set].append(v.pos)
            hsnp[v.chrom][v.sample_phase_set].append((v.pos, v.ref, v.alt))
            hbit[v.chrom][v.sample_phase_set].append(v.sample_gt.split("|")[0])
    wanted = set(chrom_lst)
    chunks = {}
    for t in list(tname2tsize):
        if t not in wanted:
            del hbit[t], hpos[t], hsnp[t]
            continue
        chunks[t] = [(t, p[0], p[-1]) for p in hpos[t].values()]
    return hbit, hpos, hsnp, chunks


# --------------------------------------------------------------------------------------
# header (vcflib.py:150-353)

_FILTERS = [
    ("PASS", "All filters passed"),
    ("LowBQ", "Base quality score is below the minimum base quality score of {min_bq}"),
    ("LowGQ", "Germline genotype quality score is below the minimum genotype quality score of {min_gq}"),
    ("IndelSite", "Somatic substitution at indel site is not considered"),
    ("HetSite", "Somatic substitution at heterzygous SNP site is not considered"),
    ("HetAltSite", "Somatic substitution at tri-allelic SNP site is not considered"),
    ("HomAltSite", "Somatic substitution at homozygous alternative SNP site is not considered"),
    ("ComSnp", "Substitution is potentially from genomic DNA contamination"),
    ("PanelOfNormal", "Substitution is found within the Panel of Normal VCF file"),
    ("LowDepth", "Read depth is below the minimum reference allele and/or alterantive allele depth threshold"),
    ("HighDepth", "Read depth is above the maximum depth threshold of {md_threshold:.1f}"),
    ("Trimmed", "Substitution is positioned near the end of reads"),
    ("MismatchConflict", "Substitution is found next to a mismatch within a given mismatch window"),
    ("Unphased", "CCS read is not haplotype phased"),
]
_FORMATS = [
    ("GT", "1", "String", "Genotype"),
    ("GQ", "1", "String", "Genotype quality score"),
    ("BQ", "1", "Float", "Base quality score"),
    ("DP", "1", "Integer", "Read depth"),
    ("AD", "R", "Integer", "Read depth for each allele"),
    ("VAF", "A", "Float", "Variant allele fractions"),
    ("PS", "1", "Integer", "Phase set"),
]


def get_himut_vcf_header(bam_file, vcf_file, phased_vcf_file, region, region_list, tname2tsize, common_snps,
                         panel_of_normals, min_qv, min_mapq, qlen_lower_limit, qlen_upper_limit,
                         min_sequence_identity, min_gq, min_bq, min_trim, mismatch_window, max_mismatch_count,
                         md_threshold, min_ref_count, min_alt_count, min_hap_count, threads, somatic_snv_prior,
                         germline_snv_prior, germline_indel_prior, phase, non_human_sample, reference_sample,
                         create_panel_of_normals, version, out_file, sample):
    """Same positional meaning as vcflib.get_himut_vcf_header plus the sample
    name (the reference reads it from the BAM @RG line, bamlib.py:89-106)."""
    lines = ["##fileformat=VCFv4.2", "##fileDate={}".format(datetime.now().strftime("%d%m%Y")), "##source=himut",
             "##source_version={}".format(version), "##content=himut somatic single base substitutions"]
    for fid, desc in _FILTERS:
        lines.append('##FILTER=<ID={},Description="{}">'.format(
            fid, desc.format(min_bq=min_bq, min_gq=min_gq, md_threshold=md_threshold)))
    for fid, num, typ, desc in _FORMATS:
        lines.append('##FORMAT=<ID={},Number={},Type={},Description="{}">'.format(fid, num, typ, desc))
    for tname in natsorted(list(tname2tsize.keys())):
        lines.append("##contig=<ID={},length={}>".format(tname, tname2tsize[tname]))

    if region_list is not None:
        region_param = "--region_list {}".format(region_list)
    elif region is not None:
        region_param = "--region {}".format(region)
    else:
        region_param = ""
    opts = [("--min_qv", min_qv), ("--min_mapq", min_mapq), ("--qlen_lower_limit", qlen_lower_limit),
            ("--qlen_upper_limit", qlen_upper_limit), ("--min_sequence_identity", min_sequence_identity),
            ("--min_gq", min_gq), ("--min_bq", min_bq), ("--min_trim", min_trim),
            ("--mismatch_window", mismatch_window), ("--max_mismatch_count", max_mismatch_count),
            ("--min_ref_count", min_ref_count), ("--min_alt_count", min_alt_count)]
    if phase:
        opts.append(("--min_hap_count", min_hap_count))
    opts += [("--somatic_snv_prior", somatic_snv_prior), ("--germline_snv_prior", germline_snv_prior),
             ("--germline_indel_prior", germline_indel_prior), ("--threads", threads), ("-o", out_file)]
    param = region_param + "".join(" {} {}".format(k, v) for k, v in opts)

    head = "##himut_command=himut call -i {}".format(bam_file)
    cmd = None
    if phase and non_human_sample:
        cmd = "{} --vcf {} --phased_vcf {} {} --phase --non_human_sample{}".format(
            head, vcf_file, phased_vcf_file, param, " --reference_sample" if reference_sample else "")
    elif phase and not non_human_sample and not reference_sample:
        cmd = "{} --phased_vcf {} {} --common_snps {} --panel_of_normals {} --phase".format(
            head, phased_vcf_file, param, common_snps, panel_of_normals)
    elif not phase and non_human_sample and not create_panel_of_normals:
        cmd = "{} --vcf {} {} --non_human_sample{}".format(
            head, vcf_file, param, " --reference_sample" if reference_sample else "")
    elif not phase and not non_human_sample and not reference_sample and create_panel_of_normals:
        cmd = "{} --vcf {} {} --create_panel_of_normals".format(head, vcf_file, param)
    elif not phase and not non_human_sample and not reference_sample:
        cmd = "{} --vcf {} {} --common_snps {} --panel_of_normals {}".format(
            head, vcf_file, param, common_snps, panel_of_normals)
    if cmd is None:
        raise UnboundLocalError("flag combination has no command line in the reference header builder")
    lines.append(cmd)
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t{}".format(sample))
    return "\n".join(lines)


# --------------------------------------------------------------------------------------
# writers (vcflib.py:820-1060)

def _body_line(rec, phased, single_molecule_file):
    chrom, pos, ref, alt, status, gq, bq, depth, ref_count, alt_count, vaf, ps = rec
    if status == "HetAltSite":
        # bq / alt_count / vaf arrive pre-formatted (caller.py:174-192).  In the
        # phased main file the FORMAT column of this line lacks ":PS" although
        # the sample column has it (vcflib.py:954).
        fmt = "GT:GQ:BQ:DP:AD:VAF:PS" if (phased and single_molecule_file) else "GT:GQ:BQ:DP:AD:VAF"
        sample = "./.:{}:{}:{:0.0f}:{:0.0f},{}:{}".format(gq, bq, depth, ref_count, alt_count, vaf)
    else:
        fmt = "GT:GQ:BQ:DP:AD:VAF:PS" if phased else "GT:GQ:BQ:DP:AD:VAF"
        sample = "./.:{}:{:0.1f}:{:0.0f}:{:0.0f},{:0.0f}:{:.2f}".format(gq, bq, depth, ref_count, alt_count, vaf)
    if phased:
        sample += ":{}".format(ps)
    return "{}\t{}\t.\t{}\t{}\t.\t{}\t.\t{}\t{}\n".format(chrom, pos, ref, alt, status, fmt, sample)


def _dump(vcf_file, vcf_header, chrom_lst, chrom2tsbs_lst, phased):
    if not vcf_file.endswith(".vcf"):
        raise ValueError("VCF file must have .vcf suffix")
    with open(vcf_file, "w") as main, open(vcf_file.replace(".vcf", ".single_molecule_mutations.vcf"), "w") as sm:
        main.write("{}\n".format(vcf_header))
        sm.write("{}\n".format(vcf_header))
        for chrom in chrom_lst:
            for rec in chrom2tsbs_lst[chrom]:
                main.write(_body_line(rec, phased, False))
                # single-molecule file: one supporting read (vcflib.py:868,896)
                single = int(rec[8]) == 1 if rec[4] == "HetAltSite" else int(rec[9]) == 1
                if single:
                    sm.write(_body_line(rec, phased, True))


def dump_records(vcf_file, vcf_header, chrom_lst, chrom2recs, phased):
    """The same two files straight from the integer records (numpy arrays of RECORD_DTYPE), formatted by the host
    library: what dump_sbs / dump_phased_sbs write for caller.records_to_tuples(chrom, recs)."""
    from . import bamio
    if not vcf_file.endswith(".vcf"):
        raise ValueError("VCF file must have .vcf suffix")
    with open(vcf_file, "wb") as main, open(vcf_file.replace(".vcf", ".single_molecule_mutations.vcf"), "wb") as sm:
        main.write("{}\n".format(vcf_header).encode())
        sm.write("{}\n".format(vcf_header).encode())
        for chrom in chrom_lst:
            main.write(bamio.format_records(chrom2recs[chrom], chrom, phased, False))
            sm.write(bamio.format_records(chrom2recs[chrom], chrom, phased, True))


def dump_sbs(vcf_file, vcf_header, chrom_lst, chrom2tsbs_lst):
    _dump(vcf_file, vcf_header, chrom_lst, chrom2tsbs_lst, False)


def dump_phased_sbs(vcf_file, vcf_header, chrom_lst, chrom2tsbs_lst):
    _dump(vcf_file, vcf_header, chrom_lst, chrom2tsbs_lst, True)


def dump_call_log(chrom_lst, chrom2tsbs_log, path="himut.log"):
    """Counter table (vcflib.py:1024-1060), written to ./himut.log."""
    table = np.zeros((len(LOG_ROWS), len(chrom_lst)))
    for i, chrom in enumerate(chrom_lst):
        for j, count in enumerate(chrom2tsbs_log[chrom]):
            table[j][i] = count
    with open(path, "w") as o:
        o.write("{:30}{}\n".format("", "\t".join(chrom_lst + ["total"])))
        for k, name in enumerate(LOG_ROWS):
            cells = [str(int(x)) for x in table[k].tolist()] + [str(int(np.sum(table[k])))]
            o.write("{:30}{}\n".format(name, "\t".join(cells)))


# --------------------------------------------------------------------------
# `himut phase`: hetSNP input and the phased-hetSNP VCF (vcflib.py:84-147,462-550,704-817)

def _is_het_snp(v):
    return v.is_snp and v.is_pass and v.is_biallelic and v.sample_gt in ("0/1", "1/0")


def _hetsnp_lines(vcf_file, chrom_lst):
    """(line, record) of every data line the phaser looks at: the whole plain .vcf in file order, or the requested
    contigs of a .bgz one after the other (what the reference's tabix queries return)."""
    if vcf_file.endswith(".vcf"):
        for line in _data_lines(vcf_file):
            yield line, VcfRecord(line)
    elif vcf_file.endswith(".bgz"):
        for chrom in chrom_lst:
            for line in _bgz_lines(vcf_file, chrom):
                yield line, VcfRecord(line)


def load_hetsnps(vcf_file, chrom, chrom_len):
    """PASS bi-allelic 0/1 SNPs of ``chrom`` in file order: (hetsnp_lst, hidx2hetsnp, hetsnp2hidx)
    (vcflib.py:462-499)."""
    hetsnp_lst = [(v.pos, v.ref, v.alt) for _, v in _hetsnp_lines(vcf_file, [chrom]) if v.chrom == chrom and _is_het_snp(v)]
    hidx2hetsnp = dict(enumerate(hetsnp_lst))
    hetsnp2hidx = {h: i for i, h in hidx2hetsnp.items()}
    return hetsnp_lst, hidx2hetsnp, hetsnp2hidx


def load_hblock_hsh(vcf_file, chrom_lst, tname2tsize, chrom2hblock_lst):
    """hetSNP -> haplotype state and hetSNP -> phase set (= position of the block's first hetSNP) for every
    hetSNP a block holds (vcflib.py:502-550).  Keys carry no contig, as in the reference: the same (pos, ref, alt)
    on two contigs share an entry."""
    per_chrom = defaultdict(list)
    for _, v in _hetsnp_lines(vcf_file, chrom_lst):
        if _is_het_snp(v):
            per_chrom[v.chrom].append((v.pos, v.ref, v.alt))
    hetsnp2hstate, hetsnp2phase_set = {}, {}
    for chrom, hblock_lst in chrom2hblock_lst.items():
        snps = per_chrom[chrom]
        for hblock in hblock_lst:
            phase_set = snps[hblock[0][0]][0]
            for hidx, hstate in hblock:
                hetsnp2hstate[snps[hidx]] = hstate
                hetsnp2phase_set[snps[hidx]] = phase_set
    return hetsnp2hstate, hetsnp2phase_set


def get_phased_vcf_header(bam_file, vcf_file, region, region_list, tname2tsize, min_bq, min_mapq, min_p_value,
                          min_phase_proportion, threads, version, out_file, sample):
    """Header of the phased-hetSNP VCF (vcflib.py:84-147); ``sample`` is the BAM's SM tag."""
    h = ["##fileformat=VCFv4.2",
         '##FILTER=<ID=PASS,Description="All filters passed">',
         "##fileDate={}".format(datetime.now().strftime("%d%m%Y")),
         "##source=himut",
         "##source_version={}".format(version),
         "##content=himut somatic single base substitutions",
         '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
         '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Conditional genotype quality">',
         '##FORMAT=<ID=BQ,Number=1,Type=Float,Description="Average base quality">',
         '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read depth">',
         '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="Read depth for each allele">',
         '##FORMAT=<ID=VAF,Number=A,Type=Float,Description="Variant allele fractions">',
         '##FORMAT=<ID=PL,Number=G,Type=Integer,Description="Phred-scaled genotype likelihoods rounded to the closest integer">',
         '##FORMAT=<ID=PS,Number=1,Type=Integer,Description="Phase set">']
    h += ["##contig=<ID={},length={}>".format(t, tname2tsize[t]) for t in natsorted(list(tname2tsize))]
    if region_list is not None:
        region_param = "--region_list {}".format(region_list)
    elif region is not None:
        region_param = "--region {}".format(region)
    else:
        region_param = ""
    h.append("##himut_command=himut phase -i {} --vcf {} {} --min_bq {} --min_mapq {} --min_p_value {} "
             "--min_phase_proportion {} --threads {} -o {}".format(bam_file, vcf_file, region_param, min_bq, min_mapq,
                                                                   min_p_value, min_phase_proportion, threads, out_file))
    h.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t{}".format(sample))
    return "\n".join(h)


def dump_phased_hetsnps(bam_file, vcf_file, region, region_list, tname2tsize, min_bq, min_mapq, min_p_value,
                        min_phase_proportion, threads, chrom_lst, chrom2hblock_lst, version, out_file, sample):
    """The input's hetSNPs again, with GT 0|1 / 1|0 and PS for those inside a block and PS "." for the others
    (vcflib.py:704-817).  The plain-.vcf branch prints the hetSNPs of every contig of the file, the .bgz branch
    those of the requested contigs."""
    hetsnp2hstate, hetsnp2phase_set = load_hblock_hsh(vcf_file, chrom_lst, tname2tsize, chrom2hblock_lst)
    with open(out_file, "w") as o:
        o.write("{}\n".format(get_phased_vcf_header(bam_file, vcf_file, region, region_list, tname2tsize, min_bq, min_mapq,
                                                    min_p_value, min_phase_proportion, threads, version, out_file, sample)))
        for line, v in _hetsnp_lines(vcf_file, chrom_lst):
            if not _is_het_snp(v):
                continue
            fmt, sample_fmt = line.strip().split()[-2:]
            hetsnp = (v.pos, v.ref, v.alt)
            if hetsnp in hetsnp2hstate:
                cols = sample_fmt.split(":")
                cols[0] = "0|1" if hetsnp2hstate[hetsnp] == "0" else "1|0"
                tail = "{}:{}".format(":".join(cols), hetsnp2phase_set[hetsnp])
            else:
                tail = "{}:.".format(sample_fmt)
            o.write("{}\t{}\t.\t{}\t{}\t{}\tPASS\t.\t{}:PS\t{}\n".format(v.chrom, v.pos, v.ref, v.alt, v.qual, fmt, tail))


# --------------------------------------------------------------------------
# --non_human_sample: germline priors from the sample's own VCF (vcflib.py:553-613)

def load_germline_counts(vcf_file, chrom_lst):
    """(het SNPs, hom-alt SNPs, het indels, hom-alt indels) among the PASS bi-allelic records of the target contigs;
    only the genotype spellings 0/1 and 1/1 count (vcflib.py:553-593).  A .vcf.bgz is read as the gzip members it
    is made of (the reference goes through cyvcf2)."""
    import gzip
    if vcf_file.endswith(".vcf"):
        fh = open(vcf_file)
    elif vcf_file.endswith(".vcf.bgz"):
        fh = gzip.open(vcf_file, "rt")
    else:
        return 0, 0, 0, 0
    n = defaultdict(int)
    with fh:
        for line in fh:
            if line.startswith("#"):
                continue
            v = VcfRecord(line)
            if v.is_pass and v.is_biallelic and v.sample_gt in ("0/1", "1/1") and (v.is_snp or v.is_indel):
                n[(v.chrom, v.is_snp, v.sample_gt)] += 1
    tot = lambda snp, gt: sum(n[(c, snp, gt)] for c in chrom_lst)
    return tot(True, "0/1"), tot(True, "1/1"), tot(False, "0/1"), tot(False, "1/1")


def get_germline_priors(chrom_lst, ref_file, vcf_file, reference_sample):
    """Germline SNV and indel priors = variants per base of the target contigs, truncated to their first
    significant digit (vcflib.py:596-613): a reference sample counts its het sites twice, any other sample its
    het and hom-alt sites once."""
    from .normcounts import read_fasta
    from .util import get_truncated_float
    refseq = read_fasta(ref_file)
    target_sum = sum(len(refseq[chrom]) for chrom in chrom_lst)
    hetsnp, homsnp, hetindel, homindel = load_germline_counts(vcf_file, chrom_lst)
    if reference_sample:
        snv, indel = (2 * hetsnp) / target_sum, (2 * hetindel) / target_sum
    else:
        snv, indel = (hetsnp + homsnp) / target_sum, (hetindel + homindel) / target_sum
    return get_truncated_float(snv), get_truncated_float(indel)
