"""Host half of the germline genotyper (reference: src/himut/gtlib.py).

The per-read log10 terms depend only on the base quality, so the device sums
table entries.  The tables are built HERE, in Python, with the same
expressions the reference evaluates per read (gtlib.py:47-69) so that every
double comes out of the same libm calls; the kernels only add them in fetch
order (gtlib.py:84-96)."""
import math

import numpy as np

GT_LST = ["AA", "TA", "CA", "GA", "TT", "CT", "GT", "CC", "GC", "GG"]  # gtlib.py:9
GT_STATES = ("homref", "het", "hetalt", "homalt")


def gt_priors(germline_snv_prior):
    """gtlib.init (gtlib.py:12-20)."""
    p = germline_snv_prior
    return {
        "het": p,
        "hetalt": p * p * 2,
        "homref": 1 - ((1.5 * p) + (p * p)),
        "homalt": p / 2,
    }


def build_tables(germline_snv_prior, n_bq=256):
    """Returns (log_hom, log_het, log_err, log_prior[homref, het, hetalt, homalt]).

    Entry 0 is NaN: the reference raises ValueError (math.log10(0)) when a
    candidate column holds a BQ 0 base; the device reports that as
    HIMUT_ERR_BQ0 instead of using the entry."""
    hom = np.full(n_bq, np.nan)
    het = np.full(n_bq, np.nan)
    err = np.full(n_bq, np.nan)
    for bq in range(1, n_bq):
        epsilon = 10 ** (-bq / 10)                       # get_epsilon, gtlib.py:47-49
        hom[bq] = math.log10(1 - epsilon)                # gtlib.py:52-53,64-65
        het[bq] = math.log10(0.5 - epsilon / 2.0)        # gtlib.py:56-57,68-69
        err[bq] = math.log10(10 ** (-(bq / 3) / 10))     # gtlib.py:60-61 with bq/3 (gtlib.py:93)
    pri = gt_priors(germline_snv_prior)
    log_prior = np.array([math.log10(pri[s]) for s in GT_STATES])  # gtlib.py:41-44
    return hom, het, err, log_prior
