"""oracle/convert_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy) of the reference's BED -> CpG-index join (`wgbstools convert -L`), for checking the HIP path
(k_convert behind wgbsseg_convert_regions):
    fast_join   convert.py:147-185 chr_thread on a chromosome whose regions do not overlap: two forward as-of joins against the
                chromosome's dictionary rows (init_genome.py:151-157: chr, locus, 1-based index)
                    startCpG = index of the first CpG with locus >= start                                   (:161-163)
                    endCpG   = index of the first CpG with locus >= end, +1 when that locus == end, and the chromosome's
                               cumulative CpG count + 1 when there is none                                  (:166-171)
                regions whose start lies beyond the last CpG, or that hold no CpG (endCpG - startCpG <= 0), get NA (:180-184)
    slow_join   convert.py:133-145 slow_conversion, used for a chromosome with overlapping regions: one GenomicRegion per row
                (genomic_region.py:126-161): end <= start, start < 1 or end > chromosome length are errors (NA); the CpGs with
                start <= locus <= end are rows first..last of the dictionary; endCpG = last + 1, or last when the last
                locus equals `end` exactly (the awk rule of :146-148); no CpG in range, or an empty range, is NA
Both return (startCpG, endCpG) int64 arrays with 0 for NA.  Parity: pinned by tests/test_convert_cpu.py against
tests/golden/convert_cases.json, captured from the reference's own Python by tests/golden/make_golden_convert.py.
Only tests/ may import this module.
"""
import numpy as np


def fast_join(L, lo, start, end):
    """L: int64 loci of one chromosome (ascending); lo: 0-based global index of its first CpG."""
    L = np.asarray(L, dtype=np.int64)
    start = np.asarray(start, dtype=np.int64)
    end = np.asarray(end, dtype=np.int64)
    s = np.searchsorted(L, start, 'left')
    j = np.searchsorted(L, end, 'left')
    hit = (j < L.size) & (L[np.minimum(j, L.size - 1)] == end)
    s_cpg = lo + s + 1
    e_cpg = lo + j + 1 + hit.astype(np.int64)
    ok = (s < L.size) & (e_cpg - s_cpg > 0)
    return np.where(ok, s_cpg, 0), np.where(ok, e_cpg, 0)


def slow_join(L, lo, start, end, chrom_bp_size):
    L = np.asarray(L, dtype=np.int64)
    start = np.asarray(start, dtype=np.int64)
    end = np.asarray(end, dtype=np.int64)
    ok = (end > start) & (start >= 1) & (end <= chrom_bp_size)
    i0 = np.searchsorted(L, start, 'left')
    i1 = np.searchsorted(L, end, 'right')
    ok &= i1 > i0
    first = lo + i0 + 1
    last = lo + i1
    r = (L[np.maximum(i1 - 1, 0)] < end).astype(np.int64)
    e_cpg = last + r
    ok &= e_cpg != first
    return np.where(ok, first, 0), np.where(ok, e_cpg, 0)


def has_overlaps(start, end):
    """convert.py:153-156 after dropping duplicate regions: sorted by start, some region begins before its predecessor ends."""
    start = np.asarray(start, dtype=np.int64)
    end = np.asarray(end, dtype=np.int64)
    o = np.argsort(start, kind='stable')
    return bool((start[o][1:] - end[o][:-1] < 0).any())
