"""Deterministic synthetic inputs for tests and benchmarks (ours; nothing like it exists in the reference).

The reference's own test inputs are private cluster files (tests/integration/test_segment_integration.py:36-37),
so every parity case in this repo is regenerated from a seed instead of stored:

* ``synth_betas``  -- the bytes of a ``.beta`` file (docs/beta_format.md:3-8: NR_SITES x 2 uint8, #meth, #cov).
  Counter-based (splitmix64 of (seed, stream, site)), integer-only, so the numpy implementation here and the
  HIP implementation in ``csrc/synth.hip`` produce identical bytes for any sub-range of any sample.
* ``synth_loci``   -- bp position of every CpG of an hg19-shaped genome (25 % dense gaps
  ``2+Geom(1/10)``, 75 % sparse gaps ``2+Geom(1/140)``), host-only.
* ``write_genome`` -- a ``references/<name>/`` directory in the reference's on-disk format
  (init_genome.py:151-187: CpG.bed.gz, CpG.chrome.size, chrome.size) plus our binary loci cache.

Model of the betas: methylation level is piecewise constant over blocks whose starts are drawn per site
(P = 1/40, forced every 4096 sites), U-shaped level distribution shared by all samples, per-sample jitter and
a 1/16 chance per (sample, block) of a differential level; coverage ~ 6 + Binomial(48, 1/2) with 5 % forced
zero-coverage sites; #meth ~ Binomial(cov, level/256) from byte-wise Bernoulli trials.
"""
import gzip
import os
import os.path as op
import zlib

import numpy as np

U64 = np.uint64
_GOLD = U64(0x9E3779B97F4A7C15)
_M1 = U64(0xBF58476D1CE4E5B9)
_M2 = U64(0x94D049BB133111EB)
_STREAM_MUL = U64(0xD1B54A32D192ED03)

FORCE_BLOCK = 4096          # a block start is forced at every multiple of this (bounds the look-back)
BLOCK_ODDS = 40             # P(block start) = 1/40 per site
ZERO_COV_THRESH = 3277      # of 65536 -> 5 %
S_BLOCK, S_LEVEL, S_LOCI_KIND, S_LOCI_U = 1, 2, 5, 6
S_ISLAND, S_ISLAND_GAP = 7, 8
S_SAMPLE0 = 16              # streams 16+4*s+{0: block jitter, 1: coverage, 2: bernoulli bytes}

# hg19 chromosome lengths chr1-22,X,Y,M -- CpG counts are made proportional to these
HG19_LEN = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022,
            141213431, 135534747, 135006516, 133851895, 115169878, 107349540, 102531392, 90354753,
            81195210, 78077248, 59128983, 63025520, 48129895, 51304566, 155270560, 59373566, 16571]
HG19_NAMES = ['chr%d' % i for i in range(1, 23)] + ['chrX', 'chrY', 'chrM']
HG19_NR_SITES = 28217448    # init_genome.py:215-218


def splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over='ignore'):
        z = (x + _GOLD).astype(U64)
        z = ((z ^ (z >> U64(30))) * _M1).astype(U64)
        z = ((z ^ (z >> U64(27))) * _M2).astype(U64)
        return z ^ (z >> U64(31))


def stream_key(seed, stream):
    with np.errstate(over='ignore'):
        return splitmix64(np.array([U64(seed) ^ (U64(stream) * _STREAM_MUL)], dtype=U64))[0]


def hash_at(seed, stream, idx):
    """h(seed, stream, idx) = splitmix64(key(seed, stream) + idx)."""
    with np.errstate(over='ignore'):
        return splitmix64((stream_key(seed, stream) + np.asarray(idx, dtype=U64)).astype(U64))


def level_map(u):
    """byte -> methylation level byte (level/256): 25 % low (0..15), 56 % high (200..254), 19 % middle."""
    u = np.asarray(u, dtype=np.int64)
    low = u >> 2
    high = 200 + ((u - 64) * 55) // 144
    mid = 16 + ((u - 208) * 184) // 48
    return np.where(u < 64, low, np.where(u < 208, high, mid)).astype(np.int64)


def block_starts(seed, a, b):
    """For sites a <= i < b (0-based global index): index of the last block start <= i."""
    a0 = (a // FORCE_BLOCK) * FORCE_BLOCK
    idx = np.arange(a0, b, dtype=np.int64)
    h = hash_at(seed, S_BLOCK, idx)
    r = ((h >> U64(32)) * U64(BLOCK_ODDS)) >> U64(32)
    flag = (r == 0) | (idx % FORCE_BLOCK == 0)
    start = np.maximum.accumulate(np.where(flag, idx, a0))
    return start[a - a0:]


def synth_betas(seed, sample, a, b):
    """uint8 array [b-a, 2] = (#meth, #cov) of sites [a, b) of synthetic sample number `sample`."""
    n = b - a
    if n <= 0:
        return np.zeros((0, 2), dtype=np.uint8)
    idx = np.arange(a, b, dtype=np.int64)
    bs = block_starts(seed, a, b)
    base = level_map(hash_at(seed, S_LEVEL, bs) & U64(0xFF))
    st = S_SAMPLE0 + 4 * sample
    hs = hash_at(seed, st + 0, bs)
    flip = (hs & U64(15)) == 0
    alt = level_map((hs >> U64(8)) & U64(0xFF))
    jit = ((hs >> U64(16)) & U64(31)).astype(np.int64) - 16
    level = np.where(flip, alt, np.clip(base + jit, 0, 255))

    hc = hash_at(seed, st + 1, idx)
    zero = (hc & U64(0xFFFF)) < U64(ZERO_COV_THRESH)
    cov = np.bitwise_count((hc >> U64(16)) & U64((1 << 48) - 1)).astype(np.int64) + 6
    cov = np.where(zero, 0, cov)

    meth = np.zeros(n, dtype=np.int64)
    for g in range(7):                       # cov <= 54 -> at most 7 groups of 8 byte-trials
        hb = hash_at(seed, st + 2, idx * 8 + g)
        for byte in range(8):
            t = g * 8 + byte
            trial = ((hb >> U64(8 * byte)) & U64(0xFF)).astype(np.int64)
            meth += ((trial < level) & (t < cov)).astype(np.int64)
    out = np.empty((n, 2), dtype=np.uint8)
    out[:, 0] = meth
    out[:, 1] = cov
    return out


def genome_shape(total_sites=HG19_NR_SITES, n_chroms=25):
    """CpG counts per chromosome, proportional to the first `n_chroms` hg19 lengths, summing to total_sites."""
    lens = np.array(HG19_LEN[:n_chroms], dtype=np.float64)
    sizes = np.floor(lens / lens.sum() * total_sites).astype(np.int64)
    sizes = np.maximum(sizes, 1)
    sizes[0] += total_sites - sizes.sum()
    assert sizes.sum() == total_sites and (sizes > 0).all()
    return HG19_NAMES[:n_chroms], sizes


def island_mask(seed, total):
    """CpG-island-like stretches: one per 1024 sites, 20..620 sites long (exponential tail, mean ~80) — about the
    count and CpG content of the hg19 island track (28 k islands, 2.1 M CpGs).  Not part of the default genome."""
    idx = np.arange(total, dtype=np.int64)
    blk = idx >> 10
    h = hash_at(seed, S_ISLAND, blk)
    off = (h & U64(1023)).astype(np.int64) % 900
    u = ((h >> U64(11)).astype(np.float64) + 1.0) / 9007199254740992.0
    ln = 20 + np.minimum(600, np.floor(-60.0 * np.log(u))).astype(np.int64)
    x = idx & 1023
    return (x >= off) & (x < off + ln)


def synth_loci(seed, chrom_sizes, islands=False):
    """uint32 bp position of every CpG; positions restart at each chromosome (strictly ascending inside one).
    islands=True adds CpG islands (gaps 2 + Geom(1/8) bp): forward windows of several hundred sites."""
    total = int(np.sum(chrom_sizes))
    idx = np.arange(total, dtype=np.int64)
    dense = (hash_at(seed, S_LOCI_KIND, idx) & U64(3)) == 0
    u = ((hash_at(seed, S_LOCI_U, idx) >> U64(11)).astype(np.float64) + 1.0) / 9007199254740992.0   # (0, 1]
    p = np.where(dense, 1.0 / 10.0, 1.0 / 140.0)
    gap = 2 + np.floor(np.log(u) / np.log1p(-p)).astype(np.int64)
    gap = np.minimum(gap, 50000)
    if islands:
        ui = ((hash_at(seed, S_ISLAND_GAP, idx) >> U64(11)).astype(np.float64) + 1.0) / 9007199254740992.0
        gi = 2 + np.floor(np.log(ui) / np.log1p(-1.0 / 8.0)).astype(np.int64)
        gap = np.where(island_mask(seed, total), gi, gap)
    loci = np.empty(total, dtype=np.int64)
    pos = 0
    for sz in chrom_sizes:
        sz = int(sz)
        g = gap[pos:pos + sz].copy()
        g[0] = 10000 + (g[0] % 1000)
        loci[pos:pos + sz] = np.cumsum(g)
        pos += sz
    assert loci.max() < 2 ** 32
    return loci.astype(np.uint32)


def checksum(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c & 0xFFFFFFFF


def write_beta(path, data):
    np.ascontiguousarray(data, dtype=np.uint8).tofile(path)


def write_genome(refdir, names, sizes, loci, gz=True):
    """Write references/<name>/ the way init_genome.py:151-187 lays it out (no tabix index: we never need one),
    plus `loci.u32` (our binary cache of column 2)."""
    os.makedirs(refdir, exist_ok=True)
    sizes = [int(s) for s in sizes]
    with open(op.join(refdir, 'CpG.chrome.size'), 'w') as f:
        for c, s in zip(names, sizes):
            f.write('%s\t%d\n' % (c, s))
    with open(op.join(refdir, 'chrome.size'), 'w') as f:
        pos = 0
        for c, s in zip(names, sizes):
            f.write('%s\t%d\n' % (c, int(loci[pos + s - 1]) + 10000))
            pos += s
    lines = []
    pos = 0
    for c, s in zip(names, sizes):
        sub = loci[pos:pos + s]
        ind = np.arange(pos + 1, pos + s + 1)
        lines.append('\n'.join('%s\t%d\t%d' % (c, l, i) for l, i in zip(sub.tolist(), ind.tolist())))
        pos += s
    text = ('\n'.join(lines) + '\n').encode()
    dict_path = op.join(refdir, 'CpG.bed.gz')
    if gz:
        with gzip.open(dict_path, 'wb', compresslevel=1) as f:
            f.write(text)
    else:
        with open(dict_path, 'wb') as f:
            f.write(text)
    rev = op.join(refdir, 'rev.CpG.bed.gz')
    if op.lexists(rev):
        os.remove(rev)
    os.symlink('CpG.bed.gz', rev)          # init_genome.py:163-168: same file, second index
    np.asarray(loci, dtype=np.uint32).tofile(op.join(refdir, 'loci.u32'))
    return refdir


def synth_pat_lines(seed, n_sites, n_reads, names=None, sizes=None):
    """Lines of a synthetic pat file over CpGs 1..n_sites (pat2beta tests): `chr \t first CpG \t pattern \t count`, reads of 1-12
    CpGs over {C, T, H, .}, counts 1-40, sorted by start as real pat files are; a few reads hang over the ends of the range."""
    idx = np.arange(n_reads, dtype=np.int64)
    h0 = hash_at(seed, 91, idx)
    start = np.sort((h0 % np.uint64(n_sites + 6)).astype(np.int64) - 2)          # -2 .. n_sites + 3
    h1 = hash_at(seed, 92, idx)
    ln = 1 + (h1 & U64(15)).astype(np.int64) % 12
    cnt = 1 + ((h1 >> U64(8)) % U64(40)).astype(np.int64)
    cum = None if sizes is None else np.cumsum(sizes)
    lines = []
    alphabet = 'CCCTTTH.'
    for i in range(n_reads):
        hp = int(hash_at(seed, 93, np.array([i], dtype=np.int64))[0])
        pat = ''.join(alphabet[(hp >> (3 * k)) & 7] for k in range(int(ln[i])))
        chrom = 'chr1' if cum is None else names[int(np.searchsorted(cum, max(int(start[i]), 1), 'left').clip(0, len(names) - 1))]
        lines.append('%s\t%d\t%s\t%d' % (chrom, start[i], pat, cnt[i]))
    return lines
