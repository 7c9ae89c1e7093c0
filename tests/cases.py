"""Seed-generated parity cases shared by the golden generator (tests/golden/make_golden.py) and the tests.

A case spec is a small dict; ``build_case(spec)`` regenerates its inputs (beta slices per sample + loci) from
the seed, so fixtures store only the spec, an input checksum and the expected borders.
"""
import numpy as np

from wgbs_tools_amd import synth

SEED = 20260926


def _loci_for(spec):
    kind = spec.get('loci', 'hg19like')
    n, a = spec['n'], spec['a']
    if kind == 'hg19like':
        # one long synthetic chromosome; take [a, a+n)
        total = a + n
        loci = synth.synth_loci(spec.get('loci_seed', SEED), [total])
        return loci[a:a + n].copy()
    if kind == 'dense':
        # CpG-island-like: gaps 2..9 bp -> hundreds of CpGs inside max_bp
        idx = np.arange(a, a + n, dtype=np.int64)
        gap = 2 + (synth.hash_at(spec.get('loci_seed', SEED), 77, idx) & np.uint64(7)).astype(np.int64)
        return (10000 + np.cumsum(gap)).astype(np.uint32)
    if kind == 'hg19like_islands':
        # sparse background with a CpG island (gaps 2..9 bp, ~360 CpGs inside max_bp=2000) every 3000 sites:
        # batches of the DP with and without windows > 64 alternate inside one chunk
        idx = np.arange(a, a + n, dtype=np.int64)
        total = a + n
        base = np.diff(np.concatenate([[0], synth.synth_loci(spec.get('loci_seed', SEED), [total]).astype(np.int64)]))[a:a + n]
        dense = 2 + (synth.hash_at(spec.get('loci_seed', SEED), 79, idx) & np.uint64(7)).astype(np.int64)
        island = (idx % 3000) >= 2600
        gap = np.where(island, dense, np.maximum(base, 2))
        return (10000 + np.cumsum(gap)).astype(np.uint32)
    if kind == 'equal_runs':
        # repeated positions (distance 0) -- legal for the reference: dists[i+j]-dists[i] == 0 <= max_bp
        idx = np.arange(a, a + n, dtype=np.int64)
        gap = (synth.hash_at(spec.get('loci_seed', SEED), 78, idx) & np.uint64(3)).astype(np.int64) * 60
        return (10000 + np.cumsum(gap)).astype(np.uint32)
    raise ValueError(kind)


def build_case(spec):
    """-> (slices: list of uint8 [n,2] per sample, loci uint32[n])"""
    n, a = spec['n'], spec['a']
    seed = spec.get('seed', SEED)
    slices = [synth.synth_betas(seed, s, a, a + n) for s in spec['samples']]
    for (s_idx, x0, x1) in spec.get('zero_ranges', []):       # zero-coverage stretches in one sample
        slices[s_idx][x0:x1, :] = 0
    for s_idx in spec.get('zero_samples', []):                # an all-zero sample
        slices[s_idx][:, :] = 0
    for (s_idx, x0, x1) in spec.get('saturate_ranges', []):   # 255/255 sites
        slices[s_idx][x0:x1, :] = 255
    loci = _loci_for(spec)
    return slices, loci


def case_checksum(slices, loci):
    return synth.checksum(loci, *slices)


# name -> spec.  pcount/max_cpg/max_bp are the segmentor's own flags (main.cpp:44-48,69-84).
CHUNK_CASES = {
    'tiny':           dict(n=400, a=1000, samples=[0, 1, 2], pcount=15.0, max_cpg=50, max_bp=700),
    'n1':             dict(n=1, a=5, samples=[0, 1], pcount=15.0, max_cpg=1000, max_bp=2000),
    'n2':             dict(n=2, a=5, samples=[0], pcount=15.0, max_cpg=1000, max_bp=2000),
    'n65':            dict(n=65, a=50, samples=[0, 1, 2], pcount=15.0, max_cpg=1000, max_bp=2000),
    'max_cpg2':       dict(n=500, a=300, samples=[0, 1], pcount=15.0, max_cpg=2, max_bp=2000),
    'max_cpg_binds':  dict(n=3000, a=7000, samples=[0, 1, 2, 3], pcount=15.0, max_cpg=10, max_bp=2000),
    'pcount0':        dict(n=3000, a=20000, samples=[0, 1, 2, 3], pcount=0.0, max_cpg=1000, max_bp=2000),
    'pcount_half':    dict(n=3000, a=20000, samples=[0, 1, 2, 3], pcount=0.5, max_cpg=1000, max_bp=2000),
    'pcount15':       dict(n=3000, a=20000, samples=[0, 1, 2, 3], pcount=15.0, max_cpg=1000, max_bp=2000),
    'zero_stretch':   dict(n=4000, a=40000, samples=[0, 1, 2], pcount=15.0, max_cpg=1000, max_bp=2000,
                           zero_ranges=[(0, 500, 900), (1, 0, 130), (2, 3900, 4000), (0, 2000, 2001)]),
    'zero_sample':    dict(n=3000, a=50000, samples=[0, 1, 2, 3], pcount=15.0, max_cpg=1000, max_bp=2000,
                           zero_samples=[2]),
    'all_zero':       dict(n=700, a=50000, samples=[0, 1], pcount=15.0, max_cpg=1000, max_bp=2000,
                           zero_samples=[0, 1]),
    'saturated':      dict(n=2500, a=60000, samples=[0, 1, 2], pcount=15.0, max_cpg=1000, max_bp=2000,
                           saturate_ranges=[(0, 100, 1400), (1, 0, 2500)]),
    'single_sample':  dict(n=5000, a=90000, samples=[5], pcount=15.0, max_cpg=1000, max_bp=2000),
    'dense_w_gt_64':  dict(n=3000, a=0, samples=[0, 1, 2], pcount=15.0, max_cpg=1000, max_bp=2000, loci='dense'),
    'dense_small_bp': dict(n=3000, a=0, samples=[0, 1, 2], pcount=15.0, max_cpg=1000, max_bp=300, loci='dense'),
    'equal_loci':     dict(n=1500, a=0, samples=[0, 1], pcount=15.0, max_cpg=1000, max_bp=2000, loci='equal_runs'),
    'dense_bp500':    dict(n=5000, a=0, samples=[0, 1, 2], pcount=15.0, max_cpg=1000, max_bp=500, loci='dense'),
    'island_mix':     dict(n=20000, a=0, samples=[0, 1, 2, 3], pcount=15.0, max_cpg=1000, max_bp=2000, loci='hg19like_islands'),
    'deep':           dict(n=6000, a=120000, samples=[0, 1, 2, 3], pcount=15.0, max_cpg=5000, max_bp=100000000),
    'deep_long':      dict(n=20000, a=300000, samples=[0, 1, 2], pcount=15.0, max_cpg=3000, max_bp=100000000),
    'n33_samples':    dict(n=2000, a=130000, samples=list(range(33)), pcount=15.0, max_cpg=1000, max_bp=2000),
    'default_chunk':  dict(n=60000, a=200000, samples=list(range(8)), pcount=15.0, max_cpg=1000, max_bp=2000),
    # atlas-scale sample counts (BASELINE.json configs[3], [4]): 7 and 16 LDS sample groups in the scoring kernel
    'n200_samples':   dict(n=1500, a=140000, samples=list(range(200)), pcount=15.0, max_cpg=1000, max_bp=2000),
    'n512_samples':   dict(n=800, a=150000, samples=list(range(512)), pcount=15.0, max_cpg=1000, max_bp=2000),
    'n200_islands':   dict(n=4000, a=0, samples=list(range(200)), pcount=15.0, max_cpg=1000, max_bp=2000, loci='hg19like_islands',
                           loci_seed=20260927),
    'n512_deep':      dict(n=1200, a=160000, samples=list(range(512)), pcount=15.0, max_cpg=5000, max_bp=1000000),
    'n200_pcount0':   dict(n=1000, a=170000, samples=list(range(200)), pcount=0.0, max_cpg=1000, max_bp=2000),
}

# Loci that are NOT ascending inside a chunk (hand-made genomes, a chunk laid across two chromosomes): the reference bars an extension
# whose locus lies behind the start's or more than max_bp ahead of it and leaves that site out of the start's running sums
# (segmentor.cpp:114-117).  `disorder` names what is done to the ascending loci of the base spec (disorder_loci below).
DISORDER_CASES = {
    'one_step_back':   dict(n=600, a=1000, samples=[0, 1, 2], pcount=15.0, max_cpg=50, max_bp=700, disorder=('step_back', 200, 1)),
    'back_run':        dict(n=700, a=3000, samples=[0, 1], pcount=15.0, max_cpg=1000, max_bp=2000, disorder=('descending_run', 300, 40)),
    'two_chromosomes': dict(n=900, a=5000, samples=[0, 1, 2], pcount=15.0, max_cpg=1000, max_bp=2000, disorder=('restart', 450, 5)),
    'restart_pcount0': dict(n=500, a=7000, samples=[0, 1], pcount=0.0, max_cpg=60, max_bp=2000, disorder=('restart', 123, 5)),
    'swapped_pairs':   dict(n=800, a=9000, samples=[0], pcount=0.5, max_cpg=1000, max_bp=2000, disorder=('swap_pairs', 64, 16)),
    'shuffled_dense':  dict(n=400, a=0, samples=[0, 1, 2], pcount=15.0, max_cpg=300, max_bp=500, loci='dense', disorder=('shuffle_window', 100, 200)),
    'equal_then_back': dict(n=600, a=0, samples=[0, 1], pcount=15.0, max_cpg=1000, max_bp=2000, loci='equal_runs', disorder=('step_back', 301, 3)),
    'back_at_the_end': dict(n=300, a=11000, samples=[0, 1, 2, 3], pcount=15.0, max_cpg=1000, max_bp=2000, disorder=('step_back', 299, 1)),
    'near_int_max':    dict(n=300, a=12000, samples=[0, 1], pcount=15.0, max_cpg=1000, max_bp=2000, disorder=('huge_then_small', 150, 0)),
    'n33_restart':     dict(n=500, a=13000, samples=list(range(33)), pcount=15.0, max_cpg=1000, max_bp=2000, disorder=('restart', 250, 5)),
}


def disorder_loci(loci, how):
    """The loci of a DISORDER_CASES spec: ascending `loci` with one stretch put out of order."""
    kind, at, arg = how
    out = loci.astype(np.int64).copy()
    if kind == 'step_back':              # `arg` sites from `at` on lie 1000 bp (or as far as the values allow) behind their place
        out[at:at + arg] -= min(1000, int(out[at]) - 1)
    elif kind == 'descending_run':       # a run of `arg` sites in descending order
        out[at:at + arg] = out[at:at + arg][::-1]
    elif kind == 'restart':              # the positions start again at `arg` (a second chromosome inside the chunk)
        out[at:] -= out[at] - arg
    elif kind == 'swap_pairs':           # every `arg`-th pair from `at` on swapped
        for x in range(at, len(out) - 1, arg):
            out[x], out[x + 1] = out[x + 1], out[x]
    elif kind == 'shuffle_window':       # `arg` sites from `at` on in a seeded random order
        perm = np.random.default_rng(at * 1000 + arg).permutation(arg)
        out[at:at + arg] = out[at:at + arg][perm]
    elif kind == 'huge_then_small':      # positions up to INT_MAX (what load_dists' std::stoi takes, segmentor.cpp:44) in the first half, small ones after: the unsigned difference wraps (:114)
        out[:at] += 2147483000 - int(out[at - 1])
    else:
        raise ValueError(kind)
    assert out.min() >= 0 and out.max() < 2 ** 32
    return out.astype(np.uint32)


def build_disorder_case(spec):
    slices, loci = build_case({k: v for k, v in spec.items() if k != 'disorder'})
    return slices, disorder_loci(loci, spec['disorder'])


# Chunks INSIDE a larger resident world (the way the driver calls the library: `-s start0 -n len` on whole-genome files), placed on the
# kernels' boundaries: carries of the scan pass sit at every 128th ABSOLUTE site, scoring tiles and units begin at chunk-relative
# multiples of 16 / 64 / 128.  `ends` are absolute end sites, `lens` chunk lengths: every (end - len, len) with len <= end is a chunk.
# (Found necessary by tools/extra_fuzz.py seed 5751: len = 1 mod 16 ending on a multiple of 128 with windows > 60.)
OFFSET_CASES = {
    'dense_1sample':  dict(n=7000, a=0, samples=[0], pcount=3.9999998, max_cpg=129, max_bp=100000, loci='dense',
                           zero_ranges=[(0, 2555, 2560), (0, 4606, 4608)],
                           ends=[2560, 3008, 4608, 6400, 6401, 7000], lens=[1, 17, 65, 129, 145, 193, 241, 1281, 1345]),
    'dense_3samples': dict(n=5000, a=7000, samples=[0, 1, 2], pcount=15.0, max_cpg=1000, max_bp=2000, loci='dense',
                           ends=[1280, 2561, 3840, 4992], lens=[1, 65, 129, 257, 449, 1217]),
    'islands_8':      dict(n=9000, a=0, samples=list(range(8)), pcount=15.0, max_cpg=1000, max_bp=2000, loci='hg19like_islands',
                           ends=[3072, 5888, 6016, 8960], lens=[65, 385, 1025, 2945]),
}


def offset_chunks(spec):
    return [(e - ln, ln) for e in spec['ends'] for ln in spec['lens'] if ln <= e]


# chr21-shaped multi-chunk case (BASELINE.json configs[1]): 400,000 CpGs x 8 samples in default 60,000-site chunks
CHR21 = dict(n=400000, a=0, samples=list(range(8)), pcount=15.0, max_cpg=1000, max_bp=2000, chunk=60000)


def lbeta_twin(data):
    """uint16 [n, 2] twin of a uint8 beta array for the .lbeta-input cases: (meth, cov) of site i times k(i) = 1 + (i * 2654435761 mod 997)
    (<= 54 * 997 < 65536; keeps meth <= cov)."""
    i = np.arange(data.shape[0], dtype=np.uint64)
    k = (1 + (i * np.uint64(2654435761)) % np.uint64(997)).astype(np.uint32)
    return (data.astype(np.uint32) * k[:, None]).astype(np.uint16)


MARKER_SPEC = dict(seed=20260929, n_sites=60000, groups=[('Liver', 4), ('Blood', 3), ('Colon', 3), ('Lung', 2), ('Solo', 1)])


def marker_world(td, spec=MARKER_SPEC):
    """Inputs of the find_markers cases, written under `td`: beta files of several groups of samples with group-specific
    differentially methylated blocks, a blocks table (with blocks of every length, a few uncovered), a groups csv.
    -> dict(betas, blocks, groups, spec).  Everything from integer hashes of the seed: regenerated, never stored."""
    import os.path as op
    seed, n = spec['seed'], spec['n_sites']
    U64 = np.uint64
    # blocks: consecutive, 3..40 sites, every 37th gap skipped
    starts, pos, k = [], 1, 0
    while pos < n - 50:
        ln = 3 + int(synth.hash_at(seed, 201, np.array([k]))[0] % U64(38))
        starts.append((pos, pos + ln))
        pos += ln + (5 if k % 37 == 36 else 0)
        k += 1
    nb = len(starts)
    bidx = np.arange(nb, dtype=np.int64)
    hb = synth.hash_at(seed, 202, bidx)
    base = np.where((hb & U64(1)) == 0, 205, 30).astype(np.int64)              # background level of the block: high or low (of 256)
    names, groups = [], []
    for g, cnt in spec['groups']:
        for i in range(cnt):
            names.append('%s_%d' % (g, i + 1)); groups.append(g)
    site_block = np.zeros(n, dtype=np.int64) - 1
    for b, (a, e) in enumerate(starts):
        site_block[a - 1:e - 1] = b
    betas = []
    gnames = [g for g, _ in spec['groups']]
    for si, (nm, g) in enumerate(zip(names, groups)):
        gi = gnames.index(g)
        special = ((hb >> U64(8)) % U64(9)).astype(np.int64) == gi             # this group's differential blocks
        lvl_b = np.where(special, 235 - base, base)                            # flipped level
        jit = (synth.hash_at(seed, 210 + si, bidx) % U64(41)).astype(np.int64) - 20
        lvl_b = np.clip(lvl_b + jit, 0, 255)
        idx = np.arange(n, dtype=np.int64)
        hc = synth.hash_at(seed, 300 + si, idx)
        cov = (4 + (hc % U64(28))).astype(np.int64)
        cov = np.where(((hc >> U64(20)) % U64(23)) == 0, 0, cov)               # ~4 % uncovered sites
        lowcov_block = ((synth.hash_at(seed, 400 + si, bidx) % U64(29)) == 0)  # whole blocks this sample does not cover
        lvl = np.where(site_block >= 0, lvl_b[np.maximum(site_block, 0)], 128)
        cov = np.where((site_block >= 0) & lowcov_block[np.maximum(site_block, 0)], 0, cov)
        meth = np.zeros(n, dtype=np.int64)
        for t in range(4):                                                     # cov <= 31: four groups of 8 byte-trials
            hbt = synth.hash_at(seed, 500 + si, idx * 4 + t)
            for byte in range(8):
                trial = ((hbt >> U64(8 * byte)) & U64(0xFF)).astype(np.int64)
                meth += ((trial < lvl) & (t * 8 + byte < cov)).astype(np.int64)
        arr = np.stack([meth, cov], axis=1).astype(np.uint8)
        p = op.join(td, nm + '.beta')
        arr.tofile(p)
        betas.append(p)
    bp = op.join(td, 'blocks.bed')
    with open(bp, 'w') as f:
        for b, (a, e) in enumerate(starts):
            f.write('chr1\t%d\t%d\t%d\t%d\n' % (1000 + 50 * a, 1000 + 50 * e + 17 * (b % 5), a, e))
    gp = op.join(td, 'groups.csv')
    with open(gp, 'w') as f:
        f.write('name,group\n')
        for nm, g in zip(names, groups):
            f.write('%s,%s\n' % (nm, g))
    return dict(betas=betas, blocks=bp, groups=gp, spec=spec, n_blocks=nb)
