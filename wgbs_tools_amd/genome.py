"""Reference-genome access for the segment path, without tabix.

Restates the parts of the reference's L2 helpers that `segment` touches:
  GenomeRefPaths          utils_wgbs.py:53-115   (directory layout, CpG.chrome.size table, default symlink)
  index2chrom             genomic_region.py:10-12
  GenomicRegion           genomic_region.py:23-247 (-s / -r parsing; the tabix lookups become searches in the
                                                   in-memory loci array)
  beta_sanity_check       utils_wgbs.py:293-304
  add_loci (BED emit)     src/cpg2bed/add_loci.cpp:22-57, cpg_dict.cpp:99-131

The reference shells out to `tabix` for every locus lookup (genomic_region.py:140-152,190-208; cpg_dict.cpp:40-56).
Here the second column of CpG.bed.gz is read once into a uint32 array (cached next to it as `loci.u32`) — the same
array the GPU path needs anyway.
"""
import gzip
import os
import os.path as op
import re
import sys

import numpy as np


class IllegalArgumentError(ValueError):
    pass


def eprint(*args, **kwargs):
    print(*args, file=sys.stderr, **kwargs)


def references_root():
    """Where references/<name>/ directories live: $WGBSTOOLS_REFERENCES, else <repo>/references (the reference
    keeps them next to its sources, utils_wgbs.py:90-92)."""
    env = os.environ.get('WGBSTOOLS_REFERENCES')
    if env:
        return env
    return op.join(op.dirname(op.dirname(op.abspath(__file__))), 'references')


class GenomeRefPaths:
    """utils_wgbs.py:53-115.  `name` may also be a path to a genome directory."""

    def __init__(self, name=None):
        self.genome = name
        self.refdir = self.build_dir()
        self.dict_path = self.join('CpG.bed.gz')
        self.chrom_cpg_sizes = self.join('CpG.chrome.size')
        self.chrom_sizes = self.join('chrome.size')
        self.revdict_path = self.join('rev.CpG.bed.gz', validate=False)
        self.ilmn2cpg_dict = self.join('ilmn2CpG.tsv.gz', validate=False)        # utils_wgbs.py:67
        self._names = None
        self._sizes = None
        self._bp_sizes = None
        self._loci = None

    def join(self, fpath, validate=True):
        path = op.join(self.refdir, fpath)
        if not op.isfile(path):
            if op.isfile(path + '.gz'):
                path += '.gz'
            else:
                if validate:
                    raise IllegalArgumentError('Invalid reference path: ' + path)
                path = None
        return path

    def build_dir(self):
        if not self.genome:
            self.genome = 'default'
        if op.isdir(self.genome) and op.isfile(op.join(self.genome, 'CpG.chrome.size')):
            refdir = op.realpath(self.genome)
            self.genome = op.basename(refdir)
            return refdir
        refdir = op.join(references_root(), self.genome)
        if self.genome == 'default':
            if not op.islink(refdir):
                raise IllegalArgumentError('Invalid reference name: default (no default genome is set)')
            self.genome = os.readlink(refdir)
            refdir = op.realpath(refdir)
        if not op.isdir(refdir):
            raise IllegalArgumentError(f'Invalid reference name: {self.genome}')
        return refdir

    def _load_tables(self):
        if self._names is None:
            names, sizes = [], []
            with open(self.chrom_cpg_sizes) as f:
                for line in f:
                    if line.strip():
                        c, s = line.rstrip('\n').split('\t')[:2]
                        names.append(c)
                        sizes.append(int(s))
            self._names = names
            self._sizes = np.array(sizes, dtype=np.int64)
            bp = {}
            with open(self.chrom_sizes) as f:
                for line in f:
                    if line.strip():
                        c, s = line.rstrip('\n').split('\t')[:2]
                        bp[c] = int(s)
            self._bp_sizes = bp

    def get_chroms(self):
        self._load_tables()
        return tuple(self._bp_sizes.keys())

    def get_chrom_cpg_sizes(self):
        """(names, sizes) of CpG.chrome.size, file order."""
        self._load_tables()
        return self._names, self._sizes

    def get_chrom_size(self, chrom):
        self._load_tables()
        return self._bp_sizes[chrom]

    def get_nr_sites(self):
        self._load_tables()
        return int(self._sizes.sum())

    def cum_sizes(self):
        self._load_tables()
        return np.cumsum(self._sizes)

    def index2chrom(self, site):
        """genomic_region.py:10-12: chromosome of 1-based CpG index `site`."""
        self._load_tables()
        return self._names[int(np.searchsorted(self.cum_sizes(), site))]

    def loci(self):
        """uint32 array: loci[i] = bp position of CpG i+1 (column 2 of CpG.bed.gz)."""
        if self._loci is None:
            cache = op.join(self.refdir, 'loci.u32')
            n = self.get_nr_sites()
            if op.isfile(cache) and op.getsize(cache) == 4 * n and op.getmtime(cache) >= op.getmtime(self.dict_path):
                self._loci = np.memmap(cache, dtype=np.uint32, mode='r')     # pages come in as the upload / the BED writer touch them
            else:
                self._loci = _parse_dict_loci(self.dict_path, n)
                try:
                    self._loci.tofile(cache)
                except OSError:
                    pass
        return self._loci


def _parse_dict_loci(path, n_expected):
    import pandas as pd
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'rb') as f:
        df = pd.read_csv(f, sep='\t', header=None, usecols=[1], dtype=np.int64, engine='c')
    loci = df.iloc[:, 0].values
    if loci.size != n_expected:
        raise IllegalArgumentError(f'{path} holds {loci.size:,} CpGs but CpG.chrome.size sums to {n_expected:,}')
    return loci.astype(np.uint32)


def beta_sanity_check(beta_path, genome):
    """utils_wgbs.py:293-304"""
    nr_sites_in_beta = op.getsize(beta_path) // 2
    if beta_path.endswith('.lbeta'):
        nr_sites_in_beta /= 2
    if int(nr_sites_in_beta) != genome.get_nr_sites():
        eprint(f'[wt beta] WARNING: beta file size ({nr_sites_in_beta:,} sites)\n'
               f'          incomatible with current genome reference '
               f'({genome.get_nr_sites():,} sites)')
        return False
    return True


# ------------------------------------------------------------------------------------------------------------
# -s / -r / --array_id: one resolver per input kind, all table-driven
#
# The ACCEPTED spellings, the order of the checks and every message are the reference's (genomic_region.py:70-232: they are what
# a user of the CLI sees); how they are organised is not: the reference walks a chain of methods that mutate one object and ask
# `tabix` per locus, here each kind of input is a pure function (text, genome) -> Span over the in-memory tables, driven by the
# small tables below, and GenomicRegion only carries the answer.
# ------------------------------------------------------------------------------------------------------------
_CHROM = r'(chr)?([\d]+|[XYM]|(MT))'
# region spellings (commas removed first), tried in this order; `kind` says what the numeric groups mean
_REGION_FORMS = (
    ('chrom', re.compile(r'^' + _CHROM + r'$')),                                  # a whole chromosome: 1 .. its length
    ('point', re.compile(r'^(?P<chrom>' + _CHROM + r'):(?P<a>[\d]+)$')),           # chr:pos  ==  chr:pos-(pos+1)
    ('range', re.compile(r'^(?P<chrom>' + _CHROM + r'):(?P<a>[\d]+)-(?P<b>[\d]+)$')),
)
# site spellings (commas removed first).  The range pattern is a PREFIX match, as upstream (re.match without '$').
_SITE_FORMS = (
    ('range', re.compile(r'([\d]+)-([\d]+)')),
    ('single', re.compile(r'^[0-9]+$')),
)


class Span:
    """What every resolver returns: chromosome, 1-based CpG range [s1, s2), and the base-pair range it prints."""
    __slots__ = ('chrom', 'sites', 'bp')

    def __init__(self, chrom, sites, bp):
        self.chrom, self.sites, self.bp = chrom, sites, bp

    @property
    def text(self):
        return f'{self.chrom}:{self.bp[0]}-{self.bp[1]}'


def _site_pair(genome, text):
    """'a-b' | 'a' (commas allowed) -> (s1, s2) with the reference's range rule and its message."""
    if not text:
        raise IllegalArgumentError(f'Empty sites string: {text}')
    plain = text.replace(',', '')
    pair = None
    for kind, rx in _SITE_FORMS:
        m = rx.match(plain)
        if m and kind == 'range':
            pair = (int(m.group(1)), int(m.group(2)))
        elif m and '-' not in plain:
            pair = (int(plain), int(plain) + 1)
        if pair:
            break
    if pair is None:
        raise IllegalArgumentError(f'sites must be of format: "start-end" or "site" .\nGot: {plain}')
    s1, s2 = pair
    top = genome.get_nr_sites() + 1
    if not (top >= s2 >= s1 >= 1):
        raise IllegalArgumentError(f'sites violate the constraints: {top} >= {s2} > {s1} >= 1')
    return (s1, s2 + 1) if s1 == s2 else (s1, s2)


def locus_of_site(genome, index):
    """(chromosome, bp position) of 1-based CpG `index` — the `tabix rev.CpG.bed.gz` lookup of the reference, from the loci array."""
    index = int(index)
    if not (genome.get_nr_sites() + 1 >= index >= 1):
        eprint('Invalid site index:', index)
        raise IllegalArgumentError('Out of range site index:', index)
    loci = genome.loci()
    if index > loci.size:
        raise IllegalArgumentError(f'Failed retrieving locus for site {index}')
    return genome.index2chrom(index), int(loci[index - 1])


def span_of_sites(genome, text):
    s1, s2 = _site_pair(genome, text)
    (c1, first), (c2, last) = locus_of_site(genome, s1), locus_of_site(genome, s2 - 1)
    if c1 != c2:
        eprint(f'ERROR: sites range cross chromosomes! ({s1}, {s2})')
        raise IllegalArgumentError('Invalid sites input')
    return Span(c1, (s1, s2), (first, last + 1))


def _bp_request(genome, text):
    """A region spelling -> (chromosome, printable region, from, to), unknown chromosomes refused with the reference's text."""
    plain = text.replace(',', '')
    known = genome.get_chroms()
    for kind, rx in _REGION_FORMS:
        m = rx.match(plain)
        if not m:
            continue
        if kind == 'chrom':
            if plain not in known:
                raise IllegalArgumentError(f'Unknown chromosome: {plain}')
            return plain, plain, 1, genome.get_chrom_size(plain)
        a = int(m.group('a'))
        shown = plain if kind == 'range' else f'{plain}-{a + 1}'
        if m.group('chrom') not in known:
            raise IllegalArgumentError(f'Unknown chromosome: {shown}')
        return m.group('chrom'), shown, a, int(m.group('b')) if kind == 'range' else a + 1
    raise IllegalArgumentError(f'Invalid genomic region: {plain}')


def sites_in_bp_range(genome, chrom, a, b, shown):
    """CpGs with a <= locus <= b on `chrom`, as the 1-based range the reference derives from `tabix CpG.bed.gz chr:a-b`
    (genomic_region.py:140-161): first row's index .. last row's index, plus one unless the last CpG sits exactly on b."""
    names, sizes = genome.get_chrom_cpg_sizes()
    ci = names.index(chrom)
    lo = int(sizes[:ci].sum())
    loci = genome.loci()[lo:lo + int(sizes[ci])]
    i0, i1 = int(np.searchsorted(loci, a, 'left')), int(np.searchsorted(loci, b, 'right'))     # rows i0 .. i1-1
    first = lo + i0 + 1
    end = lo + i1 + (1 if i1 > i0 and int(loci[i1 - 1]) < b else 0)
    if i1 <= i0 or first == end:
        raise IllegalArgumentError(f'Invalid genomic region: {shown}. No CpGs in range')
    return _site_pair(genome, f'{first}-{end}')


def span_of_region(genome, text):
    chrom, shown, a, b = _bp_request(genome, text)
    for bad, why in ((b <= a, 'end before start'), (b > genome.get_chrom_size(chrom) or a < 1, 'Out of range')):
        if bad:
            raise IllegalArgumentError(f'Invalid genomic region: {text}. {why}')
    sp = Span(chrom, sites_in_bp_range(genome, chrom, a, b, shown), (a, b))
    return sp, shown


def site_of_array_id(genome, array_id):
    """--array_id cg00001755: the CpG index of an Illumina array probe, from the genome's map file (genomic_region.py:212-232:
    `gunzip -c ilmn2CpG.tsv.gz | grep -w <id> | cut -f2`, which must come out as ONE integer)."""
    if not (array_id.startswith('cg') and len(array_id) > 2 and array_id[2:].isdigit()):
        eprint(f'ERROR: Invalid Illumina array id: {array_id}')
        raise IllegalArgumentError('Invalid Illumina array ID')
    idict = genome.ilmn2cpg_dict
    if idict is None or not op.isfile(idict):
        raise IllegalArgumentError(f'Could not find Illumina map file: {idict}')
    word = re.compile(r'(?<![A-Za-z0-9_])' + re.escape(array_id) + r'(?![A-Za-z0-9_])')      # grep -w
    with gzip.open(idict, 'rt') as f:
        hits = [line.rstrip('\n') for line in f if word.search(line)]
    second = [h.split('\t')[1] if '\t' in h else h for h in hits]                              # cut -f2 (a line without a tab passes whole)
    try:
        return int('\n'.join(second).strip())
    except ValueError as e:
        cmd = f'gunzip -c {idict} | grep -w {array_id} | cut -f2'
        raise IllegalArgumentError(f'Failed retrieving locus for site {array_id} with command:\n{cmd}\n{e}')


class GenomicRegion:
    """What `segment` / `convert` read off genomic_region.py:23-247: .sites (None = whole genome), .chrom, .region_str, .bp_tuple,
    .nr_sites, is_whole(), str().  Built from parsed CLI arguments (-s / -r / --array_id, in that order of precedence) or from one
    explicit `region=` / `sites=` text."""

    def __init__(self, args=None, region=None, sites=None, genome=None):
        self.chrom = self.sites = self.region_str = self.bp_tuple = None
        if args is not None:                                      # the CLI: an empty option is no option
            self.genome = genome if genome is not None else GenomeRefPaths(args.genome)
            sites, region = getattr(args, 'sites', None) or None, getattr(args, 'region', None) or None
            if sites is None and region is None and getattr(args, 'array_id', None):
                sites = str(site_of_array_id(self.genome, args.array_id))
            if sites is not None:
                region = None
        else:                                                     # one explicit text: whatever it holds is parsed (and may be refused)
            self.genome = genome
            if region is None and sites is None:
                raise IllegalArgumentError(f'Invalid GR init {region}')
            if region is not None:
                sites = None
        if sites is not None:
            sp = span_of_sites(self.genome, sites)
            self._take(sp, sp.text)
        elif region is not None:
            self._take(*span_of_region(self.genome, region))
        self.nr_sites = None if self.sites is None else self.sites[1] - self.sites[0]

    def _take(self, span, shown):
        self.chrom, self.sites, self.bp_tuple, self.region_str = span.chrom, span.sites, span.bp, shown

    def is_whole(self):
        return self.sites is None

    def __str__(self):                                            # genomic_region.py:239-247 (no annotation tracks here)
        if self.sites is None:
            return 'Whole genome'
        s1, s2 = self.sites
        return f'{self.region_str} - {self.bp_tuple[1] - self.bp_tuple[0] + 1:,}bp, {s2 - s1:,}CpGs: {s1}-{s2}'


def blocks_to_bed_lines(genome, start_cpg, end_cpg):
    """add_loci (src/cpg2bed/add_loci.cpp:22-57): rows `chr\\tstart\\tend\\tstartCpG\\tendCpG\\n` with
    start = loci[startCpG-1], end = loci[endCpG-2]+1 (start+2 for an empty block)."""
    names, sizes = genome.get_chrom_cpg_sizes()
    borders = np.cumsum(sizes)
    loci = genome.loci()
    s = np.asarray(start_cpg, dtype=np.int64)
    e = np.asarray(end_cpg, dtype=np.int64)
    nr = int(borders[-1])
    for arr, what in ((s, 'startCpG'), (e, 'endCpG')):
        if arr.size and arr.min() < 1:
            raise RuntimeError(f'[wt add_loci] {what} < 1')
    if (e < s).any():
        raise RuntimeError('[wt add_loci] endCpG < startCpG')
    if s.size and (s.max() > nr or e.max() > nr + 1):
        raise RuntimeError('[ cpg_dict ] Could not find chromosome for site')

    def loc2chrom_idx(x):                                      # cpg_dict.cpp:118-131
        idx = np.searchsorted(borders, x, 'left')
        return np.where(x == nr + 1, len(names) - 1, idx)
    c1 = loc2chrom_idx(s)
    c2 = loc2chrom_idx(e)
    cross = (c1 != c2) & (e - 1 != borders[c1])                # add_loci.cpp:42-49
    if cross.any():
        raise RuntimeError('[wt add_loci] line %d: Cross chromosomes' % int(np.flatnonzero(cross)[0]))
    start = loci[s - 1].astype(np.int64)
    end = np.where(e == s, start + 2, loci[np.maximum(e - 2, 0)].astype(np.int64) + 1)
    chrom = np.array(names, dtype=object)[c1]
    return chrom, start, end


def write_bed(genome, start_cpg, end_cpg, out_path=None):
    """The blocks as BED rows into out_path (None / sys.stdout: standard output) through the library's add_loci
    (include/wgbsseg.h: wgbsseg_add_loci), which restates the reference's add_loci binary; its validation failures
    surface as RuntimeError with the reference's messages (blocks_to_bed_lines above is the same rule set in numpy,
    kept for callers that want the columns rather than the text)."""
    from . import _lib
    names, sizes = genome.get_chrom_cpg_sizes()
    to_stdout = out_path is None or out_path is sys.stdout
    if to_stdout:
        sys.stdout.flush()
    try:
        _lib.add_loci(genome.loci(), names, np.cumsum(sizes), start_cpg, end_cpg, None if to_stdout else out_path)
    except _lib.SegmentorError as e:
        raise RuntimeError(e.msg)
