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


class GenomicRegion:
    """The subset of genomic_region.py:23-247 that `segment` / `convert` use: -s/--sites, -r/--region and --array_id."""

    def __init__(self, args=None, region=None, sites=None, genome=None):
        self.chrom = None
        self.sites = None
        self.region_str = None
        self.bp_tuple = None
        if args is not None:
            self.genome = genome if genome is not None else GenomeRefPaths(args.genome)
            if getattr(args, 'sites', None):
                self.parse_sites(args.sites)
            elif getattr(args, 'region', None):
                self.parse_region(args.region)
            elif getattr(args, 'array_id', None):
                self.parse_array_id(args.array_id)
        else:
            self.genome = genome
            if region is not None:
                self.parse_region(region)
            elif sites is not None:
                self.parse_sites(sites)
            else:
                raise IllegalArgumentError(f'Invalid GR init {region}')
        self.nr_sites = None if self.sites is None else self.sites[1] - self.sites[0]

    def is_whole(self):
        return self.sites is None

    def parse_array_id(self, array_id):
        """--array_id cg00001755: the CpG index of an Illumina array probe, from the genome's map file (genomic_region.py:212-232:
        `gunzip -c ilmn2CpG.tsv.gz | grep -w <id> | cut -f2`, which must come out as ONE integer)."""
        import gzip
        if not (array_id.startswith('cg') and len(array_id) > 2 and array_id[2:].isdigit()):
            eprint(f'ERROR: Invalid Illumina array id: {array_id}')
            raise IllegalArgumentError('Invalid Illumina array ID')
        idict = self.genome.ilmn2cpg_dict
        if idict is None or not op.isfile(idict):
            raise IllegalArgumentError(f'Could not find Illumina map file: {idict}')
        word = re.compile(r'(?<![A-Za-z0-9_])' + re.escape(array_id) + r'(?![A-Za-z0-9_])')      # grep -w
        hits = []
        with gzip.open(idict, 'rt') as f:
            for line in f:
                if word.search(line):
                    fields = line.rstrip('\n').split('\t')
                    hits.append(fields[1] if len(fields) > 1 else line.rstrip('\n'))       # cut -f2 (a line without a tab passes whole)
        try:
            cpg_ind = int('\n'.join(hits).strip())
        except ValueError as e:
            cmd = f'gunzip -c {idict} | grep -w {array_id} | cut -f2'
            raise IllegalArgumentError(f'Failed retrieving locus for site {array_id} with command:\n{cmd}\n{e}')
        self.parse_sites(str(cpg_ind))

    def __str__(self):                                            # genomic_region.py:239-247 (no annotation tracks here)
        if self.sites is None:
            return 'Whole genome'
        s1, s2 = self.sites
        nr_bp = self.bp_tuple[1] - self.bp_tuple[0] + 1
        return f'{self.region_str} - {nr_bp:,}bp, {s2 - s1:,}CpGs: {s1}-{s2}'

    # genomic_region.py:163-187
    def _sites_str_to_tuple(self, sites_str):
        if not sites_str:
            raise IllegalArgumentError(f'Empty sites string: {sites_str}')
        sites_str = sites_str.replace(',', '')
        m = re.match(r'([\d]+)-([\d]+)', sites_str)
        if m:
            site1, site2 = int(m.group(1)), int(m.group(2))
        elif '-' not in sites_str and sites_str.isdigit():
            site1 = int(sites_str)
            site2 = site1 + 1
        else:
            raise IllegalArgumentError(f'sites must be of format: "start-end" or "site" .\nGot: {sites_str}')
        nr = self.genome.get_nr_sites()
        if not nr + 1 >= site2 >= site1 >= 1:
            msg = 'sites violate the constraints: '
            msg += f'{nr + 1} >= {site2} > {site1} >= 1'
            raise IllegalArgumentError(msg)
        if site1 == site2:
            site2 += 1
        return site1, site2

    # genomic_region.py:189-208
    def index2locus(self, index):
        index = int(index)
        if not self.genome.get_nr_sites() + 1 >= index >= 1:
            eprint('Invalid site index:', index)
            raise IllegalArgumentError('Out of range site index:', index)
        chrom = self.genome.index2chrom(index)
        loci = self.genome.loci()
        if index > loci.size:
            raise IllegalArgumentError(f'Failed retrieving locus for site {index}')
        return chrom, int(loci[index - 1])

    # genomic_region.py:70-88
    def parse_sites(self, sites_str):
        s1, s2 = self._sites_str_to_tuple(sites_str)
        self.chrom, region_from = self.index2locus(s1)
        chrom2, region_to = self.index2locus(s2 - 1)
        region_to += 1
        if self.chrom != chrom2:
            eprint(f'ERROR: sites range cross chromosomes! ({s1}, {s2})')
            raise IllegalArgumentError('Invalid sites input')
        self.sites = (s1, s2)
        self.region_str = f'{self.chrom}:{region_from}-{region_to}'
        self.bp_tuple = (region_from, region_to)

    # genomic_region.py:94-123
    def find_region_format(self, region):
        region = region.replace(',', '')
        if re.match(r'^(chr)?([\d]+|[XYM]|(MT))$', region):
            if region not in self.genome.get_chroms():
                raise IllegalArgumentError(f'Unknown chromosome: {region}')
            self.chrom = region
            return region, 1, self.genome.get_chrom_size(region)
        uni = re.match(r'^(chr)?([\d]+|[XYM]|(MT)):([\d]+)$', region)
        if uni:
            region += f'-{int(uni.group(4)) + 1}'
        m = re.match(r'^((chr)?([\d]+|[XYM]|(MT))):([\d]+)-([\d]+)$', region)
        if not m:
            raise IllegalArgumentError(f'Invalid genomic region: {region}')
        self.chrom = m.group(1)
        if self.chrom not in self.genome.get_chroms():
            raise IllegalArgumentError(f'Unknown chromosome: {region}')
        return region, int(m.group(5)), int(m.group(6))

    # genomic_region.py:126-161
    def parse_region(self, region):
        self.region_str, region_from, region_to = self.find_region_format(region)
        if region_to <= region_from:
            raise IllegalArgumentError(f'Invalid genomic region: {region}. end before start')
        if region_to > self.genome.get_chrom_size(self.chrom) or region_from < 1:
            raise IllegalArgumentError(f'Invalid genomic region: {region}. Out of range')
        self.bp_tuple = (region_from, region_to)
        self.sites = self._region_str2sites()

    def _region_str2sites(self):
        """`tabix CpG.bed.gz chr:from-to | awk ...` (genomic_region.py:140-161): rows with from <= locus <= to; the
        range is first_index .. last_index(+1 unless the last CpG sits exactly on `to`)."""
        names, sizes = self.genome.get_chrom_cpg_sizes()
        ci = names.index(self.chrom)
        cum = np.concatenate([[0], np.cumsum(sizes)])
        lo, hi = int(cum[ci]), int(cum[ci + 1])               # 0-based [lo, hi) of this chromosome
        loci = self.genome.loci()[lo:hi]
        a, b = self.bp_tuple
        i0 = int(np.searchsorted(loci, a, 'left'))
        i1 = int(np.searchsorted(loci, b, 'right'))            # rows i0 .. i1-1
        if i1 <= i0:
            raise IllegalArgumentError(f'Invalid genomic region: {self.region_str}. No CpGs in range')
        first = lo + i0 + 1
        last = lo + i1                                          # 1-based index of the last row
        end = last + (1 if int(loci[i1 - 1]) < b else 0)
        if first == end:
            raise IllegalArgumentError(f'Invalid genomic region: {self.region_str}. No CpGs in range')
        return self._sites_str_to_tuple(f'{first}-{end}')


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
