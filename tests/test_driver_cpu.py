"""Host-side driver parity on the CPU: chunk grid, junction stitching, min_cpg filter, stderr text and BED bytes of
wgbs_tools_amd/segment.py against golden vectors captured from the reference's own Python driver
(tests/golden/driver_cases.json, made by tests/golden/make_golden_driver.py with the reference binary as the chunk
engine).  The chunk engine injected here is the ORACLE (test infrastructure) — the product's engine is the HIP
library, exercised by the -m gpu tests with the same golden vectors."""
import argparse
import contextlib
import hashlib
import io
import json
import os
import os.path as op

import numpy as np
import pytest

from oracle import oracle
from wgbs_tools_amd import genome as G
from wgbs_tools_amd import segment as S
from wgbs_tools_amd import synth

ROOT = op.dirname(op.dirname(op.abspath(__file__)))


class OracleEngine:
    """Test-only chunk engine with HipEngine's interface, backed by oracle/segment_oracle.c."""

    def __init__(self, betas, loci):
        self.betas = betas
        self.loci = loci
        self.calls = []

    def segment_many(self, sites_list, params):
        out = [None] * len(sites_list)
        idx, st0, ln = [], [], []
        for i, (start, end) in enumerate(sites_list):
            assert end - start > 0
            self.calls.append((int(start), int(end)))
            if end - start == 1:
                out[i] = np.array([start, end])
            else:
                idx.append(i); st0.append(start - 1); ln.append(end - start)
        if idx:
            res = oracle.segment_chunks(self.betas, self.loci, st0, ln, params['pcount'], params['max_cpg'],
                                        params['max_bp'], threads=os.cpu_count() or 1)
            for i, r in zip(idx, res):
                out[i] = r.astype(np.int64) + sites_list[i][0]
        return out


@pytest.fixture(scope='module')
def driver_golden():
    with open(op.join(ROOT, 'tests', 'golden', 'driver_cases.json')) as f:
        return json.load(f)


@pytest.fixture(scope='module')
def synth_world(driver_golden, tmp_path_factory):
    meta = driver_golden['meta']
    names = [c for c, _ in meta['chrom_sizes']]
    sizes = [s for _, s in meta['chrom_sizes']]
    loci = synth.synth_loci(meta['seed'], sizes)
    total = int(sum(sizes))
    betas = [synth.synth_betas(meta['seed'], s, 0, total) for s in range(meta['n_betas'])]
    d = tmp_path_factory.mktemp('world')
    refdir = synth.write_genome(str(d / 'references' / 'synth'), names, sizes, loci)
    paths = []
    for i, b in enumerate(betas):
        p = str(d / ('s%d.beta' % i))
        synth.write_beta(p, b)
        paths.append(p)
    return dict(refdir=refdir, paths=paths, betas=betas, loci=loci, names=names, sizes=sizes, dir=str(d))


def make_args(world, out_path, **kw):
    d = dict(sites=None, region=None, array_id=None, bed_file=None, genome=world['refdir'], betas=world['paths'],
             beta_file=None, chunk_size=60000, pcount=15, min_cpg=1, max_cpg=1000, max_bp=2000, out_path=out_path,
             threads=1, device=0)
    d.update(kw)
    return argparse.Namespace(**d)


CASES = ['wg_c20000', 'wg_c60000_min3', 'sites_3chunks', 'sites_single', 'wg_pcount0', 'small_chunks', 'tiny_chunks',
         'wide_bp', 'bed_regions']


@pytest.mark.parametrize('name', CASES)
def test_driver_matches_reference_driver(name, driver_golden, synth_world, tmp_path):
    g = driver_golden['cases'][name]
    kw = dict(g['args'])
    out_path = str(tmp_path / 'out.bed')
    if g['bed_rows'] is not None:
        bed = str(tmp_path / 'regions.bed')
        with open(bed, 'w') as f:
            f.write('#chr\tstart\tend\tstartCpG\tendCpG\n')
            for s, e in g['bed_rows']:
                f.write('chrN\t0\t1\t%d\t%d\n' % (s, e))
        kw['bed_file'] = bed
    args = make_args(synth_world, out_path, **kw)
    eng = OracleEngine(synth_world['betas'], synth_world['loci'])
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        sbc = S.SegmentByChunks(args, synth_world['paths'], engine=eng)
        tags, starts, ends = sbc.break_to_chunks()
        sbc.run()
    # chunk grid (segment.py:124-135)
    assert tags == g['chunks']['tags'] and starts == g['chunks']['starts'] and ends == g['chunks']['ends']
    # the patches asked for are the reference's (as a set: we batch per round and cache repeats)
    nch = len(starts)
    assert set(eng.calls[nch:]) == set(tuple(c) for c in g['patch_calls'])
    # stderr text (chunk-size warning + summary) verbatim; the golden capture also called break_to_chunks twice
    assert err.getvalue() == g['stderr']
    # output table
    rows = [l.rstrip('\n').split('\t') for l in open(out_path)]
    assert all(len(r) == 5 for r in rows)
    table = np.array([[int(r[3]), int(r[4])] for r in rows], dtype=np.int64).reshape(-1, 2)
    assert table.shape[0] == g['n_blocks']
    assert hashlib.sha1(table.tobytes()).hexdigest() == g['table_sha1']
    if g['start_cpg'] is not None:
        assert table[:, 0].tolist() == g['start_cpg'] and table[:, 1].tolist() == g['end_cpg']
    # BED columns 1-3 follow add_loci.cpp:51-54
    loci = synth_world['loci'].astype(np.int64)
    cum = np.cumsum(synth_world['sizes'])
    for r, (s, e) in list(zip(rows, table))[:: max(1, len(rows) // 500)]:
        assert r[0] == synth_world['names'][int(np.searchsorted(cum, s))]
        assert int(r[1]) == loci[s - 1] and int(r[2]) == loci[e - 2] + 1 and int(r[1]) < int(r[2]) and s < e


def test_stats_report(driver_golden, synth_world, tmp_path):
    """--stats PATH: a JSON report of the run (the reference only has its stderr lines); the stderr text is unchanged by it."""
    import json
    g = driver_golden['cases']['wg_c60000_min3']
    out_path, stats_path = str(tmp_path / 'out.bed'), str(tmp_path / 'run.json')
    args = make_args(synth_world, out_path, stats=stats_path, **dict(g['args']))
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        sbc = S.SegmentByChunks(args, synth_world['paths'], engine=OracleEngine(synth_world['betas'], synth_world['loci']))
        sbc.break_to_chunks()
        sbc.run()
    assert err.getvalue() == g['stderr']
    rep = json.load(open(stats_path))
    n_rows = sum(1 for _ in open(out_path))
    assert rep['tool'] == 'wgbstools segment' and rep['blocks_found'] == g['n_blocks'] == n_rows and rep['blocks_dropped'] >= 0
    assert rep['chunks'] == len(g['chunks']['starts']) and rep['sites'] == sum(e - s for s, e in zip(g['chunks']['starts'], g['chunks']['ends']))
    assert rep['parameters']['min_cpg'] == 3 and rep['parameters']['betas'] == len(synth_world['paths']) and rep['out_path'] == out_path
    assert rep['wall_s'] > 0 and set(rep['phases_s']) >= {'segmentation (device + stitching)', 'blocks to BED'}


def test_reference_tree_restatement_matches_reference(driver_golden):
    """tests/reftree.py (the suite's checker of the native stitcher) against vectors captured from the reference's own helpers."""
    import reftree as R
    for rec in driver_golden['funcs']:
        b1, b2 = np.array(rec['b1']), np.array(rec['b2'])
        assert R.repeated(b1, b2).astype(int).tolist() == rec['find_dups']
        assert R.shared(b1, b2) == rec['overlap']
        if rec['overlap']:
            assert R.splice(b1, b2).tolist() == rec['merge2']
    for p, m, want in driver_golden['increase_patch']:
        assert R.next_patch(p, m) == want


def test_stitch_failure_raises_like_reference(stitch_lib):
    import reftree as R
    # b2 does not continue b1 (segment.py:202-205)
    with pytest.raises(R.StitchError, match='not supposed to be merged'):
        R.join(np.array([1, 5, 9]), np.array([10, 12]), None)

    class NeverOverlaps:
        def segment_many(self, sites, params):
            return [np.array([s[0], s[1]]) + (0 if (s[0] - 1) % 4 == 0 and s[1] - s[0] <= 4 else 1000000) for s in sites]
    with pytest.raises(R.StitchError, match='Try increasing chunk size'):
        R.join(np.array([1, 5, 9]), np.array([9, 12, 20]), lambda sites: NeverOverlaps().segment_many(sites, {}))


def test_genome_validation_and_args(synth_world, tmp_path):
    bad = str(tmp_path / 'short.beta')
    synth.write_beta(bad, synth_world['betas'][0][:-5])
    args = make_args(synth_world, None, betas=[bad])
    err = io.StringIO()
    with contextlib.redirect_stderr(err), pytest.raises(G.IllegalArgumentError, match='does not match the input beta file'):
        S.SegmentByChunks(args, [bad], engine=object())
    assert 'incomatible with current genome reference' in err.getvalue()          # utils_wgbs.py:299-302 (sic)
    # max_cpg = min(max_cpg, max_bp // 2) must exceed 1 (segment.py:65-66)
    with pytest.raises(AssertionError):
        S.SegmentByChunks(make_args(synth_world, None, max_bp=3), synth_world['paths'], engine=object())
    a = S.parse_args(['--betas', 'a.beta', 'b.beta', '-r', 'chr1:1-100'])
    assert (a.chunk_size, a.pcount, a.min_cpg, a.max_cpg, a.max_bp) == (60000, 15, 1, 1000, 2000)
    with pytest.raises(SystemExit):
        S.parse_args(['--betas', 'a.beta', '-r', 'chr1', '-s', '1-5'])                 # mutually exclusive
    with pytest.raises(G.IllegalArgumentError):
        S.parse_betas_input(argparse.Namespace(betas=['nonexistent.beta'], beta_file=None))
    lst = str(tmp_path / 'list.txt')
    with open(lst, 'w') as f:
        f.write('# comment\n' + synth_world['paths'][0] + '\n\n' + synth_world['paths'][1] + '\n')
    assert S.parse_betas_input(argparse.Namespace(betas=None, beta_file=lst)) == synth_world['paths'][:2]


def test_region_and_sites_parsing(synth_world):
    gen = G.GenomeRefPaths(synth_world['refdir'])
    loci = synth_world['loci']
    n1 = synth_world['sizes'][0]
    assert gen.get_nr_sites() == sum(synth_world['sizes']) and (gen.loci() == loci).all()
    assert gen.index2chrom(1) == 'chr1' and gen.index2chrom(n1) == 'chr1' and gen.index2chrom(n1 + 1) == 'chr2'
    gr = G.GenomicRegion(sites='100-200', genome=gen)
    assert gr.sites == (100, 200) and gr.chrom == 'chr1' and gr.bp_tuple == (int(loci[99]), int(loci[198]) + 1)
    assert G.GenomicRegion(sites='77', genome=gen).sites == (77, 78)
    with pytest.raises(G.IllegalArgumentError):
        G.GenomicRegion(sites=f'{n1 - 3}-{n1 + 5}', genome=gen)                   # crosses chromosomes
    with pytest.raises(G.IllegalArgumentError):
        G.GenomicRegion(sites='0-5', genome=gen)
    # -r: CpGs with from <= locus <= to; a CpG exactly on `to` is excluded (genomic_region.py:144-148)
    a, b = int(loci[500]), int(loci[520])
    assert G.GenomicRegion(region=f'chr1:{a}-{b}', genome=gen).sites == (501, 521)
    assert G.GenomicRegion(region=f'chr1:{a}-{b + 1}', genome=gen).sites == (501, 522)
    assert G.GenomicRegion(region=f'chr1:{a + 1}-{b + 1}', genome=gen).sites == (502, 522)
    off = n1
    a2, b2 = int(loci[off + 10]), int(loci[off + 30])
    assert G.GenomicRegion(region=f'chr2:{a2:,}-{b2 + 5:,}', genome=gen).sites == (off + 11, off + 32)
    with pytest.raises(G.IllegalArgumentError, match='No CpGs in range'):
        G.GenomicRegion(region=f'chr1:{int(loci[10]) + 1}-{int(loci[11]) - 1}', genome=gen)
    with pytest.raises(G.IllegalArgumentError, match='Unknown chromosome'):
        G.GenomicRegion(region='chr9:5-10', genome=gen)
    whole = G.GenomicRegion(region='chr3', genome=gen)
    s3 = synth_world['sizes'][0] + synth_world['sizes'][1]
    assert whole.sites == (s3 + 1, s3 + synth_world['sizes'][2] + 1)


def test_array_id_parsing(synth_world, tmp_path):
    """--array_id cgNNNN -> the CpG index in the genome's ilmn2CpG.tsv.gz (genomic_region.py:212-232): whole-word match like
    `grep -w`, second column, exactly one hit; the reference's messages otherwise."""
    import argparse
    import gzip
    import shutil
    ref = tmp_path / 'references' / 'synth_ilmn'
    shutil.copytree(synth_world['refdir'], ref)
    gen = G.GenomeRefPaths(str(ref))
    with pytest.raises(G.IllegalArgumentError, match='Could not find Illumina map file'):
        G.GenomicRegion(args=argparse.Namespace(genome=str(ref), sites=None, region=None, array_id='cg00000029'), genome=gen)
    with gzip.open(ref / 'ilmn2CpG.tsv.gz', 'wt') as f:
        f.write('cg00000029\t1234\t450K\ncg000000290\t77\t850K\ncg00000108\t5\t450K\ncg00000108\t6\t850K\nch.1.1\t9\t450K\ncg99999999\tNA\t450K\n')
    gen = G.GenomeRefPaths(str(ref))
    assert gen.ilmn2cpg_dict == str(ref / 'ilmn2CpG.tsv.gz')
    ns = lambda i: argparse.Namespace(genome=str(ref), sites=None, region=None, array_id=i)
    gr = G.GenomicRegion(args=ns('cg00000029'), genome=gen)                     # not cg000000290
    assert gr.sites == (1234, 1235) and gr.chrom == 'chr1'
    assert gr.sites == G.GenomicRegion(sites='1234', genome=gen).sites and gr.region_str == G.GenomicRegion(sites='1234', genome=gen).region_str
    for bad in ('ch.1.1', 'cg', 'cgx12', '00000029'):
        with pytest.raises(G.IllegalArgumentError, match='Invalid Illumina array ID'):
            G.GenomicRegion(args=ns(bad), genome=gen)
    for bad in ('cg00000108', 'cg12345678', 'cg99999999'):                      # two hits, none, not a number
        with pytest.raises(G.IllegalArgumentError, match='Failed retrieving locus for site ' + bad):
            G.GenomicRegion(args=ns(bad), genome=gen)


def test_bed_writer_reproduces_reference_fixture(tmp_path):
    """Byte-identity with the reference's own golden BED (tests/data/segment/chr19_100k_500k.blocks.bed) given a
    loci table consistent with it: pins the add_loci formula and the text format."""
    fx = op.join(ROOT, 'tests', 'golden', 'ref_fixtures', 'chr19_100k_500k.blocks.bed')
    raw = open(fx).read()
    rows = [l.split('\t') for l in raw.strip('\n').split('\n')]
    s = np.array([int(r[3]) for r in rows]); e = np.array([int(r[4]) for r in rows])
    st = np.array([int(r[1]) for r in rows]); en = np.array([int(r[2]) for r in rows])
    first = int(s.min()); last = int(e.max())
    loci = np.zeros(last, dtype=np.int64)
    known = np.zeros(last, dtype=bool)
    for a, b, x, y in zip(s, e, st, en):
        for idx, val in ((a - 1, x), (b - 2, y - 1)):
            assert not known[idx] or loci[idx] == val
            loci[idx] = val; known[idx] = True
    k = np.flatnonzero(known)
    loci[first - 1:] = np.interp(np.arange(first - 1, last), k, loci[k]).astype(np.int64)
    loci[k] = loci[k]
    loci[:first - 1] = np.arange(first - 1) + 1

    class Gen:
        def get_chrom_cpg_sizes(self):
            return ['chrA', 'chr19'], np.array([first - 1, last - first + 1 + 10])

        def loci(self):
            return np.concatenate([loci, np.zeros(10, dtype=np.int64)]).astype(np.uint32)
    out = str(tmp_path / 'x.bed')
    G.write_bed(Gen(), s, e, out)
    assert open(out).read() == raw


def test_native_add_loci_rows_and_errors(tmp_path):
    """wgbsseg_add_loci (the library's restatement of add_loci.cpp:22-57 + cpg_dict.cpp:118-131) against the numpy
    statement of the same rules on random blocks (several formatting shards), and the reference's failure cases."""
    from wgbs_tools_amd import _lib
    rng = np.random.default_rng(5)
    sizes = np.array([70000, 1, 50000, 30000])
    names = ['chr1', 'chrTiny', 'chr_with_a_rather_long_name_' + 'x' * 80, 'chrM']
    cum = np.cumsum(sizes)
    n = int(cum[-1])
    loci = np.concatenate([np.cumsum(rng.integers(2, 300, sz)) + 10000 for sz in sizes]).astype(np.uint32)

    class Gen:
        def get_chrom_cpg_sizes(self):
            return names, sizes

        def loci(self):
            return loci
    # blocks inside chromosomes, incl. empty blocks (end == start), chromosome-final blocks and the genome's last site
    s_list, e_list = [], []
    lo = 1
    for c, hi in enumerate(cum):
        b = np.unique(np.concatenate([[lo, hi + 1], rng.integers(lo, hi + 2, 20000)]))
        s_list.append(b[:-1]); e_list.append(b[1:])
        lo = hi + 1
    s = np.concatenate(s_list + [np.array([5, n])]); e = np.concatenate(e_list + [np.array([5, n + 1])])
    out = str(tmp_path / 'a.bed')
    _lib.add_loci(loci, names, cum, s, e, out, threads=7)
    chrom, start, end = G.blocks_to_bed_lines(Gen(), s, e)
    want = ''.join('%s\t%d\t%d\t%d\t%d\n' % t for t in zip(chrom, start.tolist(), end.tolist(), s.tolist(), e.tolist()))
    assert open(out).read() == want
    _lib.add_loci(loci, names, cum, s[:3], e[:3], out, append=True, threads=1)          # append mode, one shard
    assert open(out).read() == want + ''.join(want.splitlines(True)[:3])
    # failures: message text of the reference, rows before the offending one are written
    cases = [((np.array([3, 9, 8]), np.array([5, 8, 9])), '[wt add_loci] line 1: endCpG < startCpG'),
             ((np.array([3, 0]), np.array([5, 4])), '[wt add_loci] line 1: startCpG < 1'),
             ((np.array([69990]), np.array([70005])), '[wt add_loci] line 0: Cross chromosomes'),
             ((np.array([n + 1]), np.array([n + 1])), '[ cpg_dict ] Could not find chromosome for site: %d' % (n + 1)),   # a START on nr_sites + 1: the reference reads past its loci here
             ((np.array([n + 2]), np.array([n + 3])), '[ cpg_dict ] Could not find chromosome for site: %d' % (n + 2))]
    for (bs, be), msg in cases:
        with pytest.raises(_lib.SegmentorError) as ei:
            _lib.add_loci(loci, names, cum, bs, be, out)
        assert ei.value.msg == msg
        with pytest.raises(RuntimeError):
            G.blocks_to_bed_lines(Gen(), bs, be)
    assert open(out).read() == ''                       # last case: nothing before the offending row
    # a failing row deep in a later formatting shard: every row before it is written, none after
    bad = 40001                                         # shards hold 32768 rows
    assert bad < s.size
    s2, e2 = s.copy(), e.copy()
    e2[bad] = s2[bad] - 1
    with pytest.raises(_lib.SegmentorError) as ei:
        _lib.add_loci(loci, names, cum, s2, e2, out, threads=5)
    assert ei.value.msg == '[wt add_loci] line %d: endCpG < startCpG' % bad
    assert open(out).read() == ''.join(want.splitlines(True)[:bad])
    _lib.add_loci(loci, names, cum, np.array([69990]), np.array([70001]), out)      # ends ON the chromosome border: legal
    assert open(out).read() == 'chr1\t%d\t%d\t69990\t70001\n' % (loci[69989], loci[69999] + 1)


def test_cli_dispatch_and_native_required(synth_world, capsys):
    """`wgbstools segment` must fail loudly, not fall back, when no GPU/HIP library can serve it."""
    from wgbs_tools_amd import wgbs_tools, _lib
    assert wgbs_tools.main(['wgbstools', 'view']) == 1
    if _lib.device_count() == 0 if op.isfile(_lib.LIB_PATH) else True:
        with pytest.raises((_lib.SegmentorError, _lib.NativeLibraryError)):
            wgbs_tools.main(['wgbstools', 'segment', '--betas'] + synth_world['paths'] +
                            ['--genome', synth_world['refdir'], '-s', '1-500', '-o', os.devnull])


# ------------------------------------------------------------------------------------------------------------
# the NATIVE chunk grid + stitching (wgbs_tools_amd/csrc/stitch.h, what wgbsseg_segment_regions runs around the GPU
# batches), driven here by the oracle through a callback
# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def stitch_lib():
    import ctypes as C
    import subprocess
    src = op.join(ROOT, 'tests', 'native', 'stitch_host.cpp')
    lib = op.join(ROOT, 'tests', 'native', 'libstitch_host.so')
    hdr = op.join(ROOT, 'wgbs_tools_amd', 'csrc', 'stitch.h')
    if not op.isfile(lib) or op.getmtime(lib) < max(op.getmtime(src), op.getmtime(hdr)):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', src, '-o', lib])
    return C.CDLL(lib)


def native_segment_regions(stitch_lib, engine, params, regions, chunk_size, speculate=0):
    import ctypes as C
    CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64), C.c_int64,
                     C.POINTER(C.c_int64))

    def cb(starts, ends, n, out, cap, off):
        sites = [(starts[i], ends[i]) for i in range(n)]
        res = engine.segment_many(sites, params)
        pos = 0
        for i, r in enumerate(res):
            off[i] = pos
            for v in r.tolist():
                out[pos] = v
                pos += 1
        off[n] = pos
        return 0
    rs = np.array([r[0] for r in regions], dtype=np.int64)
    re_ = np.array([r[1] for r in regions], dtype=np.int64)
    cap = int((re_ - rs).sum()) + len(regions)
    out = np.empty(cap, dtype=np.int32)
    off = np.empty(len(regions) + 1, dtype=np.int64)
    stats = np.zeros(8, dtype=np.int64)
    err = C.create_string_buffer(512)
    rc = stitch_lib.stitch_segment_regions(rs.ctypes.data_as(C.POINTER(C.c_int64)), re_.ctypes.data_as(C.POINTER(C.c_int64)),
                                           len(regions), C.c_int64(chunk_size), CB(cb), out.ctypes.data_as(C.POINTER(C.c_int32)),
                                           C.c_int64(cap), off.ctypes.data_as(C.POINTER(C.c_int64)),
                                           stats.ctypes.data_as(C.POINTER(C.c_int64)), err, 512, int(speculate))
    assert rc == 0, err.value
    return [out[off[r]:off[r + 1]].astype(np.int64) for r in range(len(regions))], stats


@pytest.mark.parametrize('speculate', [0, 1, 1 | (128 << 8), 1 | (4 << 8)])      # (bits 8..: early delivery of the first batch with edges of that many borders)
@pytest.mark.parametrize('name', CASES)
def test_native_stitching_matches_reference_driver(name, speculate, driver_golden, synth_world, stitch_lib):
    g = driver_golden['cases'][name]
    kw = dict(chunk_size=60000, pcount=15, min_cpg=1, max_cpg=1000, max_bp=2000)
    kw.update({k: v for k, v in g['args'].items() if k in kw})
    params = dict(pcount=kw['pcount'], max_cpg=min(kw['max_cpg'], kw['max_bp'] // 2), max_bp=kw['max_bp'])
    regions, seen = [], set()
    for t in g['chunks']['tags']:                       # regions in the driver's order
        if t not in seen:
            seen.add(t)
            a, b = t.split('-')
            regions.append((int(a), int(b)))
    eng = OracleEngine(synth_world['betas'], synth_world['loci'])
    res, stats = native_segment_regions(stitch_lib, eng, params, regions, kw['chunk_size'], speculate)
    nch = len(g['chunks']['starts'])
    assert stats[0] == nch
    assert eng.calls[:nch] == list(zip(g['chunks']['starts'], g['chunks']['ends']))       # chunk grid
    if speculate:       # second-attempt patches are computed ahead of need: a superset, same result
        assert set(eng.calls[nch:]) >= set(tuple(c) for c in g['patch_calls'])
        # ... and the junction rehearsal gathers what is still missing: never more device batches than the plain order
        eng2 = OracleEngine(synth_world['betas'], synth_world['loci'])
        _, stats2 = native_segment_regions(stitch_lib, eng2, params, regions, kw['chunk_size'], False)
        assert stats[2] <= stats2[2], (stats[2], stats2[2])
    else:
        assert set(eng.calls[nch:]) == set(tuple(c) for c in g['patch_calls'])            # same patches, no extras
    s = np.concatenate([r[:-1] for r in res]); e = np.concatenate([r[1:] for r in res])
    order = np.argsort(s, kind='stable'); s, e = s[order], e[order]
    keep = (e - s) > kw['min_cpg'] - 1
    table = np.stack([s[keep], e[keep]], axis=1).astype(np.int64)
    assert table.shape[0] == g['n_blocks']
    assert hashlib.sha1(np.ascontiguousarray(table).tobytes()).hexdigest() == g['table_sha1']
    for r, (a, b) in zip(res, regions):
        assert r[0] == a and r[-1] == b and (np.diff(r) > 0).all()


# ------------------------------------------------------------------------------------------------------------
# randomised worlds: the native chunk grid + stitching against a plain sequential walk of the reference's pairwise
# tree (segment.py:157-165) over stitch_2_dfs, with an adversarial engine — a pure function of (start, end) whose
# borders depend on where the DP started, so that neighbouring results disagree around many junctions and patches have to
# double (and sometimes swallow a whole operand)
# ------------------------------------------------------------------------------------------------------------
class FickleEngine:
    """Borders of a range = its ends + the sites x with h(x, flavour) == 0, flavour = a hash of the range's start:
    two DPs over the same sites agree only where their flavours do.  `agree_from`: beyond that many sites from its own
    start a DP forgets where it started (like the real one does), so that long enough patches do reconcile."""

    def __init__(self, seed, density, agree_from):
        self.seed, self.density, self.agree_from = seed, density, agree_from
        self.calls = []

    def _borders(self, start, end):
        x = np.arange(start + 1, end, dtype=np.int64)
        flav = (start * 2654435761 + self.seed) % 3
        local = (x - start) < self.agree_from
        h = (x * 0x9E3779B1 + np.where(local, flav, 0) * 0x85EBCA6B + self.seed) % (1 << 32)
        inner = x[(h >> 7) % self.density == 0]
        return np.concatenate([[start], inner, [end]]).astype(np.int64)

    def segment_many(self, sites_list, params):
        self.calls += [tuple(s) for s in sites_list]
        return [self._borders(int(a), int(b)) for a, b in sites_list]


def _tree_merge(chunks, params):
    """the reference's pairwise tree (segment.py:157-165,199-252) as restated in tests/reftree.py, patches from params['engine']"""
    import reftree
    return reftree.tree(chunks, lambda sites: params['engine'].segment_many(sites, params), error=G.IllegalArgumentError)


@pytest.mark.parametrize('seed', range(60))
def test_native_stitching_on_random_worlds_with_a_fickle_engine(seed, stitch_lib):
    rng = np.random.default_rng(4242 + seed)
    chunk = int(rng.choice([40, 60, 97, 128, 333, 1000]))
    n_regions = int(rng.integers(1, 5))
    regions, pos = [], 1
    for _ in range(n_regions):
        ln = int(rng.integers(1, 9 * chunk))
        regions.append((pos, pos + ln))
        pos += ln + int(rng.integers(0, 3))
    density = int(rng.choice([2, 3, 7, 20]))
    agree_from = int(rng.choice([5, 30, 80, 170, 400]))
    want, failed = [], None
    try:
        for a, b in regions:
            eng = FickleEngine(seed, density, agree_from)
            bords = list(range(a, b, chunk)) + [b]
            chunks = eng.segment_many(list(zip(bords[:-1], bords[1:])), {})
            want.append(_tree_merge(chunks, {'engine': eng}))
    except G.IllegalArgumentError as e:                    # the reference gives up (patch grew past an operand)
        failed = str(e)
    # speculate: 0 / 1, and 1 with EARLY DELIVERY of the first batch (round 6: the stitcher's first rehearsal plays on the edges of the
    # chunks' lists — 2, 16 or 128 borders each — while the shim keeps the lists themselves poisoned until finish() is called)
    for speculate in (0, 1, 1 | (2 << 8), 1 | (16 << 8), 1 | (128 << 8)):
        eng = FickleEngine(seed, density, agree_from)
        if failed is not None:
            with pytest.raises(AssertionError, match='Patch stitching Failed'):
                native_segment_regions(stitch_lib, eng, {}, regions, chunk, speculate)
            continue
        got, _ = native_segment_regions(stitch_lib, eng, {}, regions, chunk, speculate)
        for g, w, (a, b) in zip(got, want, regions):
            assert g.tolist() == w.tolist(), (seed, speculate, a, b)


@pytest.mark.parametrize('speculate', [0, 1])
def test_native_stitching_gives_up_like_the_reference(speculate, stitch_lib):
    """A patch that never overlaps its operands grows until it would pass one of them: segment.py:229-232."""
    class NeverOverlaps:
        def segment_many(self, sites, params):
            # chunk results (they start on the grid) are their own two ends; a patch answers with borders from nowhere
            return [np.array([a, b], dtype=np.int64) + (0 if (a - 1) % 100 == 0 and b - a <= 100 else 1000000) for a, b in sites]
    with pytest.raises(AssertionError, match='Try increasing chunk size'):
        native_segment_regions(stitch_lib, NeverOverlaps(), {}, [(1, 451)], 100, speculate)
    with pytest.raises(G.IllegalArgumentError, match='Try increasing chunk size'):
        eng = NeverOverlaps()
        bords = list(range(1, 451, 100)) + [451]
        _tree_merge(eng.segment_many(list(zip(bords[:-1], bords[1:])), {}), {'engine': eng})


def test_blocks_file_loader_of_dash_L_both_parsers(tmp_path, monkeypatch):
    """segment.py:94-103 (-L): the library's one-pass parser for plain, complete tables and the line-by-line parser for the rest
    give the same rows and the same errors."""
    from wgbs_tools_amd import segment as S
    cases_ok = {'plain': 'chr1\t10\t20\t1\t3\nchr1\t20\t40\t3\t7\n', 'header': 'chr\tstart\tend\tstartCpG\tendCpG\nchr1\t10\t20\t1\t3\n',
                'na': 'chr1\t10\t20\t1\t3\nchr1\t20\t40\tNA\tNA\nchr1\t50\t60\t9\t12\n', 'empty_field': 'chr1\t10\t20\t1\t3\nchr1\t20\t40\t\t\n',
                'comments': '# c\nchr1\t10\t20\t1\t3\n\nchr1\t20\t40\t3\t7', 'bad_order': 'chr1\t10\t20\t5\t3\n', 'short': 'chr1\t10\t20\n',
                'nan': 'chr1\t1\t2\tNaN\t4\n', 'float': 'chr1\t1\t2\t3.0\t4\n'}
    for name, text in cases_ok.items():
        p = tmp_path / (name + '.bed')
        p.write_text(text)
        res = []
        for py in (False, True):
            if py:
                monkeypatch.setenv('WGBSSEG_PY_TABLES', '1')
            try:
                res.append(('ok', S.load_blocks_file(str(p)).tolist()))
            except Exception as e:
                res.append((type(e).__name__, str(e)))
            if py:
                monkeypatch.delenv('WGBSSEG_PY_TABLES')
        assert res[0] == res[1], name
        if name == 'plain':
            assert res[0] == ('ok', [[1, 3], [3, 7]])


def test_bed_rows_straight_from_border_lists(tmp_path, capfd):
    """wgbsseg_add_loci_borders (round 4): the BED of a segmentation from its merged border lists (one CSR) == wgbsseg_add_loci over the
    (start, end) pairs numpy makes of the same lists, for every min_cpg, several formatting shards, regions without a block, file /
    append / standard output; the counts it returns are dump_result's; regions out of order and a failing row are refused / reported
    like the array form."""
    from wgbs_tools_amd import _lib
    rng = np.random.default_rng(11)
    sizes = np.array([190000, 1, 90000, 120000])
    names = ['chr1', 'chrTiny', 'chr2', 'chrM']
    cum = np.cumsum(sizes)
    n = int(cum[-1])
    loci = np.concatenate([np.cumsum(rng.integers(2, 300, sz)) + 10000 for sz in sizes]).astype(np.uint32)
    lists, lo = [], 1
    for hi in cum:                                         # one region per chromosome: borders from lo to hi + 1, blocks of 1 .. ~8 sites
        b = np.unique(np.concatenate([[lo, hi + 1], rng.integers(lo, hi + 2, int((hi - lo + 1) // 3) + 1)]))
        lists.append(b.astype(np.int32))
        lo = hi + 1
    lists.insert(2, np.array([int(cum[1]) + 1], dtype=np.int32))      # a region with ONE border: no block
    lists.insert(0, np.zeros(0, dtype=np.int32))                       # and an empty one
    flat = np.concatenate(lists)
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    s = np.concatenate([x[:-1] for x in lists if len(x) > 1]).astype(np.int64)
    e = np.concatenate([x[1:] for x in lists if len(x) > 1]).astype(np.int64)
    assert s.size > 3 * 32768                              # several formatting shards
    for min_cpg in (1, 2, 3, 7):
        keep = (e - s) > min_cpg - 1                       # segment.py:172
        a, b = str(tmp_path / 'a.bed'), str(tmp_path / 'b.bed')
        _lib.add_loci(loci, names, cum, s[keep], e[keep], a, threads=5)
        written, dropped = _lib.add_loci_borders(loci, names, cum, flat, off, min_cpg, b, threads=5)
        assert (written, dropped) == (int(keep.sum()), int((~keep).sum()))
        assert open(a, 'rb').read() == open(b, 'rb').read(), min_cpg
    whole = open(b, 'rb').read()
    w2, d2 = _lib.add_loci_borders(loci, names, cum, flat, off, 7, b, append=True, threads=1)
    assert open(b, 'rb').read() == whole + whole and (w2, d2) == (written, dropped)
    capfd.readouterr()
    _lib.add_loci_borders(loci, names, cum, flat[:off[2]], off[:3], 1, None, threads=2)     # standard output
    assert capfd.readouterr().out.encode() == open(a, 'rb').read()[:0] + ''.join(
        '%s\t%d\t%d\t%d\t%d\n' % (names[0], loci[x - 1], loci[y - 2] + 1, x, y) for x, y in zip(lists[1][:-1].tolist(), lists[1][1:].tolist())).encode()
    # regions out of order: refused (the rows would not come out sorted by startCpG, segment.py:169)
    with pytest.raises(_lib.SegmentorError, match='begins before'):
        _lib.add_loci_borders(loci, names, cum, np.concatenate([lists[3], lists[1]]), np.array([0, len(lists[3]), len(lists[3]) + len(lists[1])]), 1, b)
    with pytest.raises(_lib.SegmentorError, match='region 2 begins before region 0 ends'):       # ... also across a region without borders
        _lib.add_loci_borders(loci, names, cum, np.concatenate([lists[3], lists[1]]), np.array([0, len(lists[3]), len(lists[3]), len(lists[3]) + len(lists[1])]), 1, b)
    # a descending pair INSIDE a region: refused like the array form's "endCpG < startCpG", not counted as a dropped short block
    desc = lists[1].copy()
    desc[[40, 41]] = desc[[41, 40]]
    with pytest.raises(_lib.SegmentorError, match=r'region 0: border 41 \(%d\) follows %d' % (desc[41], desc[40])):
        _lib.add_loci_borders(loci, names, cum, desc, np.array([0, len(desc)]), 1, b)
    # a failing row: the reference's message with the row's position among the WRITTEN rows, rows before it written
    bad = np.array([1, 5, 9, int(cum[0]) - 3, int(cum[0]) + 5], dtype=np.int32)            # the last block crosses chr1 -> chrTiny/chr2
    with pytest.raises(_lib.SegmentorError) as ei:
        _lib.add_loci_borders(loci, names, cum, bad, np.array([0, 5]), 5, b)              # min_cpg 5 drops the first two blocks: the bad row is line 1
    assert ei.value.msg == '[wt add_loci] line 1: Cross chromosomes'
    assert open(b).read().count('\n') == 1


def test_dump_result_csr_reports_before_a_failing_writer(synth_world, tmp_path):
    """dump_result_csr on a list the writer refuses (a block across two chromosomes): the summary line and the --stats counts are still
    produced — the reference prints them before it writes (segment.py:180) — and the error is the writer's."""
    out_path = str(tmp_path / 'out.bed')
    args = make_args(synth_world, out_path, min_cpg=3)
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        sbc = S.SegmentByChunks(args, synth_world['paths'], engine=object())
        sbc.report = {}
        c0 = int(synth_world['sizes'][0])
        # two regions + an empty one between them; blocks of 4, 2 (short), 5 sites | 1 (short), 9, then one that crosses chromosome 1's end
        flat = np.array([1, 5, 7, 12, 20, 21, 30, c0 - 2, c0 + 6], dtype=np.int32)
        off = np.array([0, 4, 4, 9], dtype=np.int64)
        with pytest.raises(RuntimeError, match='Cross chromosomes'):
            sbc.dump_result_csr(flat, off)
    text = err.getvalue()
    assert '[wt segment] found 5 blocks\n             (dropped 2 short blocks)' in text
    assert sbc.report == {'blocks_found': 5, 'blocks_dropped': 2}
    assert sum(1 for _ in open(out_path)) == 4          # the rows before the failing one are written, as the reference's streaming loop would have
    # an ARGUMENT the library refuses (descending borders) is no list of blocks: the error, and no made-up summary
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        sbc.report = {}
        with pytest.raises(RuntimeError):
            sbc.dump_result_csr(np.array([1, 9, 5, 12], dtype=np.int32), np.array([0, 4], dtype=np.int64))
    assert 'found' not in err.getvalue() and sbc.report == {}
