"""`wgbstools pat2beta` on MI355X: a pat file -> the `.beta` / `.lbeta` file `segment` reads.

Drop-in for the reference's src/python/pat2beta.py (same flags, output names, skip / overwrite rule), written against the
formats:

    pat     tab-separated text (optionally gzip): chr, index of the read's first CpG, pattern over {C, T, H, .}, number of
            reads with that pattern (+ ignored columns).  Every site under a C / T / H gains `count` coverage, under C / H also
            `count` methylated (src/pat2beta/stdin2beta.cpp:59-93).
    .beta   nr_sites x (#meth, #cov) uint8; `-l`: .lbeta, uint16.  A site whose coverage exceeds the type's maximum M is
            stored as (trunc(meth / cov * M), M) (utils_wgbs.py:277-290).

The reference pipes `gunzip -c x.pat.gz` into its stdin2beta binary; here the text is decompressed on the host in chunks and
counted on the GPU (wgbsseg_patbeta_*, include/wgbsseg.h) while the next chunk is being decompressed.  No CPU fallback.
"""
import argparse
import gzip
import os.path as op

from .convert import delete_or_skip
from .genome import GenomeRefPaths, IllegalArgumentError
from .segment import add_multi_thread_args, validate_single_file

CHUNK_BYTES = 64 << 20


def splitextgz(path):
    """utils_wgbs.py: the extension of x.pat.gz is .pat.gz"""
    if path.endswith('.gz'):
        a, b = op.splitext(path[:-3])
        return a, b + '.gz'
    return op.splitext(path)


def pat_chunks(pat_path, chunk_bytes=CHUNK_BYTES):
    """the text of a .pat / .pat.gz file in pieces that end on line boundaries"""
    opener = gzip.open if pat_path.endswith('.gz') else open
    rest = b''
    with opener(pat_path, 'rb') as f:
        while True:
            buf = f.read(chunk_bytes)
            if not buf:
                break
            buf = rest + buf
            cut = buf.rfind(b'\n') + 1
            rest = buf[cut:]
            if cut:
                yield buf[:cut]
    if rest:
        yield rest + b'\n'


def pat2beta(pat_path, out_dir, args, force=True):
    """pat2beta.py:17-44 for one file; returns the path written (None when skipped)."""
    validate_single_file(pat_path)
    if not (pat_path.endswith('.pat.gz') or pat_path.endswith('.pat')):
        raise IllegalArgumentError(f'Invalid pat suffix: {pat_path}')
    suff = '.lbeta' if args.lbeta else '.beta'
    out_beta = op.join(out_dir, splitextgz(op.basename(pat_path))[0] + suff)
    if not delete_or_skip(out_beta, force):
        return None
    from . import _lib
    nr_sites = GenomeRefPaths(args.genome).get_nr_sites()
    with _lib.PatBeta(1, nr_sites + 1, device=getattr(args, 'device', 0)) as pb:
        for chunk in pat_chunks(pat_path):
            pb.feed(chunk)
        try:
            rows = pb.finish(lbeta=args.lbeta)
        except _lib.SegmentorError as e:
            raise IllegalArgumentError(e.msg)
    rows.tofile(out_beta)
    return out_beta


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description=main.__doc__)
    parser.add_argument('pat_paths', help='pat[.gz] files', nargs='+')
    parser.add_argument('-f', '--force', action='store_true', help='Overwrite existing file if existed')
    parser.add_argument('-o', '--out_dir', help='Output directory for the beta file. [.]', default='.')
    parser.add_argument('-l', '--lbeta', action='store_true', help='Use lbeta file (uint16) instead of beta (uint8)')
    parser.add_argument('--genome', help='Genome reference name.')
    add_multi_thread_args(parser)
    parser.add_argument('--device', type=int, default=0, help='HIP device index [0]')
    return parser.parse_args(argv)


def main(argv=None):
    """
    Generate a beta file from a pat file
    """
    args = parse_args(argv)
    for pat in args.pat_paths:
        pat2beta(pat, args.out_dir, args, args.force)


if __name__ == '__main__':
    main()
