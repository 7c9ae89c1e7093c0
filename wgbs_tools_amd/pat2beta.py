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
import os
import os.path as op

from .convert import delete_or_skip
from .genome import GenomeRefPaths, IllegalArgumentError
from .cliutil import add_threads_option, require_file

CHUNK_BYTES = 64 << 20


def splitextgz(path):
    """utils_wgbs.py: the extension of x.pat.gz is .pat.gz"""
    if path.endswith('.gz'):
        a, b = op.splitext(path[:-3])
        return a, b + '.gz'
    return op.splitext(path)


def _bgzf_block_size(buf, pos):
    """Total size of the BGZF block that begins at buf[pos] (the SAM specification, section 4.1: a gzip member whose extra field
    holds the subfield 'BC' with the block's size - 1), or None when the bytes there are not such a header."""
    if len(buf) - pos < 18 or buf[pos:pos + 4] != b'\x1f\x8b\x08\x04':
        return None
    xlen = buf[pos + 10] | (buf[pos + 11] << 8)
    q, end = pos + 12, pos + 12 + xlen
    if end > len(buf):
        return None
    while q + 4 <= end:
        slen = buf[q + 2] | (buf[q + 3] << 8)
        if buf[q] == 66 and buf[q + 1] == 67 and slen == 2 and q + 6 <= end:
            return (buf[q + 4] | (buf[q + 5] << 8)) + 1
        q += 4 + slen
    return None


def _inflate_block(block):
    import zlib
    xlen = block[10] | (block[11] << 8)
    return zlib.decompress(block[12 + xlen:-8], -15)                 # raw deflate between the header and CRC32 + ISIZE


def bgzf_pieces(path, read_bytes=8 << 20, threads=None):
    """The text of a BGZF file (what `bgzip` writes and wgbstools' .pat.gz are: independent gzip blocks of <= 64 KB) in pieces,
    the blocks of every piece inflated on a pool of threads (zlib releases the interpreter lock) — a .pat.gz of a deep sample is
    gigabytes of text behind ONE `gunzip -c` in the reference (pat2beta.py:30).  Yields nothing and returns False when the file
    does not begin with a BGZF header (plain gzip: the caller's gzip.open path)."""
    from concurrent.futures import ThreadPoolExecutor
    with open(path, 'rb') as f:
        buf = f.read(read_bytes)
        if _bgzf_block_size(buf, 0) is None:
            return False
        n_thr = threads or min(32, os.cpu_count() or 1)
        with ThreadPoolExecutor(n_thr) as pool:
            while buf:
                blocks, pos = [], 0
                while True:
                    size = _bgzf_block_size(buf, pos)
                    if size is None or pos + size > len(buf):
                        break
                    blocks.append(buf[pos:pos + size])
                    pos += size
                if not blocks:
                    if len(buf) >= 18 and buf[0] == 0x1f and buf[1] == 0x8b and _bgzf_block_size(buf, 0) is None:
                        # a member that is gzip but not BGZF (`cat a.pat.gz b.pat.gz` of different writers): the reference's
                        # `gunzip -cd` reads such a file, so hand the rest to zlib's multi-member inflater, in bounded pieces
                        import zlib
                        d = zlib.decompressobj(wbits=31)
                        fresh = True                              # `d` has not seen a byte yet (the stream may end between two members)
                        try:
                            while buf:
                                out = d.decompress(buf)
                                fresh = False
                                while d.eof:                      # next member
                                    rest = d.unused_data
                                    d = zlib.decompressobj(wbits=31)
                                    fresh = True
                                    if not rest:
                                        break
                                    out += d.decompress(rest)
                                    fresh = False
                                if out:
                                    yield out
                                buf = f.read(read_bytes)
                        except zlib.error as e:                   # trailing junk, a corrupt member: what the BGZF path and gzip.open refuse as well
                            raise IllegalArgumentError(f'Invalid gzip data in {path}: {e}')
                        if not fresh and not d.eof:               # the last member stops in the middle (gzip.open: EOFError; the BGZF path: "truncated")
                            raise IllegalArgumentError(f'Invalid gzip data in {path}: truncated (the last gzip member is incomplete)')
                        return True
                    more = f.read(read_bytes)
                    if not more or len(buf) > (1 << 20):          # a BGZF block is at most 64 KB: more than that without one is not BGZF
                        raise IllegalArgumentError(f'Invalid gzip data in {path}: truncated or not BGZF after the first block')
                    buf = buf[pos:] + more
                    continue
                yield b''.join(pool.map(_inflate_block, blocks, chunksize=64))
                buf = buf[pos:] + f.read(read_bytes)
    return True


def pat_chunks(pat_path, chunk_bytes=CHUNK_BYTES):
    """the text of a .pat / .pat.gz file in pieces that end on line boundaries"""
    def pieces():
        if pat_path.endswith('.gz') and os.environ.get('WGBSSEG_PY_GUNZIP', '0') in ('', '0'):
            gen = bgzf_pieces(pat_path)
            bgzf = yield from gen
            if bgzf:
                return
        opener = gzip.open if pat_path.endswith('.gz') else open
        with opener(pat_path, 'rb') as f:
            while True:
                buf = f.read(chunk_bytes)
                if not buf:
                    return
                yield buf
    rest = b''
    for buf in pieces():
        buf = rest + buf
        if len(buf) < chunk_bytes and not buf.endswith(b'\n'):      # keep collecting: pieces of a BGZF file are smaller than a chunk
            rest = buf
            continue
        cut = buf.rfind(b'\n') + 1
        rest = buf[cut:]
        if cut:
            yield buf[:cut]
    if rest:
        yield rest if rest.endswith(b'\n') else rest + b'\n'


def pat2beta(pat_path, out_dir, args, force=True):
    """pat2beta.py:17-44 for one file; returns the path written (None when skipped)."""
    require_file(pat_path)
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
    add_threads_option(parser)
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
