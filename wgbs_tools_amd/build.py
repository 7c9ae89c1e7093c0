"""Build the native pieces in-tree (hipcc cross-compiles gfx950 without a GPU).

  libwgbsseg.so    the product: HIP kernels + C ABI (include/wgbsseg.h)            <- csrc/wgbsseg.hip
  libwgbssynth.so  synthetic-input generator on the device (bench / tests only)     <- csrc/synth.hip
  segmentor        the reference's per-chunk executable (same argv / stdin / stdout) over libwgbsseg.so   <- csrc/segmentor_main.cpp

The built .so files are git-ignored but travel with gpurun snapshots.
"""
import os
import os.path as op
import shutil
import subprocess
import sys

HERE = op.dirname(op.abspath(__file__))
CSRC = op.join(HERE, 'csrc')
ARCH = 'gfx950'
# -ffp-contract=off is part of the numerical contract (no fused multiply-add anywhere on the scoring path)
HIPFLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-pthread']

TARGETS = {
    'libwgbsseg.so': (['wgbsseg.hip'], ['seg_kernels.h', 'plain_dp.h', 'wave_prims.h', 'exact_log2.h', 'stitch.h', 'add_loci.h', 'table_io.h', '../../include/wgbsseg.h']),
    'libwgbssynth.so': (['synth.hip'], []),
}


def source_hash():
    """sha256 (first 16 hex digits) over the sources of libwgbsseg.so: what a profile under profiles/ is keyed on — a PMC or
    instruction-mix file describes the kernels of exactly one source state (bench.py reports such a file only on a match)."""
    import hashlib
    srcs, hdrs = TARGETS['libwgbsseg.so']
    h = hashlib.sha256()
    for f in sorted(op.normpath(op.join(CSRC, x)) for x in srcs + hdrs):
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and op.isfile(cand):
            return cand
    raise RuntimeError('hipcc not found: cannot build the gfx950 libraries')


# host executables linked against the library next to them (rpath $ORIGIN): name -> (sources, headers)
PROGRAMS = {
    'segmentor': (['segmentor_main.cpp'], ['../../include/wgbsseg.h']),
}


def _stale(out, deps):
    if not op.isfile(out):
        return True
    t = op.getmtime(out)
    return any(op.getmtime(d) > t for d in deps if op.isfile(d))


def build(force=False, verbose=False):
    built = []
    for lib, (srcs, hdrs) in TARGETS.items():
        out = op.join(CSRC, lib)
        srcp = [op.join(CSRC, s) for s in srcs]
        deps = srcp + [op.normpath(op.join(CSRC, h)) for h in hdrs]
        if not all(op.isfile(s) for s in srcp):
            continue
        if force or _stale(out, deps):
            cmd = [hipcc()] + HIPFLAGS + srcp + ['-o', out]
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            subprocess.check_call(cmd, cwd=CSRC)
            built.append(lib)
    for exe, (srcs, hdrs) in PROGRAMS.items():
        out = op.join(CSRC, exe)
        srcp = [op.join(CSRC, s) for s in srcs]
        deps = srcp + [op.normpath(op.join(CSRC, h)) for h in hdrs] + [op.join(CSRC, 'libwgbsseg.so')]
        if force or _stale(out, deps):
            cmd = [hipcc(), '-O2', '-std=c++17'] + srcp + ['-L' + CSRC, '-lwgbsseg', '-Wl,-rpath,$ORIGIN', '-o', out]
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            subprocess.check_call(cmd, cwd=CSRC)
            built.append(exe)
    return built


if __name__ == '__main__':
    print(build(force='-f' in sys.argv, verbose=True))
