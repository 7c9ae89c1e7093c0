"""Sanitizer builds of the host-side native code (the reference has none; its `segmentor` has a real UB at max_bp == 0):
csrc/stitch.h + csrc/add_loci.h (chunk grid, junction stitching on the thread pool, BED rows) and the oracle's C restatement of the
chunk DP, each compiled plain, with AddressSanitizer + UndefinedBehaviorSanitizer, and with ThreadSanitizer, and run on
deterministic toy inputs: every build must finish clean and print the same lines."""
import os.path as op
import shutil
import subprocess

import pytest

ROOT = op.dirname(op.dirname(op.abspath(__file__)))
NATIVE = op.join(ROOT, 'tests', 'native')
FLAVOURS = {'plain': [], 'asan_ubsan': ['-fsanitize=address,undefined', '-fno-sanitize-recover=all'], 'tsan': ['-fsanitize=thread']}


def _have(flags):
    if not shutil.which('g++'):
        return False
    r = subprocess.run(['g++', '-x', 'c++', '-', '-o', '/dev/null'] + flags, input='int main(){return 0;}', text=True, capture_output=True)
    return r.returncode == 0


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)
    assert r.returncode == 0, '%s\n%s\n%s' % (' '.join(cmd), r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


@pytest.mark.skipif(not _have(FLAVOURS['asan_ubsan']) or not _have(FLAVOURS['tsan']), reason='no g++ with sanitizer runtimes')
def test_stitching_and_bed_rows_under_sanitizers(tmp_path):
    outs = {}
    for name, flags in FLAVOURS.items():
        exe = str(tmp_path / ('san_host_' + name))
        _run(['g++', '-std=c++17', '-O1', '-g', '-pthread', '-I', op.join(ROOT, 'wgbs_tools_amd', 'csrc'), op.join(NATIVE, 'san_host.cpp'), '-o', exe] + flags)
        for threads in ('1', '4'):
            bed = str(tmp_path / ('%s_%s.bed' % (name, threads)))
            text = _run([exe, threads, bed], env={'TSAN_OPTIONS': 'halt_on_error=1', 'ASAN_OPTIONS': 'detect_leaks=1', 'PATH': '/usr/bin:/bin'})
            outs[(name, threads)] = (text, open(bed, 'rb').read())
    ref = outs[('plain', '1')]
    assert 'checksum' in ref[0] and ref[0].count('world') == 15 and 'rc 0' in ref[0] and len(ref[1]) > 5000
    assert 'sitetable: ' in ref[0] and 'mismatches 0, concurrent misses 0' in ref[0]
    assert 'parse_blocks: rc 0 rows 40000 na 413' in ref[0] and 'parse_blocks on a float field: rc 1' in ref[0]
    assert ref[0].count('write_table pass') == 2 and 'DIFFERS' not in ref[0] and 'write_bedgraph: rc 0' in ref[0]
    assert 'parse_bed: rc 0 rows 50000 width 6 unknown 237' in ref[0] and 'parse_bed on a float column: rc 1' in ref[0] and 'write_annotated_bed: rc 0' in ref[0]
    for key, val in outs.items():
        assert val == ref, key
    # with and without speculation the same world gives the same borders
    lines = [l for l in ref[0].splitlines() if 'fuzzy=40' in l]
    assert len({l.split('checksum')[1] for l in lines}) == 1


@pytest.mark.skipif(not _have(FLAVOURS['asan_ubsan']) or not _have(FLAVOURS['tsan']), reason='no gcc with sanitizer runtimes')
def test_oracle_restatement_under_sanitizers(tmp_path):
    outs = {}
    for name, flags in FLAVOURS.items():
        exe = str(tmp_path / ('san_oracle_' + name))
        _run(['gcc', '-std=c99', '-O1', '-g', '-ffp-contract=off', '-pthread', op.join(NATIVE, 'san_oracle.c'), op.join(ROOT, 'oracle', 'segment_oracle.c'),
              op.join(ROOT, 'oracle', 'libm_probe.c'), '-o', exe, '-lm'] + flags)
        outs[name] = _run([exe], env={'TSAN_OPTIONS': 'halt_on_error=1', 'PATH': '/usr/bin:/bin'})
    assert 'DIFFERENT' not in outs['plain'] and outs['plain'].count('identical') == 7 and 'bad data: rc' in outs['plain']
    assert outs['asan_ubsan'] == outs['plain'] and outs['tsan'] == outs['plain']
