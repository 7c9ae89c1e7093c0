"""bench.py's N > 1 line states the devices it really used (missing 2): `n_gpus` counts DISTINCT physical devices, the
number of shares / ranks and the number asked for are separate fields.  Pure formatter: no GPU, no torch."""
import importlib.util
import os.path as op

ROOT = op.dirname(op.dirname(op.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench_module', op.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_eight_shares_wrapped_onto_one_gpu_are_one_gpu():
    ndev, gpus = 1, 8
    devices = [d % ndev for d in range(gpus)]            # what bench.py's group mode does on a 1-GPU box
    r = bench.device_report('group', gpus, [('box', d) for d in devices])
    assert r['n_gpus'] == 1 and r['shares'] == 8 and r['gpus_requested'] == 8 and r['distinct_devices'] == 1
    assert r['oversubscribed'] is True
    assert 'OVERSUBSCRIBED' in r['sharding'] and '8 shares on 1 distinct GPU' in r['sharding']
    assert 'NOT a 8-GPU measurement' in r['sharding']


def test_a_real_eight_gpu_group_says_eight():
    r = bench.device_report('group', 8, [('node', d) for d in range(8)])
    assert r['n_gpus'] == 8 and r['shares'] == 8 and r['oversubscribed'] is False
    assert 'OVERSUBSCRIBED' not in r['sharding'] and '8 shares on 8 distinct GPUs' in r['sharding']


def test_ranks_sharing_devices_are_counted_once():
    # two ranks of a torchrun on a 1-GPU box (gloo oversubscription mode), and four ranks on a 2-GPU box
    r = bench.device_report('ranks', 2, [('box', 0), ('box', 0)])
    assert r['n_gpus'] == 1 and r['shares'] == 2 and r['oversubscribed'] is True and 'OVERSUBSCRIBED' in r['sharding']
    r = bench.device_report('ranks', 4, [('box', 0), ('box', 1), ('box', 0), ('box', 1)])
    assert r['n_gpus'] == 2 and r['shares'] == 4 and r['oversubscribed'] is True
    # the same device index on two hosts is two devices
    r = bench.device_report('ranks', 2, [('a', 0), ('b', 0)])
    assert r['n_gpus'] == 2 and r['oversubscribed'] is False


def test_a_launch_with_fewer_gpus_than_shares_is_refused_unless_it_opts_in():
    # `bench.py --gpus 8` on a 1-GPU box, two ranks on one device: refused, loudly; --oversubscribe turns them into dry runs; real launches pass
    msg = bench.refuse_oversubscription('group', 8, 8, 1, False)
    assert msg and '--oversubscribe' in msg and 'not be a 8-GPU measurement' in msg and 'shows 1 GPU' in msg
    assert bench.refuse_oversubscription('ranks', 2, 2, 1, False)
    assert bench.refuse_oversubscription('group', 8, 8, 1, True) is None
    assert bench.refuse_oversubscription('group', 8, 8, 8, False) is None
    assert bench.refuse_oversubscription('ranks', 8, 8, 8, False) is None
    assert bench.refuse_oversubscription('one', 1, 1, 1, False) is None
    assert bench.refuse_oversubscription('ranks', 4, 4, 8, False) is None


def test_bench_refuses_eight_shares_without_gpus(tmp_path):
    """the refusal end to end: this container has no GPU at all, so `bench.py --gpus 8` must stop with exit code 2 and the message on stderr before any device work"""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, op.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    import torch
    if torch.cuda.device_count() >= 8:
        return
    assert res.returncode == 2 and b'--oversubscribe' in res.stderr and res.stdout.strip() == b'', (res.returncode, res.stderr[-500:])


def test_one_gpu_line():
    r = bench.device_report('one', 1, [('box', 0)])
    assert r == {'n_gpus': 1, 'gpus_requested': 1, 'shares': 1, 'distinct_devices': 1, 'oversubscribed': False, 'sharding': 'one GPU'}


def test_synthetic_pat_text_and_bgzf_writer(tmp_path):
    """bench.py's `extras` feed pat2beta a text built with numpy and a BGZF file written by hand: every line must be a pat line the reference's parser accepts
    (checked against the oracle's restatement of stdin2beta.cpp:59-93), and the file must read back through gzip AND through the block-parallel BGZF reader."""
    import gzip
    from oracle import pat2beta_oracle as OP
    from wgbs_tools_amd import pat2beta
    n_sites = 50000
    text, n_chars = bench.synth_pat_text(7, n_sites, 30000)
    lines = text.decode().split('\n')
    assert lines[-1] == '' and len(lines) == 30001
    tok = [l.split('\t') for l in lines[:-1]]
    assert all(len(t) == 4 and t[0] == 'chr1' and 1 <= int(t[1]) <= n_sites and 1 <= len(t[2]) <= 12 and set(t[2]) <= set('CTH.') and 1 <= int(t[3]) <= 40 for t in tok)
    assert [int(t[1]) for t in tok] == sorted(int(t[1]) for t in tok) and sum(len(t[2]) for t in tok) == n_chars
    counts = OP.counts(lines[:-1], 1, n_sites + 1)
    assert counts is not None and counts[:, 1].sum() > 0 and (counts[:, 0] <= counts[:, 1]).all()
    gz = str(tmp_path / 'x.pat.gz')
    bench.write_bgzf(gz, text, level=1)
    assert gzip.open(gz, 'rb').read() == text
    assert b''.join(pat2beta.bgzf_pieces(gz)) == text
    assert b''.join(pat2beta.pat_chunks(gz)) == text
