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


def test_one_gpu_line():
    r = bench.device_report('one', 1, [('box', 0)])
    assert r == {'n_gpus': 1, 'gpus_requested': 1, 'shares': 1, 'distinct_devices': 1, 'oversubscribed': False, 'sharding': 'one GPU'}
