# tools/extra_fuzz.py [first_seed] [n_seeds]: the random-parameter parity test of tests/test_gpu_parity.py (test_13) on many more seeds than the suite runs
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import test_gpu_parity as T
from wgbs_tools_amd import _lib
seg = _lib.Segmenter(0)
bad = 0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 160
for seed in range(first, first + count):
    try:
        T.test_13_random_parameters_and_adversarial_inputs_match_oracle(seg, seed)
    except AssertionError as e:
        bad += 1
        print('seed', seed, 'FAILED', str(e)[:300])
print('done, failures:', bad)
