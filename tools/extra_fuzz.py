# tools/extra_fuzz.py [first_seed] [n_seeds] [seconds]: the random-parameter parity test of tests/test_gpu_parity.py (test_13) on many more seeds than the suite runs
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import test_gpu_parity as T
from wgbs_tools_amd import _lib
seg = _lib.Segmenter(0)
bad = 0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 160
import time
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9          # stop cleanly after this many seconds
t0 = time.time()
done = 0
for seed in range(first, first + count):
    if time.time() - t0 > budget:
        break
    done += 1
    try:
        T.test_13_random_parameters_and_adversarial_inputs_match_oracle(seg, seed)
    except AssertionError as e:
        bad += 1
        print('seed', seed, 'FAILED', str(e)[:300], flush=True)
print('done: seeds %d .. %d, failures: %d' % (first, first + done - 1, bad), flush=True)
