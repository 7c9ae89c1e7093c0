# tools/aligned_fuzz.py [first_seed] [n_seeds] [seconds]: like tools/extra_fuzz.py, but the chunk starts, ends and lengths are drawn ON and
# next to the multiples of 16 / 64 / 128 / 256 where the kernels' tiles, units, carry groups and batches begin (the case a uniform draw
# meets once in thousands of chunks: seed 5751 of the uniform fuzz), window limits next to the tile-class boundary (60 / 64 / 128 sites).
# The draws live in tests/fuzzlib.py; a time-boxed slice of the same fuzz runs inside the GPU suite (tests/test_gpu_fuzz.py).
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import fuzzlib
from wgbs_tools_amd import _lib
import oracle.oracle as oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
seg = _lib.Segmenter(0)
done, chunks, bad = fuzzlib.run_aligned(seg, oracle, first, count, budget, os.cpu_count() or 1, log=lambda m: print(m, flush=True))
print('done: seeds %d .. %d, %d chunks, differences: %d' % (first, first + done - 1, chunks, len(bad)), flush=True)
