#!/bin/bash
# round 5, call 9: the adversarial world at full size against the reference binary (x8 and x32)
set -u
O=$PWD/gpurun_out/r05c9; mkdir -p $O
for n in 8 32; do
  timeout 1200 python tools/full_vs_reference.py --samples $n --adversarial > $O/adversarial_x$n.log 2>&1; echo "adversarial x$n: rc $? $(tail -1 $O/adversarial_x$n.log | cut -c1-330)"
  grep "^\[\|adversarial world\|DIFFERENT" $O/adversarial_x$n.log | cut -c1-260
done
