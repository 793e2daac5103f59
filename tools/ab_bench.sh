#!/bin/bash
# Same-box A/B of two builds of the library on the headline workload: alternates `python bench.py --no-extras` between the
# shipped library (A) and $1 (B, a path for NST_LIB) $2 times each and prints the closure rates.  Boxes of the pool differ by
# several per cent, and one box drifts by ~1 %: decide on alternated runs of one box only.
B=$1; N=${2:-3}; shift; shift
for i in $(seq 1 $N); do
  for v in A B; do
    if [ $v = A ]; then unset NST_LIB; else export NST_LIB=$B; fi
    python bench.py --gpus 1 --steps 200 --warmup 20 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'],2), 'it/s  closure', round(d['kernel_ms_per_closure']['closure'],3), 'ms  conv', round(d['kernel_ms_per_closure']['conv3x3_mfma'],3), 'ms')"
  done
done
