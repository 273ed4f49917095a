#!/bin/bash
# local A/B: K4 efficiency curve + stress predict per variant library
for n in "$@"; do
  echo "== $n"
  export GPISMAP_AMD_LIB=tools/ab/lib_$n.so
  for N in 60 120 240 300 470 600; do python tools/k4_bench.py $N 64 8192 3 2>&1 | tail -1; done
  python tools/stress_bench.py 50000 2>&1 | grep -i "predict" | tail -2
done
