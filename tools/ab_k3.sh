#!/bin/bash
# local A/B: tools/ab_k3.sh <lib names...>  (K3 micro-benchmark per variant library under tools/ab/)
for n in "$@"; do
  echo "== $n"
  for cfg in "350 8" "350 256" "200 256" "550 200" "700 64"; do
    GPISMAP_AMD_LIB=tools/ab/lib_$n.so python tools/k3_bench.py $cfg 2>&1 | tail -1
  done
done
