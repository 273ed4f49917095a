#!/bin/bash
# local: one-workgroup vs cooperative factorisation of the same clusters (instrumented store: schedule from the environment)
export GPISMAP_AMD_LIB=${ABLIB:-tools/ab/lib_sinstr.so}
for m in 1 8 64 256; do
  echo "== $m clusters, one workgroup each"; K3_MINNB=999 python tools/k3_bench.py 350 $m 2>&1 | tail -1
done
for g in 2 4 8 16; do
  echo "== 1 cluster, G=$g"; K3_GMAX=$g K3_GDIV=1 K3_MINNB=8 python tools/k3_bench.py 350 1 2>&1 | tail -1
done
for g in 2 4 8; do
  echo "== 16 clusters nb 59, G=$g"; K3_GMAX=$g K3_GDIV=1 K3_MINNB=8 python tools/k3_bench.py 550 16 2>&1 | tail -1
done
echo "== 16 clusters nb 59, one workgroup each"; K3_MINNB=999 python tools/k3_bench.py 550 16 2>&1 | tail -1
