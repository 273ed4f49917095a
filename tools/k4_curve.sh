#!/bin/bash
# K4 efficiency against cluster size: tools/k4_bench.py for N = 60 ... 600 points per cluster (K ~ 3.4 N), 64 clusters x 8192 queries.
mkdir -p gpurun_out
OUT=gpurun_out/${ROUND:-r06}_k4_curve.txt
: > $OUT
for N in 60 120 240 300 470 600; do python3 tools/k4_bench.py $N 64 8192 3 2>&1 | tail -1 >> $OUT; done
cat $OUT
