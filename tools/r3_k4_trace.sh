#!/bin/bash
# GPU box: instrumented build; cycle stamps of sampled K4 workgroups for a small-cluster run (stage boundaries per wave)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
touch gpismap_amd/csrc/ongpis_test.hip
make -C gpismap_amd/csrc EXTRA=-DGPIS_INSTRUMENT > /tmp/mk.log 2>&1 || { tail -20 /tmp/mk.log; exit 1; }
for N in ${1:-60} ${2:-240}; do
  python3 tools/k4_bench.py $N 64 8192 1 2>&1 | tail -1
  echo "--- N=$N: stamps relative to the workgroup's first (start, staged, table, first chunk, products done, end)"
  head -24 gpurun_out/k4_trace.txt
done
