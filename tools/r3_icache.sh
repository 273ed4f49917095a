#!/bin/bash
# GPU box: instruction-cache counters of the factorisation kernels (tools/k3_bench.py)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/k3_icache.txt
: > $OUT
for a in 0 1; do
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/ic
  GPIS_ASYNC_CHOL=$a rocprofv3 --pmc $pass --kernel-trace -d /tmp/ic -o p -- python3 tools/k3_bench.py 350 512 > /tmp/ic.log 2>&1
  db=$(find /tmp/ic -name "*.db" | head -1)
  echo "== GPIS_ASYNC_CHOL=$a rocprofv3 --pmc $pass" >> $OUT
  python3 profiles/summarize_pmc.py "$db" | grep -E "chol_|inv_kernel|^kernel" >> $OUT
  python3 profiles/summarize_rocpd.py "$db" | grep -E "chol_|inv_kernel" >> $OUT
  tail -2 /tmp/ic.log | grep -v "^N=" | head -2 >> $OUT
done
done
cut -c1-170 $OUT
