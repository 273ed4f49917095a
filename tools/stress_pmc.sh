#!/bin/bash
# HBM-side traffic of the stress training kernels (BASELINE config 5): separate rocprofv3 --pmc passes, --kernel-trace only.
# Usage (GPU box): tools/stress_pmc.sh [clusters]   -> gpurun_out/${ROUND}_stress_pmc.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROUND=${ROUND:-r06}; export ROUND
mkdir -p gpurun_out
NCL=${1:-50000}
OUT=gpurun_out/${ROUND}_stress_pmc.txt
: > $OUT
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/sp_$tag
  rocprofv3 --pmc $pass --kernel-trace -d /tmp/sp_$tag -o p -- python3 tools/stress_bench.py $NCL > /tmp/sp_$tag.log 2>&1
  db=$(find /tmp/sp_$tag -name "*.db" | head -1)
  echo "== rocprofv3 --pmc $pass --kernel-trace -- python3 tools/stress_bench.py $NCL   (FETCH/WRITE_SIZE in KiB; FETCH x2 on gfx950 for 16-byte streams)" >> $OUT
  python3 profiles/summarize_pmc.py "$db" | grep -E "train_fused|chol_kernel|inv_kernel|buildK|^kernel" >> $OUT
  python3 profiles/summarize_rocpd.py "$db" | grep -E "train_fused|chol_kernel|inv_kernel|buildK" >> $OUT
done
cut -c1-170 $OUT
