#!/bin/bash
# GPU box: counters of the training kernels inside GPisMap3.update() (kernels serialised by the counter collection: the
# durations are each kernel's time ALONE).  -> gpurun_out/update_pmc.txt      Usage: tools/update_pmc.sh [traffic|sq]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/update_pmc.txt
: > $OUT
if [ "${1:-traffic}" = traffic ]; then
  PASSES=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
else
  PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE")
fi
for pass in "${PASSES[@]}"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/up_$tag
  rocprofv3 --pmc $pass --kernel-trace -d /tmp/up_$tag -o p -- python3 tools/update_profile.py 5 > /tmp/up_$tag.log 2>&1
  db=$(find /tmp/up_$tag -name "*.db" | head -1)
  echo "== rocprofv3 --pmc $pass --kernel-trace -- python3 tools/update_profile.py 5   (FETCH/WRITE_SIZE in KiB)" >> $OUT
  python3 profiles/summarize_pmc.py "$db" | grep -E "train_fused|chol_|inv_kernel|buildK|^kernel" >> $OUT
  python3 profiles/summarize_rocpd.py "$db" | grep -E "train_fused|chol_|inv_kernel|buildK" >> $OUT
done
cut -c1-175 $OUT
