#!/bin/bash
# PMC passes (each in its own rocprofv3 run with --kernel-trace only) over tools/k4_bench.py; summaries to gpurun_out/pmc_<tag>.txt
# Usage: tools/k4_pmc.sh <tag> <N> [models] [queries]
cd "$(dirname "$0")/.."
tag=$1; N=${2:-240}; M=${3:-64}; Q=${4:-8192}
export TMPDIR=/tmp
mkdir -p gpurun_out
pass() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o p -- python3 tools/k4_bench.py $N $M $Q 1 > /tmp/pmc_$name.log 2>&1
  db=$(find /tmp/pmc_$name -name "*.db" | head -1)
  echo "== pass $name: $*" >> gpurun_out/pmc_$tag.txt
  python3 profiles/summarize_pmc.py "$db" | grep -E "eval_kernel|inv_kernel|chol_kernel|^kernel" >> gpurun_out/pmc_$tag.txt
  python3 profiles/summarize_rocpd.py "$db" | grep -E "eval_kernel|inv_kernel|chol_kernel" >> gpurun_out/pmc_$tag.txt
  tail -1 /tmp/pmc_$name.log >> gpurun_out/pmc_$tag.txt
}
: > gpurun_out/pmc_$tag.txt
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
[ -n "$ONLY1" ] && { cat gpurun_out/pmc_$tag.txt; exit 0; }
pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
pass sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
cat gpurun_out/pmc_$tag.txt
