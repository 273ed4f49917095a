#!/bin/bash
# GPU box: the barrier-free one-workgroup factorisation against the barrier kernel: tests, kernel time (tools/k3_bench.py
# under rocprofv3 --kernel-trace --stats), K3 chain of the synthetic frames.  Usage: tools/r3_async_ab.sh ["<EXTRA flags>" ...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for extra in "${@:-}"; do
  touch gpismap_amd/csrc/ongpis_train.hip
  make -C gpismap_amd/csrc EXTRA="$extra" > /tmp/mk.log 2>&1 || { tail -20 /tmp/mk.log; exit 1; }
  echo "##### EXTRA=$extra"
  GPIS_ASYNC_CHOL=1 timeout 700 python -m pytest tests/test_gpu_ongpis.py tests/test_gpu_golden.py -q -x 2>&1 | tail -1
  for a in 0 1; do
    echo "== GPIS_ASYNC_CHOL=$a"
    for cfg in "200 512" "350 512"; do
      rm -rf /tmp/k3b
      GPIS_ASYNC_CHOL=$a rocprofv3 --kernel-trace --stats -d /tmp/k3b -o k -- python3 tools/k3_bench.py $cfg > /tmp/k3b.log 2>&1
      grep "^N=" /tmp/k3b.log
      db=$(find /tmp/k3b -name "*.db" | head -1)
      python3 profiles/summarize_rocpd.py "$db" | grep -E "chol_async|chol_kernel" | cut -c1-150
    done
    GPIS_ASYNC_CHOL=$a timeout 120 python3 tools/update_profile.py 8 2>/dev/null | grep "^frame [1234567]" | sed "s/| pts.*//; s/.*K3 device/K3/; s/)//" | tr "\n" " "; echo
  done
done
