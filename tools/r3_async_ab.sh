#!/bin/bash
# GPU box: the barrier-free one-workgroup factorisation (opt-in) against the barrier kernel: tests, kernel time (tools/k3_bench.py
# under rocprofv3 --kernel-trace --stats), K3 chain of the synthetic frames.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_gpu_ongpis.py -q -x 2>&1 | tail -2
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
