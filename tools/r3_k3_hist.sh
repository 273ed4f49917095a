#!/bin/bash
# GPU box: instrumented build; size histogram of the training groups per frame + F1 test
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_golden.py -q -x 2>&1 | tail -3
touch gpismap_amd/csrc/ongpis_store.cpp gpismap_amd/csrc/gpismap3.cpp
make -C gpismap_amd/csrc EXTRA=-DGPIS_INSTRUMENT > /tmp/mk.log 2>&1 || { tail -20 /tmp/mk.log; exit 1; }
python3 tools/update_profile.py 5 2>&1 | grep "^frame\|\[train\]" > gpurun_out/k3_hist.txt
cat gpurun_out/k3_hist.txt
