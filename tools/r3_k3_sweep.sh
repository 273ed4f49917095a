#!/bin/bash
# GPU box: instrumented build (schedule knobs from the environment), update() per-frame times for a few K3 / K3b schedules.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/k3_sweep.txt
: > $OUT
touch gpismap_amd/csrc/ongpis_store.cpp gpismap_amd/csrc/gpismap3.cpp
make -C gpismap_amd/csrc EXTRA=-DGPIS_INSTRUMENT > /tmp/mk.log 2>&1 || { tail -20 /tmp/mk.log; exit 1; }
run() { echo "== $*" >> $OUT; env "$@" python3 tools/update_profile.py ${NF:-16} 2>/dev/null | grep "^frame" | awk -v a=${F0:-8} 'NR>a' | sed 's/| pts.*//; s/.*K3 device/K3/; s/)//' | tr '\n' ' ' >> $OUT; echo >> $OUT; }
F0=1 run K3_GDIV=900
F0=1 run K3_GDIV=1400
F0=1 run K3_GDIV=2000
F0=1 run K3_GDIV=1400 K3_GMAX=4
F0=1 run K3_MINNB=40 K3_GDIV=1400
cat $OUT
