#!/bin/bash
# GPU box: instrumented build (schedule knobs from the environment), update() per-frame times for a few K3 / K3b schedules.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/k3_sweep.txt
: > $OUT
touch gpismap_amd/csrc/ongpis_store.cpp gpismap_amd/csrc/gpismap3.cpp
make -C gpismap_amd/csrc EXTRA=-DGPIS_INSTRUMENT > /tmp/mk.log 2>&1 || { tail -20 /tmp/mk.log; exit 1; }
run() { echo "== $*" >> $OUT; env "$@" python3 tools/update_profile.py 5 2>/dev/null | grep "^frame [1234]" | sed 's/| pts.*//' >> $OUT; }
run K3B_LONGCOL=24
run K3B_LONGCOL=1000
run K3B_LONGCOL=48
run K3B_LONGCOL=24 K3_GMAX=12 K3_GDIV=450
run K3B_LONGCOL=1000 K3_GMAX=12 K3_GDIV=450
run K3B_LONGCOL=1000 K3_MINNB=20
cat $OUT
