#!/bin/bash
# GPU box: instrumented build (schedule knobs from the environment), update() per-frame times for a few K3 / K3b schedules.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/k3_sweep.txt
: > $OUT
touch gpismap_amd/csrc/ongpis_store.cpp gpismap_amd/csrc/gpismap3.cpp
make -C gpismap_amd/csrc EXTRA=-DGPIS_INSTRUMENT > /tmp/mk.log 2>&1 || { tail -20 /tmp/mk.log; exit 1; }
run() { echo "== $*" >> $OUT; env "$@" python3 tools/update_profile.py 8 2>/dev/null | grep "^frame [234567]" | sed 's/| pts.*//; s/.*K3 device/K3/; s/)//' | tr '\n' ' ' >> $OUT; echo >> $OUT; }
run K3_MAXWG=240
run K3_MAXWG=192
run K3_MAXWG=160
run K3_MAXWG=128
run K3_MAXWG=96
run K3_MAXWG=160 K3_MINNB=40
run K3_MAXWG=128 K3_MINNB=48
run K3_MAXWG=240
cat $OUT
