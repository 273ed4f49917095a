#!/bin/bash
# GPU box: instrumented build; fine-grained host laps of update() ([upd] lines), last frame
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
touch gpismap_amd/csrc/ongpis_store.cpp gpismap_amd/csrc/gpismap3.cpp
make -C gpismap_amd/csrc EXTRA=-DGPIS_INSTRUMENT > /tmp/mk.log 2>&1 || { tail -20 /tmp/mk.log; exit 1; }
GPIS_PIPELINE_UPDATE=${PIPE:-0} python3 tools/update_profile.py 5 2>&1 | grep "^frame\|\[upd\]" > gpurun_out/upd_laps.txt
tail -22 gpurun_out/upd_laps.txt
