#!/bin/bash
# GPU box: kernel timeline of GPisMap3.update() on the synthetic frames (rocprofv3 kernel trace -> tools/ktimeline.py).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/utl
rocprofv3 --kernel-trace -d /tmp/utl -o k -- python3 tools/update_profile.py ${1:-5} > gpurun_out/update_profile.txt 2>&1
db=$(find /tmp/utl -name "*.db" | head -1)
python3 tools/ktimeline.py "$db" > gpurun_out/update_timeline.txt
grep "^frame" gpurun_out/update_profile.txt
wc -l gpurun_out/update_timeline.txt
