#!/bin/bash
# kernel timeline of small test() calls (tools/small_test_timeline.py) -> gpurun_out/test_timeline.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/ttl
rocprofv3 --kernel-trace -d /tmp/ttl -o k -- python3 tools/small_test_timeline.py > gpurun_out/test_profile.txt 2>&1
db=$(find /tmp/ttl -name "*.db" | head -1)
python3 tools/ktimeline.py "$db" > gpurun_out/test_timeline.txt
tail -1 gpurun_out/test_profile.txt; tail -40 gpurun_out/test_timeline.txt | cut -c1-170
