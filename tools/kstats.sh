#!/bin/bash
# rocprofv3 --kernel-trace --stats of a python command; prints the per-kernel summary (profiles/summarize_rocpd.py)
# Usage: tools/kstats.sh <out.txt> <script.py> [args...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=$1; shift
rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats -d /tmp/kst -o k -- python3 "$@" > /tmp/kst.log 2>&1
db=$(find /tmp/kst -name "*.db" | head -1)
python3 profiles/summarize_rocpd.py "$db" > "$out"
grep '^{"metric"' /tmp/kst.log | tail -1 >> "$out"     # (the bench line of the profiled run, when the command was bench.py)
tail -3 /tmp/kst.log | grep -v "^{" >> "$out"
cat "$out"
