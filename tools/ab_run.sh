#!/bin/bash
# GPU box: run a command once per variant library in tools/ab/ (built locally by tools/ab_build.sh): tools/ab_run.sh "<cmd>" [grep pattern]
cmd=$1; pat=${2:-.}
for lib in tools/ab/lib_*.so; do
  n=$(basename $lib .so); n=${n#lib_}
  echo "== $n"
  GPISMAP_AMD_LIB=$PWD/$lib timeout 300 $cmd 2>&1 | grep -E "$pat" | tail -4
done
