#!/bin/bash
# GPU box: K4 variants built locally into tools/ab/ (tools/ab_build.sh <name> "<flags>" ongpis_test.hip): size curve + bench pass each.
mkdir -p gpurun_out
OUT=gpurun_out/k4_ab2.txt
: > $OUT
for lib in tools/ab/lib_*.so; do
  tag=$(basename $lib .so); tag=${tag#lib_}
  for N in ${NS:-120 300 600}; do echo -n "$tag: " >> $OUT; GPISMAP_AMD_LIB=$PWD/$lib timeout 200 python3 tools/k4_bench.py $N 64 8192 3 2>&1 | tail -1 | sed 's/models=64 queries.model=8192: //' >> $OUT; done
  echo -n "$tag bench: " >> $OUT
  GPISMAP_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --stress 0 --no-host-api --update-repeats 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])" >> $OUT
done
cat $OUT
