#!/bin/bash
# GPU box: A/B of K4 build variants (tools/k4_ablate.sh) on the size curve and the bench pass.
mkdir -p gpurun_out
export VARIANTS="${VARIANTS:-r1:-DK4_AVBUF=1 r3:-DK4_AVBUF=3}"
bash tools/k4_ablate.sh build > gpurun_out/k4ab_build.txt 2>&1
OUT=gpurun_out/k4_ab.txt
: > $OUT
for v in $VARIANTS; do
  tag=${v%%:*}
  for N in ${NS:-60 240 600}; do echo -n "$tag: " >> $OUT; GPISMAP_AMD_LIB=$PWD/.ab/libk4_$tag.so python3 tools/k4_bench.py $N 64 8192 3 2>&1 | tail -1 >> $OUT; done
  echo -n "$tag bench: " >> $OUT
  GPISMAP_AMD_LIB=$PWD/.ab/libk4_$tag.so python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --stress 0 --no-host-api 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])" >> $OUT
done
cat $OUT; tail -3 gpurun_out/k4ab_build.txt
