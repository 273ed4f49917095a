#!/bin/bash
# GPU box: the round's profile evidence, ONE command: the default bench line, its kernel stats (checked against the line), K4
# traffic / counters, stress kernel stats and counters, the update() counters / timeline / pipelined timing, the K4 efficiency
# curve, the bundled sequences.  Results under gpurun_out/ (copy into profiles/).  Usage: ROUND=r06 bash tools/round_profiles.sh [quick]
# EXIT CODE 3 when the kernel-stats file contradicts the bench line it belongs to (sum of K4 per pass > ms_per_step).
mkdir -p gpurun_out
export TMPDIR=/tmp
export ROUND=${ROUND:-r06}
R=$ROUND
# FIRST, before any profiler run: the pipelined update needs its kernels to overlap, and for a while after a rocprofv3 counter
# collection (which serialises kernels) the device still runs them one at a time -- frames of 40-60 ms that are not the library's
python3 tools/update_pipeline.py 8 2>&1 | tail -5 > gpurun_out/${R}_update_pipeline.txt
# The kernel-stats collection runs the SAME bench command with one update fusion and the size-class launches of small test() passes
# on one stream: with several fusions and the forked side streams the process holds more HIP streams than hardware queues, and the
# profiler's per-dispatch intervals of kernels on oversubscribed queues include the time they sat behind each other (the r04 file:
# 2175 ms of K4 "per pass" inside a 1203 ms step).  The check below refuses such a file.
GPIS_K4_SERIAL=1 bash tools/kstats.sh gpurun_out/${R}_bench_kernel_stats.txt bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-api --update-repeats 1 --stress 0 > /dev/null 2>&1
python3 - <<'PY'
import json, os, re, sys
R = os.environ["ROUND"]
p = "gpurun_out/%s_bench_kernel_stats.txt" % R
txt = open(p).read()
k4 = sum(int(m.group(1)) for m in re.finditer(r"ongpis_eval_kernelILi8\S*\s+\d+\s+(\d+)", txt))
line = None
for l in txt.splitlines():
    if l.startswith("{") and '"ms_per_step"' in l:
        line = json.loads(l)
if line is None:
    print("round_profiles: no bench line at the end of %s" % p); sys.exit(3)
passes = line["steps"] + line["warmup"]
k4_pass = k4 / 1e6 / passes
head = "# K4<8,...> per 256^3 pass from this trace: %.1f ms (sum / %d passes); ms_per_step of the SAME run: %.1f; collected with GPIS_K4_SERIAL=1 --update-repeats 1\n" % (k4_pass, passes, line["ms_per_step"])
open(p, "w").write(head + txt)
print(head.strip())
if k4_pass > 1.02 * line["ms_per_step"]:
    print("round_profiles: INCONSISTENT kernel stats (K4 per pass exceeds the step): do not commit this file"); sys.exit(3)
PY
rc=$?
if [ $rc != 0 ]; then echo "kernel-stats check failed ($rc)"; [ "$1" = quick ] || exit 3; fi
[ "$1" = quick ] && exit $rc
bash tools/kstats.sh gpurun_out/${R}_stress_kstats.txt tools/stress_bench.py 50000 > /dev/null 2>&1
bash tools/stress_pmc.sh 50000 > /dev/null 2>&1
bash tools/measure_traffic.sh > /dev/null 2>&1
bash tools/update_pmc.sh traffic > /dev/null 2>&1; cp gpurun_out/update_pmc.txt gpurun_out/${R}_update_pmc.txt
bash tools/update_pmc.sh sq > /dev/null 2>&1; cat gpurun_out/update_pmc.txt >> gpurun_out/${R}_update_pmc.txt
bash tools/update_timeline.sh 5 > /dev/null 2>&1; { cat gpurun_out/update_profile.txt; grep -E "ongpis|obsgp|fused" gpurun_out/update_timeline.txt; } > gpurun_out/${R}_update_timeline.txt
bash tools/k4_curve.sh > /dev/null 2>&1
# K4 cycle stamps (instrumented build, tools/ab/lib_instr.so = `bash tools/ab_full.sh instr -DGPIS_INSTRUMENT` of the same source): bench pass, then two small sizes
if [ -f tools/ab/lib_instr.so ]; then
  rm -f gpurun_out/k4_trace.txt
  GPISMAP_AMD_LIB=$PWD/tools/ab/lib_instr.so python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-host-api --update-repeats 1 --stress 0 > /dev/null 2>&1
  python3 tools/k4_trace_summary.py gpurun_out/k4_trace.txt 1000 > gpurun_out/${R}_k4_stamps.txt 2>&1
  for N in 60 120 300; do
    rm -f gpurun_out/k4_trace.txt
    GPISMAP_AMD_LIB=$PWD/tools/ab/lib_instr.so python3 tools/k4_bench.py $N 64 8192 1 > /dev/null 2>&1
    echo "== k4_bench N=$N (64 clusters x 8192 queries)" >> gpurun_out/${R}_k4_stamps.txt
    python3 tools/k4_trace_summary.py gpurun_out/k4_trace.txt 1000 | tail -1 >> gpurun_out/${R}_k4_stamps.txt
  done
fi
# the pivot step of the in-register Cholesky (tools/ubench/pivot_chain.hip; profiles/r06_pivot_step.txt holds the round's annotated copy)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Igpismap_amd/csrc -Iinclude tools/ubench/pivot_chain.hip -o /tmp/pivot_chain > /dev/null 2>&1 && /tmp/pivot_chain > gpurun_out/${R}_pivot_step_raw.txt 2>&1
{ python3 tools/seq_bench.py --gpu-alone; python3 tools/seq_bench.py; } > gpurun_out/${R}_seq_bench.txt 2>&1
mkdir -p profiles; cp gpurun_out/${R}_k4_traffic.json profiles/ 2>/dev/null     # (bench.py reports roofline.traffic from it while ongpis_test.hip keeps its sha)
python bench.py > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench_err.txt
head -12 gpurun_out/${R}_bench_kernel_stats.txt | cut -c1-150; head -6 gpurun_out/${R}_stress_kstats.txt | cut -c1-150; cat gpurun_out/${R}_k4_traffic.json; cat gpurun_out/${R}_k4_curve.txt; head -8 gpurun_out/${R}_seq_bench.txt | cut -c1-250
