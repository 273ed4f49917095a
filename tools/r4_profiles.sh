#!/bin/bash
# GPU box: the round-4 profile evidence: bench kernel stats, K4 traffic, stress kernel stats and counters of the fused training
# kernel, the update() counters / timeline / pipelined timing, the K4 efficiency curve, the bundled sequences (GPU alone and
# beside the CPU oracle), the default bench line.  Results under gpurun_out/ (copy into profiles/).
mkdir -p gpurun_out
export TMPDIR=/tmp ROUND=r04
bash tools/kstats.sh gpurun_out/r04_bench_kernel_stats.txt bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-api > /dev/null 2>&1
bash tools/kstats.sh gpurun_out/r04_stress_kstats.txt tools/stress_bench.py 50000 > /dev/null 2>&1
bash tools/stress_pmc.sh 50000 > /dev/null 2>&1
bash tools/measure_traffic.sh > /dev/null 2>&1
bash tools/update_pmc.sh traffic > /dev/null 2>&1; cp gpurun_out/update_pmc.txt gpurun_out/r04_update_pmc.txt
bash tools/update_pmc.sh sq > /dev/null 2>&1; cat gpurun_out/update_pmc.txt >> gpurun_out/r04_update_pmc.txt
bash tools/update_timeline.sh 5 > /dev/null 2>&1; { cat gpurun_out/update_profile.txt; grep -E "ongpis|obsgp|fused" gpurun_out/update_timeline.txt; } > gpurun_out/r04_update_timeline.txt
python3 tools/update_pipeline.py 8 2>&1 | tail -5 > gpurun_out/r04_update_pipeline.txt
bash tools/k4_curve.sh > /dev/null 2>&1
{ python3 tools/seq_bench.py --gpu-alone; python3 tools/seq_bench.py; } > gpurun_out/r04_seq_bench.txt 2>&1
mkdir -p profiles; cp gpurun_out/r04_k4_traffic.json profiles/ 2>/dev/null     # (bench.py reports roofline.traffic from it while ongpis_test.hip keeps its sha)
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_err.txt
head -12 gpurun_out/r04_bench_kernel_stats.txt | cut -c1-150; head -6 gpurun_out/r04_stress_kstats.txt | cut -c1-150; cat gpurun_out/r04_k4_traffic.json; cat gpurun_out/r04_k4_curve.txt; head -8 gpurun_out/r04_seq_bench.txt | cut -c1-250
