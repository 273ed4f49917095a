#!/bin/bash
# GPU box: the whole GPU suite (no -x), then the round-3 profile evidence: bench kernel stats, K4 traffic, stress kernel
# stats and traffic counters of the fused training kernel, the update() timeline / counters / pipelined timing, the K4
# efficiency curve.  Results under gpurun_out/ (copy into profiles/).
mkdir -p gpurun_out
export TMPDIR=/tmp ROUND=r03
python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/full_tests2.txt
tail -4 gpurun_out/full_tests2.txt
bash tools/kstats.sh gpurun_out/r03_bench_kernel_stats.txt bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-api > /dev/null 2>&1
bash tools/kstats.sh gpurun_out/r03_stress_kstats.txt tools/stress_bench.py 50000 > /dev/null 2>&1
bash tools/stress_pmc.sh 50000 > /dev/null 2>&1
bash tools/measure_traffic.sh > /dev/null 2>&1
bash tools/update_pmc.sh traffic > /dev/null 2>&1; cp gpurun_out/update_pmc.txt gpurun_out/r03_update_pmc.txt
bash tools/update_pmc.sh sq > /dev/null 2>&1; cat gpurun_out/update_pmc.txt >> gpurun_out/r03_update_pmc.txt
bash tools/update_timeline.sh 5 > /dev/null 2>&1; { cat gpurun_out/update_profile.txt; grep -E "ongpis|obsgp|fused" gpurun_out/update_timeline.txt; } > gpurun_out/r03_update_timeline.txt
python3 tools/update_pipeline.py 8 2>&1 | tail -5 > gpurun_out/r03_update_pipeline.txt
bash tools/k4_curve.sh > /dev/null 2>&1
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/bench_err.txt
head -12 gpurun_out/r03_bench_kernel_stats.txt | cut -c1-150; head -8 gpurun_out/r03_stress_kstats.txt | cut -c1-150; cut -c1-150 gpurun_out/r03_stress_pmc.txt | head -20; cat gpurun_out/r03_k4_traffic.json
cat gpurun_out/r03_update_pipeline.txt; cat gpurun_out/r03_k4_curve.txt
