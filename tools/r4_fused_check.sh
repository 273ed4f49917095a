#!/bin/bash
# GPU box: parity of the fused training kernels (data-flow schedule vs the barrier schedule), then timings (plain and instrumented).
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ongpis.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4_fused_tests.txt
timeout 300 python tools/stress_bench.py > gpurun_out/r4_fused_stress.txt 2>&1
GPIS_FUSED_V1=1 timeout 300 python tools/stress_bench.py > gpurun_out/r4_fused_stress_v1.txt 2>&1
mkdir -p /tmp/ri && cp -r gpismap_amd include /tmp/ri/ && make -s -C /tmp/ri/gpismap_amd/csrc clean && make -s -j8 -C /tmp/ri/gpismap_amd/csrc EXTRA=-DGPIS_INSTRUMENT >/dev/null 2>&1
GPISMAP_AMD_LIB=/tmp/ri/gpismap_amd/libgpismap_amd.so timeout 300 python tools/stress_bench.py > gpurun_out/r4_fused_stress_instr.txt 2>&1
tail -5 gpurun_out/r4_fused_tests.txt; tail -3 gpurun_out/r4_fused_stress.txt; tail -3 gpurun_out/r4_fused_stress_v1.txt; grep "fused trace" gpurun_out/r4_fused_stress_instr.txt | tail -2
