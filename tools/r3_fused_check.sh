#!/bin/bash
# GPU box: parity of the fused training kernel, then timings (plain and instrumented builds).
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ongpis.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/fused_tests.txt
python tools/stress_bench.py > gpurun_out/fused_stress.txt 2>&1
mkdir -p /tmp/ri && cp -r gpismap_amd include /tmp/ri/ && make -s -C /tmp/ri/gpismap_amd/csrc clean && make -s -j8 -C /tmp/ri/gpismap_amd/csrc EXTRA=-DGPIS_INSTRUMENT >/dev/null 2>&1
GPISMAP_AMD_LIB=/tmp/ri/gpismap_amd/libgpismap_amd.so python tools/stress_bench.py > gpurun_out/fused_stress_instr.txt 2>&1
tail -5 gpurun_out/fused_tests.txt; tail -8 gpurun_out/fused_stress.txt; grep "fused trace" gpurun_out/fused_stress_instr.txt | tail -3
