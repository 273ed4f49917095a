#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ongpis.py tests/test_gpu_stress.py tests/test_gpu_map2.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/small_tests.txt
python tools/stress_bench.py > gpurun_out/small_stress.txt 2>&1
python tools/k4_bench.py 51 64 8192 3 > gpurun_out/small_k4bench.txt 2>&1
tail -4 gpurun_out/small_tests.txt; tail -3 gpurun_out/small_stress.txt; tail -2 gpurun_out/small_k4bench.txt
