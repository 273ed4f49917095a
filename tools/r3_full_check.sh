#!/bin/bash
# GPU box: the whole GPU suite, then the bench line.
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/full_tests.txt
python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_err.txt
tail -5 gpurun_out/full_tests.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step','update_ms_per_frame')}, d['roofline']['frac'], d['stress'] and (d['stress']['train_ms'], d['stress']['train_tflops'], d['stress']['predict_tflops']), d['update_roofline']['achieved'])
print(d['cpu_baseline'] and {k:v for k,v in d['cpu_baseline'].items() if 'rmse' in k or 'points' in k})
PY
