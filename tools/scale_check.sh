#!/bin/bash
# 8-GPU node (driver): the scaling curve and the checks the one-GPU box cannot do over RCCL.
#   tools/scale_check.sh [max_gpus]        default 8
# For N = 1, 2, 4, 8 and the three training modes (replicated, sharded, lead = sharded with update()'s host logic run once on rank 0): `python3 bench.py --gpus N --verify` -- prints the bench line's value, per-rank
# GP evaluations, K4 ms, pass / gather / exchange ms, and FAILS if the assembled 256^3 map differs by a bit from rank 0's
# single-rank pass on a 64^3 sub-grid (bench.py --verify raises).  Then the in-library multi-device map (one process, N
# devices: gpis3_create_multi) against a one-device map.  Logs: gpurun_out/scale_*.json.
set -e
cd "$(dirname "$0")/.."
MAXN=${1:-8}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 1 2 4 8; do
  [ $N -gt $MAXN ] && break
  for MODE in replicated sharded lead; do
    [ $N -eq 1 ] && [ $MODE != replicated ] && continue
    OUT=gpurun_out/scale_${N}_${MODE}.json
    python3 bench.py --gpus $N --train $MODE --verify --cpu-sample 0 --stress 0 --no-host-api > $OUT
    python3 - "$OUT" $N $MODE <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
pr = d["per_rank"]
n = int(sys.argv[2])
print("N=%s %-10s value %.3e pts/s  %.1f ms/step  update %.1f ms/frame" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["update_ms_per_frame"]))
if d.get("exchange_record_bytes_per_frame"):
    rec, got = d["exchange_record_bytes_per_frame"], d["exchange_bytes_per_frame"]
    ex = [v for v in pr.get("exchange_ms_per_frame", []) if v > 0]
    ms = max(ex) if ex else float("nan")
    # all-gather over xGMI: a rank receives (n-1)/n of the records over its n-1 links; one link carries ~ records / n per frame
    print("   exchange: records %.3f GB/frame, received per rank %.3f GB, %.2f ms/frame -> %.1f GB/s into each rank, %.3f GB and %.1f GB/s per link"
          % (rec / 1e9, got / 1e9, ms, got / 1e6 / ms, rec / n / 1e9, rec / n / 1e6 / ms))
print("   gp_evals/rank", [round(v / 1e6, 2) for v in pr["gp_evals"]], "M   k4_ms", [round(v, 1) for v in pr["k4_ms_per_step"]])
print("   host replays of update() per rank over the last fusion", pr.get("host_replays"), " frame record %.2f MB/frame" % (d.get("frame_record_bytes_per_frame", 0) / 1e6))
if "pass_ms" in pr:
    print("   pass_ms", [round(v, 1) for v in pr["pass_ms"]], " gather_ms", [round(v, 1) for v in pr["gather_ms"]], " exchange_ms", [round(v, 1) for v in pr["exchange_ms_per_frame"]],
          " identical", pr.get("assembled_equals_single_rank_on_64cubed"))
PY
  done
done
# one process driving N devices through the library (what the unchanged mex gateway does with GPIS_DEVICES set)
python3 - $MAXN <<'PY'
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, gpismap_amd, replay
n = min(int(sys.argv[1]), gpismap_amd.device_count())
one = gpismap_amd.GPisMap3(); many = gpismap_amd.GPisMap3(devices=list(range(n)))
for f in range(3):
    d = replay.synthetic_depth(f)
    one.update(d, replay.IDENTITY_POSE)
    t0 = time.perf_counter(); many.update(d, replay.IDENTITY_POSE); tu = (time.perf_counter() - t0) * 1e3
x = replay.synthetic_grid(128)
t0 = time.perf_counter(); a = one.test(x); t1 = time.perf_counter(); b = many.test(x); t2 = time.perf_counter()
same = bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
print("in-library multi-device map, %d devices: update %.1f ms, test(128^3, host pointers) %.0f ms vs %.0f ms on one device, identical %s" % (n, tu, (t2 - t1) * 1e3, (t1 - t0) * 1e3, same))
assert same
PY
