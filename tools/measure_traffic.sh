#!/bin/bash
# HBM-side traffic and pipe occupancy of K4 for the bench configuration, collected exactly as MI355X_MICROARCH.md prescribes:
# separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_* + GRBM_GUI_ACTIVE), each with --kernel-trace only.
# Writes profiles/${ROUND}_k4_pmc.txt (per-kernel sums) and profiles/${ROUND}_k4_traffic.json, which bench.py reads as
# roofline.traffic ONLY while ongpis_test.hip still has the sha recorded here.
# Usage (GPU box): tools/measure_traffic.sh          -> results under gpurun_out/, copy into profiles/ afterwards
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROUND=${ROUND:-r06}; export ROUND
mkdir -p gpurun_out
CMD="bench.py --steps 1 --warmup 0 --cpu-sample 0 --stress 0 --no-host-api --update-repeats 1"   # (one fusion per accounting: with three or more AND the side streams of small test() passes in use, rocprofv3 --pmc shows 0.65 GB of extra reads and writes per K4 launch -- profiles/README.md; run time is the same)
OUT=gpurun_out/${ROUND}_k4_pmc.txt
: > $OUT
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/mt_$tag
  rocprofv3 --pmc $pass --kernel-trace -d /tmp/mt_$tag -o p -- python3 $CMD > /tmp/mt_$tag.log 2>&1
  db=$(find /tmp/mt_$tag -name "*.db" | head -1)
  echo "== rocprofv3 --pmc $pass --kernel-trace -- python3 $CMD" >> $OUT
  python3 profiles/summarize_pmc.py "$db" | grep -E "eval_kernel|^kernel" >> $OUT
  python3 profiles/summarize_rocpd.py "$db" | grep -E "eval_kernel" >> $OUT
done
python3 - <<'PY'
import hashlib, json, os, re
txt = open("gpurun_out/%s_k4_pmc.txt" % os.environ.get("ROUND", "r03")).read()
def total(counter):
    s = 0.0
    for line in txt.splitlines():
        m = re.match(r"\S*eval_kernel\S*\s+%s\s+(\d+)\s+(\d+)" % counter, line)
        if m: s += float(m.group(2))
    return s
launches = 0
for line in txt.splitlines():
    m = re.match(r"\S*eval_kernel\S*\s+FETCH_SIZE\s+(\d+)\s+(\d+)", line)
    if m: launches += int(m.group(1))
fetch_kib, write_kib = total("FETCH_SIZE"), total("WRITE_SIZE")
# gfx950: FETCH_SIZE reports half the bytes of wide (16 B/lane) streaming reads (MI355X_MICROARCH.md, HBM section): x2
hbm = fetch_kib * 1024 * 2 + write_kib * 1024
sha = hashlib.sha256(open("gpismap_amd/csrc/ongpis_test.hip", "rb").read()).hexdigest()[:16]
json.dump({"ongpis_test_sha": sha, "grid": 256, "frames": 5, "k4_launches": launches, "fetch_kib_raw": fetch_kib, "write_kib_raw": write_kib,
           "hbm_bytes_per_pass": hbm, "hbm_bytes_per_launch": hbm / max(1, launches),
           "note": "FETCH_SIZE x 2 (gfx950 correction for 16-B/lane streams) + WRITE_SIZE (raw); one 256^3 test() pass after 5 synthetic frames"},
          open("gpurun_out/%s_k4_traffic.json" % os.environ.get("ROUND", "r03"), "w"), indent=1)
print(open("gpurun_out/%s_k4_traffic.json" % os.environ.get("ROUND", "r03")).read())
PY
cat $OUT | cut -c1-160
