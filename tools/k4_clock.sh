#!/bin/bash
# For each K4 variant library in .ab/ (tools/k4_ablate.sh build): kernel time, effective clock (GRBM_GUI_ACTIVE / 8 / time) and
# matrix-pipe occupancy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM/8 x 1024 SIMDs)) of the eval kernel in tools/k4_bench.py N 64 8192.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-240}
for lib in .ab/libk4_*.so; do
  tag=$(basename $lib .so); rm -rf /tmp/kc_$tag
  GPISMAP_AMD_LIB=$PWD/$lib rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace -d /tmp/kc_$tag -o p -- python3 tools/k4_bench.py $N 64 8192 1 > /tmp/kc_$tag.log 2>&1
  db=$(find /tmp/kc_$tag -name "*.db" | head -1)
  python3 - "$db" "$tag" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
scol = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
nc = "kernel_name" if "kernel_name" in scol else scol[-1]
q = ("select s.%s, d.id, d.end-d.start, p.name, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id=p.id "
     "join rocpd_kernel_dispatch d on d.event_id=e.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id" % nc)
rows = {}
for name, did, dur, pn, val in c.execute(q):
    if "eval_kernel" not in name: continue
    r = rows.setdefault(did, {"dur": dur}); r[pn] = r.get(pn, 0) + val
for did, r in rows.items():
    g = r.get("GRBM_GUI_ACTIVE", 0) / 8
    print("%-14s %.3f ms  clock %.3f GHz  mfma busy %.3f  wait_any/wave_cycles %.2f" % (sys.argv[2], r["dur"]/1e6, g/r["dur"], r.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(g*1024), r.get("SQ_WAIT_ANY",0)/max(1,r.get("SQ_WAVE_CYCLES",1))))
PY
done
