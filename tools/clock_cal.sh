cd /root/repo; export TMPDIR=/tmp; rm -rf /tmp/cal
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d /tmp/cal -o p -- ${CALCMD:-.ab/mfma_power} > /tmp/cal.log 2>&1
db=$(find /tmp/cal -name "*.db" | head -1)
python3 - "$db" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
q = ("select d.id, d.end-d.start, p.name, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id=p.id "
     "join rocpd_kernel_dispatch d on d.event_id=e.event_id order by d.id")
rows = {}
for did, dur, name, val in c.execute(q):
    rows.setdefault(did, {"dur": dur})
    rows[did][name] = rows[did].get(name, 0) + val
for did, r in sorted(rows.items()):
    if r["dur"] > 2e6:
        g = r.get("GRBM_GUI_ACTIVE", 0)
        print("dispatch %d: %.3f ms  GRBM/8 per ns = %.3f GHz  mfma busy/(GRBM/8*1024) = %.3f" % (did, r["dur"]/1e6, g/8/r["dur"], r.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(g/8*1024) if g else 0))
PY
