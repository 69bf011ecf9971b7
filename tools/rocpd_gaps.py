"""Idle time between kernels of a rocprofv3 kernel trace (rocpd sqlite): busy = sum of kernel durations, span = last end -
first start over the steady-state part (the last `frac` of the dispatches), gap histogram.
usage: python tools/rocpd_gaps.py <results.db> [frac=0.5]"""
import sqlite3
import sys

db = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
con = sqlite3.connect(db)
rows = con.execute("select start, end from kernels order by start").fetchall()
rows = rows[int(len(rows) * (1 - frac)):]
busy = sum(e - s for s, e in rows)
span = rows[-1][1] - rows[0][0]
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
overlap = sum(max(0, rows[i][1] - rows[i + 1][0]) for i in range(len(rows) - 1))
print("kernels %d  span %.2f ms  busy %.2f ms (%.1f %%)  idle gaps %.2f ms  overlap %.2f ms" %
      (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(gaps) / 1e6, overlap / 1e6))
gs = sorted(gaps)
print("gap ns: median %d  p90 %d  p99 %d  max %d;  gaps > 5 us: %d (%.2f ms)" %
      (gs[len(gs) // 2], gs[int(len(gs) * 0.9)], gs[int(len(gs) * 0.99)], gs[-1], sum(g > 5000 for g in gs),
       sum(g for g in gs if g > 5000) / 1e6))
