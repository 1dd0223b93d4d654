#!/usr/bin/env python3
"""Launch-by-launch durations of one kernel from a rocprofv3 rocpd database, in launch order (is a bimodal average a
drift over time or a per-launch coin toss?).  Usage: rocpd_series.py results.db KERNEL_SUBSTRING [per_line]"""
import sqlite3
import sys

from rocpd_summary import table

db = sqlite3.connect(sys.argv[1])
kd, ks = table(db, "rocpd_kernel_dispatch"), table(db, "rocpd_info_kernel_symbol")
rows = list(db.execute("select d.start, d.end - d.start from %s d join %s s on d.kernel_id = s.id where s.display_name like ? "
                       "order by d.start" % (kd, ks), ("%" + sys.argv[2] + "%",)))
per = int(sys.argv[3]) if len(sys.argv) > 3 else 20
t0 = rows[0][0]
print("%d launches of *%s*; rows: start offset ms | durations us" % (len(rows), sys.argv[2]))
for k in range(0, len(rows), per):
    print("%8.2f | %s" % ((rows[k][0] - t0) / 1e6, " ".join("%5.0f" % (r[1] / 1e3) for r in rows[k:k + per])))
