#!/usr/bin/env python3
"""tools/pmc_json.py W.db R.db ENV_ID ENVS TAG > profiles/pmc_latest.json -- HBM bytes per launch of the raster kernel from the
two rocprofv3 PMC passes of tools/profile_round.sh (WRITE_SIZE and FETCH_SIZE: separate runs, counters in KiB; FETCH_SIZE
doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950; steady state = the last 20 launches)."""
import json
import sqlite3
import sys


def table(db, prefix):
    for (n,) in db.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix + "_0"):
            return n
    raise KeyError(prefix)


def steady(path, counter):
    db = sqlite3.connect(path)
    kd, ks = table(db, "rocpd_kernel_dispatch"), table(db, "rocpd_info_kernel_symbol")
    pe, pi = table(db, "rocpd_pmc_event"), table(db, "rocpd_info_pmc")
    rows = db.execute("select s.display_name, e.value, d.start from %s e join %s p on e.pmc_id = p.id join %s d on d.event_id = e.event_id "
                      "join %s s on d.kernel_id = s.id where p.name = ? and s.display_name like '%%raster_kernel%%' order by d.start desc limit 20"
                      % (pe, pi, kd, ks), (counter,)).fetchall()
    return rows[0][0], sum(r[1] for r in rows) / len(rows) * 1024.0


w_db, r_db, env_id, envs, tag = sys.argv[1:6]
name, wbytes = steady(w_db, "WRITE_SIZE")
_, fbytes = steady(r_db, "FETCH_SIZE")
print(json.dumps({"env_id": env_id, "envs_per_gpu": int(envs), "kernel": name[:90], "hbm_bytes_per_launch": wbytes + 2 * fbytes,
                  "write_bytes": wbytes, "fetch_bytes_corrected_x2": 2 * fbytes,
                  "source": "profiles/%s (tools/profile_round.sh: separate rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of bench.py, "
                            "last 20 raster launches; counters are in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md)" % tag}, indent=1))
