#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) run: per-kernel launch statistics and, if the run collected
PMC counters, per-kernel counter averages.  Usage: rocpd_summary.py results.db [results2.db ...] > profiles/xxx.md"""
import sqlite3
import sys


def table(db, prefix):
    for (n,) in db.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix + "_0"):
            return n
    raise KeyError(prefix)


def summarise(path):
    db = sqlite3.connect(path)
    kd, ks = table(db, "rocpd_kernel_dispatch"), table(db, "rocpd_info_kernel_symbol")
    print("## %s\n" % path)
    # "steady us" = average over the LAST 200 launches of the kernel: bench.py's timed region (the one-off placement
    # probe and the warm-up launch the same kernels first; they are in "avg us" but not in "steady us")
    print("| kernel | launches | total ms | avg us | steady us | min us | max us | grid | wg | lds B | vgpr | sgpr |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    q = ("select s.display_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, "
         "max(d.end-d.start)/1e3, max(d.grid_size_x), max(d.workgroup_size_x), max(d.group_segment_size), "
         "max(s.arch_vgpr_count), max(s.sgpr_count), d.kernel_id from %s d join %s s on d.kernel_id = s.id group by s.display_name "
         "order by 3 desc" % (kd, ks))
    for r in db.execute(q).fetchall():
        name = r[0] if len(r[0]) < 70 else r[0][:67] + "..."
        last = [x[0] for x in db.execute("select d.end - d.start from %s d join %s s on d.kernel_id = s.id where s.display_name = ? "
                                         "order by d.start desc limit 200" % (kd, ks), (r[0],))]
        steady = sum(last) / len(last) / 1e3
        print("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.2f | %d | %d | %d | %d | %d |" % ((name,) + tuple(r[1:4]) + (steady,) + tuple(r[4:11])))
    pe, pi = table(db, "rocpd_pmc_event"), table(db, "rocpd_info_pmc")
    rows = list(db.execute(
        "select s.display_name, p.name, count(*), avg(e.value), min(e.value), max(e.value) from %s e join %s p on e.pmc_id = p.id "
        "join %s d on d.event_id = e.event_id join %s s on d.kernel_id = s.id group by s.display_name, p.name order by 1, 2"
        % (pe, pi, kd, ks)))
    if rows:
        print("\n| kernel | counter | samples | avg per launch | min | max |")
        print("|---|---|---:|---:|---:|---:|")
        for r in rows:
            name = r[0] if len(r[0]) < 70 else r[0][:67] + "..."
            print("| `%s` | %s | %d | %.1f | %.1f | %.1f |" % (name, r[1], r[2], r[3], r[4], r[5]))
    print()


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
