#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 `rocpd` sqlite database (what `--stats` prints as CSV in other builds).
    python tools/rocpd_stats.py gpurun_out/<dir> [top_n] [--csv out.csv]"""
import collections
import glob
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 30
    csv_out = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
    dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
    agg = collections.defaultdict(lambda: [0, 0])
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        names = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
        for t in names:
            suf = t.replace("rocpd_kernel_dispatch", "")
            q = (f"select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{suf} d "
                 f"join rocpd_info_kernel_symbol{suf} s on d.kernel_id = s.id")
            for n, s, e in cur.execute(q):
                a = agg[re.sub(r"\s+", " ", n)]
                a[0] += 1
                a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for n, (c, t) in rows[:top]:
        print("%-120s calls=%5d tot=%9.3fms avg=%9.1fus %5.1f%%" % (n[:120], c, t / 1e6, t / c / 1e3, 100.0 * t / tot))
    print("total kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(v[0] for v in agg.values())))
    if csv_out:
        with open(csv_out, "w") as f:
            f.write("kernel,calls,total_ms,avg_us,percent\n")
            for n, (c, t) in rows:
                f.write('"%s",%d,%.3f,%.1f,%.2f\n' % (n.replace('"', "'"), c, t / 1e6, t / c / 1e3, 100.0 * t / tot))


if __name__ == "__main__":
    main()
