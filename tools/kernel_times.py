"""Per-kernel average durations from a rocprofv3 --kernel-trace sqlite database (rocprofv3's default output here).
usage: python tools/kernel_times.py <dir-or-db> [substring ...]"""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    want = sys.argv[2:]
    for db_path in dbs:
        cur = sqlite3.connect(db_path).cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if "kernel_dispatch" in t][0]
        ks = [t for t in tabs if "kernel_symbol" in t][0]
        q = ("select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s "
             "on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, ks))
        for name, n, avg, lo, hi in cur.execute(q):
            if not want or any(w in name for w in want):
                print("%-72s n=%-4d avg=%8.1f us  min=%8.1f  max=%8.1f" % (name[:72], n, avg / 1e3, lo / 1e3, hi / 1e3))


if __name__ == "__main__":
    main()
