#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite or *_kernel_stats.csv / *_counter_collection.csv) as text
for profiles/.  usage: rocprof_summary.py <dir-or-db> > profiles/rNN_xxx.txt"""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    c = sqlite3.connect(path)
    print(f"# rocprofv3 kernel summary from {os.path.basename(path)} (durations in ns)")
    print("name,calls,total_ns,avg_ns,percent")
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        name = r[0].split("(")[0].replace("void ", "")
        print(f"{name},{r[1]},{r[2]*1000:.0f},{r[3]*1000:.0f},{r[4]:.2f}")
    try:
        rows = list(c.execute(
            "select k.name, p.name, avg(e.value), count(*) from pmc_events e join kernels k on k.id = e.event_id "
            "join pmc_info p on p.id = e.pmc_id group by k.name, p.name"))
    except Exception:
        rows = []
    if rows:
        print("\n# counters (average per dispatch)")
        print("kernel,counter,avg_value,dispatches")
        for k, p, v, n in rows:
            print(f"{k.split('(')[0].replace('void ', '')},{p},{v:.1f},{n}")


def from_csv_dir(d):
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print(f"# {os.path.relpath(f, d)}")
        print(open(f).read())
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
                a = acc.setdefault(key, [0.0, 0])
                a[0] += float(row["Counter_Value"])
                a[1] += 1
        print(f"# {os.path.relpath(f, d)}  (average per dispatch)")
        print("kernel,counter,avg_value,dispatches")
        for (k, cn), (s, n) in sorted(acc.items()):
            print(f"{k},{cn},{s / n:.1f},{n}")


if __name__ == "__main__":
    p = sys.argv[1]
    if os.path.isdir(p):
        dbs = glob.glob(os.path.join(p, "**", "*.db"), recursive=True)
        for db in dbs:
            from_db(db)
        from_csv_dir(p)
    else:
        from_db(p)
