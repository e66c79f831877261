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


def steady_state(d, skip=100):
    """Per kernel from the dispatch trace: the average over ALL dispatches (what --stats prints: it includes the
    process's cold first frames, when the GPU has not reached its running clocks) next to the average over the
    dispatches after the first `skip` of that kernel (the steady state bench.py's dispatch-bound events measure;
    VERDICT r02 #7: the two pieces of evidence must agree without prose), minimum and median."""
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        per = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                per.setdefault(name, []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        print(f"# {os.path.relpath(f, d)}  (durations in ns; steady = dispatches after the first {skip} of each kernel, or the second half)")
        print("kernel,dispatches,avg_all_ns,steady_dispatches,avg_steady_ns,median_steady_ns,min_ns")
        for name, v in sorted(per.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
            v.sort()
            dur = [x[1] for x in v]
            k = skip if len(dur) > 2 * skip else len(dur) // 2
            st = sorted(dur[k:]) or dur
            print(f"{name},{len(dur)},{sum(dur) / len(dur):.0f},{len(st)},{sum(st) / len(st):.0f},{st[len(st) // 2]},{min(dur)}")
        print()


def from_csv_dir(d):
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print(f"# {os.path.relpath(f, d)}")
        print(open(f).read())
    steady_state(d)
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
