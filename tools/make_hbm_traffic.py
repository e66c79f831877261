#!/usr/bin/env python3
"""profiles/hbm_traffic_n<N>[_f16].json from the PMC passes of tools/gpu_evidence.sh / tools/gpu_pmc_sizes.sh:
    python tools/make_hbm_traffic.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <N> <run id> [f16] [bfp16] [normals] [staged]
Per kernel: average FETCH_SIZE / WRITE_SIZE (KB) per dispatch and hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the
gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide streaming read; re-calibrated with
tools/membench.hip: a 1 GiB copy gives FETCH_SIZE = 524 296 KB, WRITE_SIZE = 1 048 576 KB)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("ocean::", "")
                short = name.split("<")[0]
                for suffix in ("_split", "_real"):                 # the N >= 8192 kernels of the same two passes
                    if short.endswith(suffix):
                        short = short[:-len(suffix)]
                a = acc.setdefault(short, [0.0, 0, name])
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return {k: (v[0] / v[1], v[1], v[2]) for k, v in acc.items()}


def main():
    d, n, run = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    f16 = "f16" in sys.argv[4:]
    staged = "staged" in sys.argv[4:]
    bfp16 = "bfp16" in sys.argv[4:]
    normals = "normals" in sys.argv[4:]                             # bench.py --normals: the frame with the normal field
    batch = next((a for a in sys.argv[4:] if a.startswith("batch")), "")   # bench.py --batch K: K frames per dispatch
    fetch, write = counters(d, "FETCH_SIZE"), counters(d, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) & set(write)):
        if not k.startswith("k_") or k.startswith("k_checksum") or k.startswith("k_pack"):
            continue
        kernels[k] = {"kernel": fetch[k][2], "dispatches": fetch[k][1], "FETCH_SIZE_KB": round(fetch[k][0], 1),
                      "WRITE_SIZE_KB": round(write[k][0], 1), "hbm_bytes": (2.0 * fetch[k][0] + write[k][0]) * 1024.0}
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of the bench command, average per dispatch; "
                     "gfx950 correction hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (tools/make_hbm_traffic.py)",
           "run": run, "n": n, "spectrum": "f16" if f16 else "f32", "intermediate": "bfp16" if bfp16 else "f32", "normals": normals, "batch": batch or None, "kernels": kernels}
    name = f"hbm_traffic_{'staged_' if staged else ''}n{n}{'_f16' if f16 else ''}{'_bfp16' if bfp16 else ''}{'_normals' if normals else ''}{'_' + batch if batch else ''}.json"
    with open(os.path.join(ROOT, "profiles", name), "w") as f:
        json.dump(out, f, indent=1)
    print(name, {k: round(v["hbm_bytes"] / 1e6, 1) for k, v in kernels.items()})


if __name__ == "__main__":
    main()
