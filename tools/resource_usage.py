#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: kernel, VGPRs, spills, scratch, LDS.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2>&1 | python tools/resource_usage.py [filter]"""
import re
import subprocess
import sys


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    recs, cur = [], None
    for line in sys.stdin:
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            recs.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in recs),
                           capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(recs, names):
        short = re.sub(r"\(.*", "", n).replace("void ocean::", "")
        if flt and flt not in short:
            continue
        print(f"{short:44s} VGPR {r.get('VGPRs', '?'):>4s} spill {r.get('VGPRs Spill', '?'):>3s} "
              f"scratch {r.get('ScratchSize [bytes/lane]', '?'):>4s} occ {r.get('Occupancy [waves/SIMD]', '?'):>2s} "
              f"LDS {r.get('LDS Size [bytes/block]', '?')}")


if __name__ == "__main__":
    main()
