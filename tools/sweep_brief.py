#!/usr/bin/env python3
"""tools/sweep.py output, one short line per N (fps and per-kernel microseconds)."""
import json
import sys

for line in sys.stdin:
    try:
        r = json.loads(line)
    except Exception:
        print(line.strip())
        continue
    print(sys.argv[1] if len(sys.argv) > 1 else "", r["n"], round(r["fused_fps"], 1),
          {k: round(v * 1000, 1) for k, v in r["fused"].items()})
