#!/usr/bin/env python3
"""Reads bench.py's JSON line(s) on stdin and prints a few fields of the last one (GPU-box logs stay readable)."""
import json
import sys

last = None
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        try:
            last = json.loads(line)
        except ValueError:
            pass
if last is None:
    sys.exit("no JSON line")
keys = sys.argv[1:] or ["metric", "value", "ms_per_step", "kernel_avg_ms", "roofline"]
for k in keys:
    cur = last
    for part in k.split("."):
        cur = cur.get(part) if isinstance(cur, dict) else None
    print(f"{k}: {json.dumps(cur)[:1500]}")
