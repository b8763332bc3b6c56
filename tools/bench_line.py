#!/usr/bin/env python3
import json, sys
"""usage: bench.py | bench_line.py [tag]   or   bench_line.py file.json [tag]"""
import os
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    d, tag = json.load(open(sys.argv[1])), (sys.argv[2] if len(sys.argv) > 2 else "")
else:
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    d = json.loads(sys.stdin.read())
print(tag, d["config"]["kernel_variant"], "ms/step", round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 4),
      "frac", round(d["roofline"]["frac"], 3), "Gtests/s", round(d["value"] / 1e9, 1))
