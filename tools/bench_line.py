#!/usr/bin/env python3
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read())
print(tag, d["config"]["kernel_variant"], "ms/step", round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 4),
      "frac", round(d["roofline"]["frac"], 3), "Gtests/s", round(d["value"] / 1e9, 1))
