#!/usr/bin/env python3
"""Markdown table of the kernels' resource usage from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: make -C toroidal_ray_tracing_amd/csrc asm 2> /tmp/asm.log; python tools/isa_table.py /tmp/asm.log"""
import re, subprocess, sys
text = open(sys.argv[1]).read()
rows, cur = [], None
for ln in text.splitlines():
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("| kernel (`<Real, STATS, DK, RD>` for the listed kernel) | SGPRs | VGPRs | scratch B/lane | waves/SIMD | LDS B/block |")
print("|---|---|---|---|---|---|")
for r, n in zip(rows, names):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n).replace("trt::", "")
    print(f"| `{n}` | {r.get('TotalSGPRs')} | {r.get('VGPRs')} | {r.get('ScratchSize')} | {r.get('Occupancy')} | {r.get('LDS')} |")
