#!/bin/bash
# usage (GPU box): tools/gaps.sh <tag> [bench args] — per-kernel durations and the idle gaps between consecutive kernels
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gaps_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log || tail -5 $OUT/err.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics
rows = []
for f in glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'):
    rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-28:]) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[-60:]
gaps = {}
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps.setdefault(n0 + ' -> ' + n1, []).append((s1 - e0) / 1e3)
dur = {}
for s, e, n in rows:
    dur.setdefault(n, []).append((e - s) / 1e3)
for k, v in dur.items(): print(f"dur  {k:60s} median {statistics.median(v):7.1f} us")
for k, v in gaps.items(): print(f"gap  {k:60s} median {statistics.median(v):7.1f} us")
print("period", (rows[-1][0] - rows[-41][0]) / 20e3, "us per frame")
PY
