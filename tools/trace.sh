#!/bin/bash
# usage (on the GPU box): tools/trace.sh <tag> <bench args...> — rocprofv3 kernel trace + stats of a bench run
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log || tail -5 $OUT/err.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics
for f in glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'):
    d = {}
    for r in csv.DictReader(open(f)):
        d.setdefault(r['Kernel_Name'].split('(')[0][-40:], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in d.items():
        print(f"{k:42s} n={len(v):3d} median {statistics.median(v):8.1f} us  min {min(v):8.1f}")
PY
