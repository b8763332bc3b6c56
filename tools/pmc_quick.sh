#!/bin/bash
# usage (GPU box): tools/pmc_quick.sh <tag> [bench args] — instruction/cycle counters of the render kernels
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcq_$TAG; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/err.log || tail -3 $OUT/err.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][-36:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'render' in k:
        print(k, ' '.join(f"{c.replace('SQ_','')}={v[-1]/1e6:.1f}M" for c, v in sorted(d.items())))
PY
