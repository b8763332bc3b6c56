#!/bin/bash
# usage: tools_trace.sh <tag> <bench args...>  — kernel-trace only, prints per-kernel stats
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log || tail -5 $OUT/err.log
cat $OUT/*/*kernel_stats.csv | cut -c1-160
