#!/bin/bash
# Runs ON THE GPU BOX: per-kernel time of the re-projection (8.4 M random points -> 2048^2) under rocprofv3 --kernel-trace --stats,
# then FETCH_SIZE / WRITE_SIZE per kernel in separate --pmc passes.  Usage: tools/prof_splat.sh [tag] [pmc]
set -u
TAG=${1:-splat}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/run_splat.py 30 > $OUT/trace.log 2>&1 || tail -3 $OUT/trace.log
grep "splat" $(ls $OUT/trace/*/*kernel_stats.csv | head -1) | cut -c1-200
if [ "${2:-}" = "pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_$C
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/tools/run_splat.py 10 > $OUT/pmc_$C.log 2>&1 || tail -3 $OUT/pmc_$C.log
  done
  python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "splat" in r["Kernel_Name"]:
                d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        v.sort()
        print(c, k, "median KiB per launch", v[len(v) // 2], "->", v[len(v) // 2] * 1024 / 1e6 * (2 if c == "FETCH_SIZE" else 1), "MB" + (" (doubled: gfx950 correction)" if c == "FETCH_SIZE" else ""))
PY
fi
