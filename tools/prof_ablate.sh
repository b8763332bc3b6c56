#!/bin/bash
# Runs ON THE GPU BOX: per-kernel durations of the re-projection under the ablations given (tools/ablate_splat.py).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ablate; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in "$@"; do
  rm -rf $OUT/t_$S
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$S -- python3 $R/tools/ablate_splat.py $S > $OUT/run_$S.log 2>&1
  echo "== skip $S"
  python3 - $OUT/t_$S/*/*kernel_stats.csv <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "splat" in r["Name"]:
        print(f'{r["Name"].split("(")[0][-48:]:50s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1000:8.2f} us  min {float(r["MinNs"])/1000:8.2f}')
P
  rm -rf $OUT/t_$S
done
