#!/bin/bash
# usage (GPU box): tools/pmc_trace.sh <tag> [aimed|camera] — SQ counters of trace_kernel (per launch medians)
TAG=$1; KIND=${2:-aimed}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmct_$TAG; mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
         "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS" \
         "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $R/tools/run_trace.py $KIND 5 > $OUT/p$i.out 2> $OUT/p$i.err || tail -3 $OUT/p$i.err
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in agg.items():
    if 'trace' in k:
        out[k] = {c: sorted(v)[len(v) // 2] for c, v in sorted(d.items())}
        for c, v in out[k].items():
            print(f"{k:50s} {c:26s} {v/1e6:12.3f} M")
json.dump(out, open(sys.argv[1] + '/summary.json', 'w'), indent=1)
PY
