#!/bin/bash
# usage (GPU box): tools/pmc_live.sh <tag> [bench args] — wave-cycle breakdown of the render kernel with the
# CLEAR tiles skipped (TRT_DEBUG_SKIP=1): where does a tracing wave spend its lifetime?
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcl_$TAG; mkdir -p $OUT
export TRT_DEBUG_SKIP=${TRT_DEBUG_SKIP-1}
export TRT_LIB=$R/toroidal_ray_tracing_amd/libtrt_tuning.so   # the knobs exist in the -DTRT_TUNING build only
rocprofv3 -L > $OUT/counters.txt 2>&1
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU" \
         "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "GRBM_GUI_ACTIVE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --frames-per-step 2 --no-secondary --no-cpu-baseline "$@" > /dev/null 2> $OUT/p$i.err || tail -3 $OUT/p$i.err
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'render' in k and 'true' not in k:
        for c, v in sorted(d.items()):
            print(f"{k:42s} {c:28s} {v[-1]/1e6:12.3f} M")
PY
