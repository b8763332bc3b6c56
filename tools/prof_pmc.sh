#!/bin/bash
# usage: tools_prof.sh <variant> <tag>   (run on the GPU box from the repo root)
set -e
V=$1; TAG=$2; shift 2; EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TRT_LIB=$R/toroidal_ray_tracing_amd/libtrt_tuning.so   # the knobs exist in the -DTRT_TUNING build only
TRT_DEBUG_SKIP=${TRT_DEBUG_SKIP:-0} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --variant $V --steps 20 --warmup 3 --frames-per-step 4 --no-secondary --no-cpu-baseline $EXTRA > $OUT/bench.json 2> $OUT/trace.err || tail -5 $OUT/trace.err
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  TRT_DEBUG_SKIP=${TRT_DEBUG_SKIP:-0} rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --variant $V --steps 3 --warmup 1 --frames-per-step 2 --no-secondary --no-cpu-baseline $EXTRA > /dev/null 2> $OUT/pmc_$N.err || tail -3 $OUT/pmc_$N.err
done
find $OUT -name "*.csv" | head -30
