#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: collects the rocprofv3 evidence that
# bench.py's roofline numbers are judged against.  Usage: tools/make_profiles.sh r01 [variant]
# Writes gpurun_out/profiles_<round>/…; tools/collect_profiles.py then copies the summaries
# into profiles/ (tracked).
set -u
ROUND=${1:-r01}; V=${2:-listed}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles_$ROUND
mkdir -p $OUT
# the kernel sources these passes measure: bench.py reports `roofline.traffic` only while they are unchanged
cat $R/toroidal_ray_tracing_amd/csrc/trt_kernels.hip $R/toroidal_ray_tracing_amd/csrc/trt_device.hpp $R/toroidal_ray_tracing_amd/csrc/trt_kernels.hpp \
    $R/toroidal_ray_tracing_amd/csrc/trt_api.hip | sha256sum | cut -d" " -f1 > $OUT/kernel_sources.sha256
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel time of the command the driver runs, with the timed loop itself rotating over FOUR output sets (--output-sets 4:
#    what bench.py's roofline pass does — no Infinity-Cache reuse between frames): `roofline.frac` of the bench line must be
#    reproducible as 44 B x 4096^2 / (tile_classify + render_listed average of THIS file) / 8 TB/s.  Kernel trace + stats only.
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --variant $V --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --output-sets 4 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || tail -5 $OUT/trace.err
# 1b. the same with ONE output set and the secondary configurations (every kernel of the library appears)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 $R/bench.py --variant $V --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof_one_set.json 2> $OUT/trace1.err || tail -5 $OUT/trace1.err
# 2. HBM traffic, separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass); four output sets as in 1.
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --variant $V --steps 3 --warmup 1 --frames-per-step 4 --no-cpu-baseline --output-sets 4 > /dev/null 2> $OUT/pmc_$C.err || tail -3 $OUT/pmc_$C.err
done
# 3. issue / occupancy counters of the dominant kernel
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_SQ -- python3 $R/bench.py --variant $V --steps 3 --warmup 1 --frames-per-step 4 --no-cpu-baseline --output-sets 4 > /dev/null 2> $OUT/pmc_SQ.err || tail -3 $OUT/pmc_SQ.err
# 4. the un-profiled bench line, for comparison (never compare a profiled arm with an un-profiled one)
python3 $R/bench.py --variant $V --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
ls $OUT
