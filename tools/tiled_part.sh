#!/bin/bash
# Runs ON THE GPU BOX: separates the host's launch cost from GPU latency for a 1/8 part of the config-3 frame
# (VERDICT r02 item 1a).  Wall clock (eager and hipGraph, interleaved in one process) and, from separate
# `rocprofv3 --kernel-trace` runs, the GPU-side span / busy time per frame.  Usage: tools/tiled_part.sh [parts=8]
set -u
P=${1:-8}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tiled_part; mkdir -p $OUT; rm -f $OUT/report.json
python3 $R/tools/bench_tiled_streams.py --parts $P --streams 1 2 3 4 --batch 2 4 $P --both 2>&1 | grep -v amdgpu.ids | tee $OUT/wall.log
# what bench.py --gpus N runs: N frames per launch, the launches alternating over two streams; and the full frame (1 part) beside it
python3 $R/tools/bench_tiled_streams.py --parts $P --streams --batch $P --batch-streams 2 3 --both 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wall.log
python3 $R/tools/bench_tiled_streams.py --parts 1 --streams 1 2 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wall.log
cd /tmp && export TMPDIR=/tmp
for MODE in eager graph; do
  for K in 1 2 4; do
    FLAG=""; [ $MODE = graph ] && FLAG="--graph"
    rm -rf $OUT/trace_${MODE}_$K
    rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_${MODE}_$K -- python3 $R/tools/bench_tiled_streams.py --parts $P --streams $K $FLAG > $OUT/run_${MODE}_$K.log 2>&1 || tail -3 $OUT/run_${MODE}_$K.log
    T=$(ls $OUT/trace_${MODE}_$K/*/*kernel_trace.csv | head -1)
    python3 $R/tools/tiled_trace_report.py $T --label "${P} parts, part 0, $K stream(s), $MODE (under rocprofv3 --kernel-trace)" --json $OUT/report.json
    grep "us per frame" $OUT/run_${MODE}_$K.log
    rm -rf $OUT/trace_${MODE}_$K
  done
done
# $P frames per launch on one stream (trt_render_batch_dev)
rm -rf $OUT/trace_batch
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_batch -- python3 $R/tools/bench_tiled_streams.py --parts $P --streams --batch $P > $OUT/run_batch.log 2>&1 || tail -3 $OUT/run_batch.log
T=$(ls $OUT/trace_batch/*/*kernel_trace.csv | head -1)
python3 $R/tools/tiled_trace_report.py $T --min-frames 16 --frames-per-kernel $P --label "${P} parts, part 0, 1 stream, $P frames per launch, eager (under rocprofv3 --kernel-trace)" --json $OUT/report.json
grep "us per frame" $OUT/run_batch.log
rm -rf $OUT/trace_batch
# ... and the same on two streams (bench.py --gpus N)
rm -rf $OUT/trace_batch2
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_batch2 -- python3 $R/tools/bench_tiled_streams.py --parts $P --streams --batch $P --batch-streams 2 > $OUT/run_batch2.log 2>&1 || tail -3 $OUT/run_batch2.log
T=$(ls $OUT/trace_batch2/*/*kernel_trace.csv | head -1)
python3 $R/tools/tiled_trace_report.py $T --min-frames 16 --frames-per-kernel $P --label "${P} parts, part 0, 2 streams, $P frames per launch, eager (under rocprofv3 --kernel-trace)" --json $OUT/report.json
grep "us per frame" $OUT/run_batch2.log
rm -rf $OUT/trace_batch2
