#!/bin/bash
# Runs ON THE GPU BOX: the long checks and the tiled / batched timings whose last lines DESIGN.md quotes, on the build in the
# tree; everything lands in gpurun_out/evidence_<round>/ and is copied to profiles/<round>_*.log afterwards (tools/keep_evidence.py).
# Usage: tools/collect_evidence.sh r03
set -u
RND=${1:-r03}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/evidence_$RND; mkdir -p $OUT
cd $R
# (the tool writes straight into a file under gpurun_out/ — progress gpurun can see; the kept log is its last lines)
run() { name=$1; shift; echo "== $name: $*" | tee $OUT/$name.log; "$@" > $OUT/$name.full 2>&1; echo "exit code $?" >> $OUT/$name.full; grep -v amdgpu.ids $OUT/$name.full | tail -${TAIL:-12} | tee -a $OUT/$name.log; rm -f $OUT/$name.full; }
run fuzz_parity   python3 tools/fuzz_parity.py 20000 100
run fuzz_batch    python3 tools/fuzz_batch.py 6000 5000
run fuzz_trace    python3 tools/fuzz_trace.py 1500
run fuzz_splat    python3 tools/fuzz_splat.py 3000 1
run soak_streams  python3 tools/soak_streams.py 6 1500
TAIL=40 run bench_tiled_batch   python3 tools/bench_tiled.py --batch
TAIL=40 run bench_tiled_single  python3 tools/bench_tiled.py
TAIL=40 run bench_tiled_streams python3 tools/bench_tiled_streams.py --parts 8 --streams 1 2 3 4 --batch 2 4 8
TAIL=40 run bench_configs       python3 tools/bench_configs.py
ls $OUT
