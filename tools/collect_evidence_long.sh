#!/bin/bash
# Runs ON THE GPU BOX: the long differential runs, ten times the size of collect_evidence.sh's, on the build in the tree;
# logs land beside the others (gpurun_out/evidence_<round>/*_long.log -> profiles/<round>_*_long.log via keep_evidence.py).
set -u
RND=${1:-r03}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/evidence_$RND; mkdir -p $OUT
cd $R
run() { name=$1; shift; echo "== $name: $*" | tee $OUT/$name.log; "$@" > $OUT/$name.full 2>&1; echo "exit code $?" >> $OUT/$name.full; grep -v amdgpu.ids $OUT/$name.full | tail -${TAIL:-6} | tee -a $OUT/$name.log; rm -f $OUT/$name.full; }
run fuzz_parity_long   python3 tools/fuzz_parity.py 150000 100000
run fuzz_trace_long    python3 tools/fuzz_trace.py 6000
run fuzz_batch_long    python3 tools/fuzz_batch.py 30000 50000
run fuzz_splat_long    python3 tools/fuzz_splat.py 12000 7
