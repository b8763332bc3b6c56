#!/usr/bin/env python3
"""Copies the logs of tools/collect_evidence.sh (gpurun_out/evidence_<round>/*.log) and of tools/tiled_part.sh
(gpurun_out/tiled_part/) into profiles/ (tracked): profiles/<round>_<name>.log, profiles/<round>_tiled_part.json.
usage: keep_evidence.py r03"""
import glob, json, os, shutil, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", f"evidence_{rnd}", "*.log"))):
    shutil.copy(f, os.path.join(dst, f"{rnd}_{os.path.basename(f)}"))
    print(os.path.basename(f), "->", open(f).read().strip().splitlines()[-1][:160])
tp = os.path.join(root, "gpurun_out", "tiled_part")
if os.path.exists(os.path.join(tp, "report.json")):
    rep = {"what": "a 1/8 part of the config-3 frame (4096^2, maxDepth 5, part 0, groups of 32 rows), per frame: wall clock of the tool "
                   "(tools/bench_tiled_streams.py, batches of 192 frames) beside the GPU-side figures of `rocprofv3 --kernel-trace` runs of "
                   "the same tool (tools/tiled_trace_report.py): span = first kernel start to last kernel end, busy = union of the kernel "
                   "intervals, kernel_sum = summed durations.  eager = one Python + C-ABI call per frame; graph = one hipGraph replay per "
                   "batch; 'frames per launch' = trt_render_batch_dev",
           "wall_clock_without_profiler": [ln for ln in open(os.path.join(tp, "wall.log")).read().splitlines() if "us per frame" in ln],
           "kernel_trace": json.load(open(os.path.join(tp, "report.json")))}
    json.dump(rep, open(os.path.join(dst, f"{rnd}_tiled_part.json"), "w"), indent=1)
    print("tiled_part.json kept")
