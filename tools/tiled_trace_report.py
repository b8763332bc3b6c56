#!/usr/bin/env python3
"""GPU-side figures of a `rocprofv3 --kernel-trace` of tools/bench_tiled_streams.py: the trace is cut into batches (bursts of
kernels separated by more than --gap-us of idle GPU: the tool synchronises between batches) and for every batch of at
least --min-frames frames reports, per frame: the span (first kernel start → last kernel end, ÷ frames: what the GPU
needed, gaps included), the busy time (union of the kernel intervals ÷ frames), the summed kernel durations and the
average duration of the classification and render kernels.  Beside the tool's own wall-clock lines this separates the
host's launch cost from GPU latency.  usage: tiled_trace_report.py <kernel_trace.csv> [--label eager_k4] [--json out.json]"""
import argparse, csv, json, statistics, sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--gap-us", type=float, default=300.0)
ap.add_argument("--min-frames", type=int, default=64)
ap.add_argument("--frames-per-kernel", type=int, default=1, help="frames one render launch covers (trt_render_batch_dev)")
ap.add_argument("--label", default="")
ap.add_argument("--json", default="")
a = ap.parse_args()
rows = []
for r in csv.DictReader(open(a.trace)):
    name = r["Kernel_Name"]
    if "trt::" not in name:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "classify" if "classify" in name else "render" if "render" in name else "other"))
rows.sort()
batches, cur, end = [], [], 0
for s, e, k in rows:
    if cur and s - end > a.gap_us * 1e3:
        batches.append(cur)
        cur = []
    cur.append((s, e, k))
    end = max(end, e)
if cur:
    batches.append(cur)
res = []
for b in batches:
    frames = sum(1 for x in b if x[2] == "render") * a.frames_per_kernel
    if frames < a.min_frames:
        continue
    span = max(x[1] for x in b) - b[0][0]
    busy, end = 0, 0
    for s, e, _ in b:   # union of intervals (sorted by start)
        if e > end:
            busy += e - max(s, end)
            end = e
    res.append({"frames": frames, "span_us_per_frame": span / frames / 1e3, "busy_us_per_frame": busy / frames / 1e3,
                "kernel_sum_us_per_frame": sum(e - s for s, e, _ in b) / frames / 1e3,
                "classify_avg_us": statistics.mean((e - s) / 1e3 for s, e, k in b if k == "classify") if any(k == "classify" for _, _, k in b) else None,
                "render_avg_us": statistics.mean((e - s) / 1e3 for s, e, k in b if k == "render")})
if not res:
    sys.exit("no batch of the requested size in the trace")


def med(key):
    v = [r[key] for r in res if r[key] is not None]
    return statistics.median(v) if v else None


out = {"label": a.label, "batches": len(res), "frames_per_batch": res[-1]["frames"],
       **{k: med(k) for k in ("span_us_per_frame", "busy_us_per_frame", "kernel_sum_us_per_frame", "classify_avg_us", "render_avg_us")}}
print(json.dumps(out))
if a.json:
    try:
        allj = json.load(open(a.json))
    except Exception:
        allj = []
    allj.append(out)
    json.dump(allj, open(a.json, "w"), indent=1)
