#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of gpurun_out/profiles_<round>/ into profiles/ (tracked) and
derives profiles/traffic_<round>.json (HBM bytes per launch of the render kernel).
Usage: python tools/collect_profiles.py r01"""
import collections, csv, glob, json, os, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", f"profiles_{rnd}"), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    """newest match (gpurun_out/ accumulates the outputs of every run of make_profiles.sh)"""
    files = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return files[-1] if files else None


stats = one("trace/*/*kernel_stats.csv")
for pattern, suffix in (("trace/*/*kernel_stats.csv", "kernel_stats.csv"), ("trace1/*/*kernel_stats.csv", "kernel_stats_one_output_set.csv")):
    f = one(pattern)
    if f:
        # this library's kernels only (bench.py's secondary configurations build their inputs with torch kernels)
        lines = open(f).read().splitlines()
        keep = [lines[0]] + [ln for ln in lines[1:] if "trt::" in ln.split(",")[0]]
        open(os.path.join(dst, f"{rnd}_{suffix}"), "w").write("\n".join(keep) + "\n")
for name in ("bench.json", "bench_under_rocprof.json", "bench_under_rocprof_one_set.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{rnd}_{name}"))

counters = collections.defaultdict(lambda: collections.defaultdict(list))
for pdir in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    f = one(os.path.relpath(pdir, src) + "/*/*counter_collection.csv") if os.path.isdir(pdir) else None
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        counters[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: {c: {"per_launch_median": sorted(v)[len(v) // 2], "launches": len(v)} for c, v in d.items()}
           for k, d in counters.items() if "trt::" in k}
json.dump(summary, open(os.path.join(dst, f"{rnd}_pmc_summary.json"), "w"), indent=1, sort_keys=True)

traffic = {}
variant = json.load(open(os.path.join(src, "bench.json")))["config"]["kernel_variant"] if os.path.exists(
    os.path.join(src, "bench.json")) else "listed"
fetch = write = 0.0
# one frame = one classify launch + one render launch; the render kernel has two instantiations
# (with / without query counters): take the one the timed steps use (most launches)
# the headline frame = tile_classify_kernel + the plain FP32 instantiation of the variant's render kernel (the PMC passes
# also run bench.py's secondary configurations: FP64, RenderedData, persistent … — not part of the headline traffic)
want = {"listed": "render_listed_kernel<float, false, false, false, false, false>", "persistent": "render_persistent_kernel<float>",
        "static": "render_static_kernel<float, 8, false>"}[variant]
for k, d in summary.items():
    if want in k or k.endswith("tile_classify_kernel<false, false>"):
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md §HBM: FETCH_SIZE
        # tallies 128-B read requests at 64 B, so wide coalesced reads count HALF -> doubled here;
        # WRITE_SIZE is exact for 16-B-per-lane streaming stores (this kernel's dominant stores).
        fetch += 2.0 * 1024.0 * d.get("FETCH_SIZE", {}).get("per_launch_median", 0.0)
        write += 1024.0 * d.get("WRITE_SIZE", {}).get("per_launch_median", 0.0)
shafile = os.path.join(src, "kernel_sources.sha256")
if os.path.exists(shafile):
    traffic["kernel_sources_sha256"] = open(shafile).read().strip()
traffic["size"], traffic["depth"] = 4096, 5   # the workload of the passes (bench.py defaults); bench.py ignores the file for any other
traffic[variant] = {"hbm_bytes_per_launch": fetch + write, "write_bytes": write, "fetch_bytes_corrected": fetch,
                    "note": "classify + render kernels of one frame, frames rotating over four output sets (bench.py --output-sets 4); "
                            "FETCH_SIZE doubled per the gfx950 correction"}
json.dump(traffic, open(os.path.join(dst, f"traffic_{rnd}.json"), "w"), indent=1)
print("\n".join(ln[:160] for ln in open(os.path.join(dst, f"{rnd}_kernel_stats.csv")).read().splitlines()) if stats else "no kernel stats")
print(json.dumps(traffic, indent=1))
