"""Condense a gpurun_out/prof_<tag> directory (written by profiles/run_profile.sh on the GPU box)
into the small tracked files under profiles/: the rocprofv3 --stats kernel table, the PMC means
per kernel, and traffic.json (HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, both in KB;
the factor 2 is the gfx950 FETCH_SIZE correction for 128-B requests tallied at 64 B)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")

stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    with open(stats[0]) as f, open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as g:
        g.write(f.read())
bench = os.path.join(src, "trace_bench.json")
if os.path.exists(bench):
    with open(bench) as f, open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w") as g:
        g.write(f.read())

agg = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(d)):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
lines = ["kernel,counter,launches,mean_per_launch"]
means = {}
for (k, c), v in sorted(agg.items()):
    if "rocclr" in k:
        continue
    means[(k, c)] = sum(v) / len(v)
    lines.append(f"\"{k}\",{c},{len(v)},{means[(k, c)]:.6g}")
open(os.path.join(dst, f"{tag}_pmc.csv"), "w").write("\n".join(lines) + "\n")

traffic = {}
for k in {k for (k, c) in means}:
    # the timed instantiation: trace_primary_kernel<R, COUNT = false, ...>
    if "trace_primary_kernel<double, false" in k or "trace_primary_kernel<float, false" in k:
        prec = "f64" if "<double" in k else "f32"
        fs, ws = means.get((k, "FETCH_SIZE")), means.get((k, "WRITE_SIZE"))
        if fs is not None and ws is not None:
            traffic[prec] = int((2 * fs + ws) * 1024)
        # VALU busy = cycles the SIMDs spend issuing VALU instructions / cycles available:
        # SQ_ACTIVE_INST_VALU counts in units of 4 cycles, summed over the chip's 1024 SIMDs;
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        av, gui = means.get((k, "SQ_ACTIVE_INST_VALU")), means.get((k, "GRBM_GUI_ACTIVE"))
        if av is not None and gui:
            traffic[prec + "_valu_busy"] = round(av * 4 / 1024 / (gui / 8), 4)
        wc, wa, wi = means.get((k, "SQ_WAVE_CYCLES")), means.get((k, "SQ_WAIT_ANY")), means.get((k, "SQ_WAIT_INST_ANY"))
        if wc and gui:
            traffic[prec + "_waves_per_simd"] = round(wc * 4 / 1024 / (gui / 8), 3)
if traffic:
    traffic["source"] = f"profiles/{tag}_pmc.csv: (2 x FETCH_SIZE + WRITE_SIZE) KB per launch of trace_primary_kernel"
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_pmc.csv")).read())
print(traffic)
