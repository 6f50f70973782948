"""Timeline of the gather kernels of the last frame in a rocprofv3 kernel trace: python3 tools/c4_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gather", "shade", "resolve", "trace_queue", "render_primary"))]
last = [i for i, r in enumerate(sel) if "render_primary" in r["Kernel_Name"]][-1]
t0 = int(sel[last]["Start_Timestamp"])
for r in sel[last:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  {r['Kernel_Name'].split('(')[0][-40:]}")
