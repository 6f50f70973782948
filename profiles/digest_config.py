"""Condense gpurun_out/prof_<tag>_<cfg>/ (written by profiles/profile_config.sh on the GPU box) into tracked files:
  profiles/<tag>_<cfg>_kernel_stats.csv   the rocprofv3 --kernel-trace --stats table
  profiles/<tag>_<cfg>_pmc.csv            mean counter value per kernel and launch
  profiles/<tag>_<cfg>_digest.txt         per kernel: average duration, cycles, VALU busy, waves per SIMD, lanes per vector
                                          instruction, L2 hit rate, HBM-side traffic per launch
  profiles/traffic.json                   [cfg][kernel family] -> the figures bench.py quotes beside its rooflines
HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE tallies 128-byte requests as 64.
    python3 profiles/digest_config.py <tag> <cfg> [<cfg> ...]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "profiles")
FAMILIES = ("trace_primary_kernel<double, false", "trace_primary_kernel<float, false", "render_primary_kernel", "trace_queue_kernel", "shade_kernel",
            "gather_tile_kernel", "gather_alongside_kernel", "gather_kernel", "gather_stream_kernel", "gather_stream_sss_kernel", "emit_kernel", "resolve_kernel")


def family(kernel):
    for f in FAMILIES:
        if f in kernel:
            return f.replace("<double, false", "<double>").replace("<float, false", "<float>")
    return None


def digest(tag, cfg):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{cfg}")
    stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
    avg_ns = {}
    if stats:
        text = open(stats[0]).read()
        open(os.path.join(DST, f"{tag}_{cfg}_kernel_stats.csv"), "w").write(text)
        for r in csv.DictReader(text.splitlines()):
            f = family(r["Name"])
            if f and "rtkd::" not in r["Name"]:
                calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
                a = avg_ns.setdefault(f, [0, 0.0, 1e30, 0.0])
                a[0] += calls; a[1] += tot; a[2] = min(a[2], float(r["MinNs"])); a[3] = max(a[3], float(r["MaxNs"]))
    agg = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(src, "p*", "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(d)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    lines = ["kernel,counter,launches,mean_per_launch"]
    fam = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, c), v in sorted(agg.items()):
        if "rocclr" in k or "at::native" in k:
            continue
        lines.append(f"\"{k}\",{c},{len(v)},{sum(v) / len(v):.6g}")
        f = family(k)
        if f and "rtkd::" not in k:
            fam[f][c] += v
    open(os.path.join(DST, f"{tag}_{cfg}_pmc.csv"), "w").write("\n".join(lines) + "\n")
    out = {}
    txt = [f"# {cfg} ({tag}): per kernel family and launch; sources profiles/{tag}_{cfg}_kernel_stats.csv, profiles/{tag}_{cfg}_pmc.csv",
           f"{'kernel':34s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'cycles':>10s} {'VALU busy':>9s} {'waves/SIMD':>10s} {'wait any':>8s} {'lanes/VALU':>10s} "
           f"{'VALU inst':>10s} {'L2 hit':>7s} {'HBM-side MB':>11s} {'fetch MB':>9s} {'write MB':>9s}"]
    for f in sorted(set(fam) | set(avg_ns)):
        c = {n: sum(v) / len(v) for n, v in fam[f].items()}
        g = lambda n: c.get(n, 0.0)
        cyc = g("GRBM_GUI_ACTIVE") / 8
        e = {}
        if f in avg_ns:
            n, tot, mn, mx = avg_ns[f]
            e.update(calls=n, avg_us=round(tot / n / 1e3, 2), min_us=round(mn / 1e3, 2), max_us=round(mx / 1e3, 2))
        if cyc > 0:
            e.update(cycles=round(cyc), valu_busy=round(g("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cyc, 4), waves_per_simd=round(g("SQ_WAVE_CYCLES") * 4 / 1024 / cyc, 3),
                     wait_any=round(g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1), 3),
                     lanes_per_valu=round(g("SQ_THREAD_CYCLES_VALU") / max(g("SQ_ACTIVE_INST_VALU"), 1), 1), valu_insts=g("SQ_INSTS_VALU"))
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e.update(hbm_side_bytes=int((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024), fetch_bytes=int(2 * g("FETCH_SIZE") * 1024), write_bytes=int(g("WRITE_SIZE") * 1024))
        if g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
            e["l2_hit"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
        if g("TCP_TOTAL_CACHE_ACCESSES_sum") > 0:
            e["l1_hit"] = round(1 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum"), 4)
        e["source"] = [f"profiles/{tag}_{cfg}_kernel_stats.csv", f"profiles/{tag}_{cfg}_pmc.csv"]
        out[f] = e
        mb = lambda k: ("%.1f" % (e[k] / 1e6)) if k in e else "-"
        txt.append(f"{f:34s} {e.get('calls', 0):6d} {e.get('avg_us', 0):9.2f} {e.get('min_us', 0):9.2f} {e.get('max_us', 0):9.2f} {e.get('cycles', 0):10d} {e.get('valu_busy', 0):9.3f} "
                   f"{e.get('waves_per_simd', 0):10.2f} {e.get('wait_any', 0):8.3f} {e.get('lanes_per_valu', 0):10.1f} {e.get('valu_insts', 0):10.3g} {e.get('l2_hit', 0):7.3f} "
                   f"{mb('hbm_side_bytes'):>11s} {mb('fetch_bytes'):>9s} {mb('write_bytes'):>9s}")
    open(os.path.join(DST, f"{tag}_{cfg}_digest.txt"), "w").write("\n".join(txt) + "\n")
    print("\n".join(txt))
    return out


def main():
    tag, cfgs = sys.argv[1], sys.argv[2:]
    tpath = os.path.join(DST, "traffic.json")
    try:
        traffic = json.load(open(tpath))
    except Exception:
        traffic = {}
    if "f64" in traffic:      # (the earlier flat form of the file)
        traffic = {}
    for cfg in cfgs:
        traffic[cfg] = digest(tag, cfg)
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
