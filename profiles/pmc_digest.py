"""Digest of the PMC summaries (profiles/run_pmc.sh): per kernel and launch, the figures the design notes quote.
  python3 profiles/pmc_digest.py gpurun_out/pmc_c3/summary.csv [more summaries] > profiles/<tag>_pmc_digest.txt
VALU / LDS busy are SQ_ACTIVE_INST_* x 4 / (1024 SIMDs or 256 CUs x ... ) over the kernel's cycles, GRBM_GUI_ACTIVE / 8
(the counter sums the eight XCDs); waits are fractions of the wave cycles."""
import csv
import sys

KEEP = ("gather_tile_kernel", "gather_kernel", "gather_alongside_kernel", "trace_queue_kernel", "shade_kernel", "render_primary_kernel",
        "trace_primary_kernel", "emit_kernel")


def main():
    for path in sys.argv[1:]:
        rows = {}
        for r in csv.DictReader(open(path)):
            k = r["kernel"]
            if not any(k.endswith(x) or x + "<" in k or k.split("::")[-1].startswith(x) for x in KEEP):
                continue
            rows.setdefault(k, {})[r["counter"]] = (float(r["mean"]), int(r["launches"]))
        print(f"# {path}")
        print(f"{'kernel':60s} {'launches':>8s} {'cycles':>10s} {'VALU busy':>9s} {'LDS busy':>8s} {'waves/SIMD':>10s} {'wait any':>8s} {'wait inst':>9s} "
              f"{'VALU inst':>10s} {'SALU inst':>10s} {'LDS inst':>10s} {'VMEM rd':>10s} {'lanes/VALU':>10s}")
        for k, c in sorted(rows.items()):
            g = lambda n: c.get(n, (0.0, 0))[0]
            cyc = g("GRBM_GUI_ACTIVE") / 8
            if cyc <= 0:
                continue
            wave_cyc = g("SQ_WAVE_CYCLES") * 4
            busy = g("SQ_BUSY_CYCLES")
            print(f"{k[-60:]:60s} {c['GRBM_GUI_ACTIVE'][1]:8d} {cyc:10.0f} {g('SQ_ACTIVE_INST_VALU') * 4 / 1024 / cyc:9.3f} {g('SQ_ACTIVE_INST_LDS') * 4 / 1024 / cyc:8.3f} "
                  f"{wave_cyc / 1024 / cyc:10.2f} {g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1):8.3f} {g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):9.3f} "
                  f"{g('SQ_INSTS_VALU'):10.3g} {g('SQ_INSTS_SALU'):10.3g} {g('SQ_INSTS_LDS'):10.3g} {g('SQ_INSTS_VMEM_RD'):10.3g} "
                  f"{g('SQ_THREAD_CYCLES_VALU') / max(g('SQ_ACTIVE_INST_VALU'), 1) / 4 * 4:10.1f}")
        print()


if __name__ == "__main__":
    main()
