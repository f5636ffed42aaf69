#!/usr/bin/env python
"""Per-kernel SQ / GRBM counters of one rocprofv3 --pmc pass (counter_collection.csv), averaged per dispatch.
usage: tools/pmc_sq_summary.py <dir> [--top N]

Derived columns (MI355X_MICROARCH.md §rocprofv3 PMC slots / §per-instruction constants):
  clock      GRBM_GUI_ACTIVE is summed over the 8 XCDs and carries a constant per-dispatch offset (counter start/stop window),
             so the effective shader clock is the SLOPE of GUI/8 against kernel duration over the kernels of this run
             (least squares; printed in the header), not GUI / duration of a single kernel
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * clock * duration): fraction of SIMD-cycles with the matrix pipe busy
  parked / issue_stall / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint buckets)"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(?:\(anonymous namespace\)::|::)?([A-Za-z_][A-Za-z0-9_]*(?:<[^>]*>)?)\(", name)
    return m.group(1) if m else name[:60]


def main():
    d = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur[key].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    rows = []
    for key, ctrs in acc.items():
        v = {c: sum(x) / len(x) for c, x in ctrs.items()}
        n = max(len(x) for x in ctrs.values())
        t_ns = sum(dur[key]) / len(dur[key]) if dur[key] else 0.0
        rows.append((n * t_ns, key, n, t_ns, v))
    rows.sort(reverse=True)
    fit = [(t_ns, v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0) for _, _, n, t_ns, v in rows if n >= 5 and t_ns > 5e3 and v.get("GRBM_GUI_ACTIVE")]
    ghz = 2.1
    if len(fit) >= 2:
        mx, my = sum(a for a, _ in fit) / len(fit), sum(b for _, b in fit) / len(fit)
        den = sum((a - mx) ** 2 for a, _ in fit)
        if den > 0:
            ghz = sum((a - mx) * (b - my) for a, b in fit) / den
    print(f"# effective shader clock from the GRBM_GUI_ACTIVE/8 vs duration slope over {len(fit)} kernels: {ghz:.2f} GHz")
    print(f"{'kernel':40s} {'grid':>8s} {'n':>4s} {'us':>8s} {'mfma_busy':>9s} {'parked':>7s} {'stall':>6s} {'active':>6s}")
    for _, key, n, t_ns, v in rows[:top]:
        wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        print(f"{key[0][:40]:40s} {key[1]:8d} {n:4d} {t_ns / 1e3:8.1f} "
              f"{v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024 * ghz * t_ns) if t_ns else 0:9.3f} {v.get('SQ_WAIT_ANY', 0.0) / wc:7.3f} "
              f"{v.get('SQ_WAIT_INST_ANY', 0.0) / wc:6.3f} {v.get('SQ_ACTIVE_INST_ANY', 0.0) / wc:6.3f}")


if __name__ == "__main__":
    main()
