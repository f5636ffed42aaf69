#!/usr/bin/env python
"""Per-kernel HBM traffic from the two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950:
MI355X_MICROARCH.md §rocprofv3 PMC slots).

usage: tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [--json out.json]

Units and gfx950 corrections (MI355X_MICROARCH.md §HBM): both counters are reported in KiB; FETCH_SIZE counts the 128-B
requests of a wide coalesced read at 64 B, i.e. HALF the bytes -> it is doubled here.  WRITE_SIZE is "uncalibrated" in the
guide, so the table also prints the layernorm kernel (a pure streaming kernel whose byte counts are known exactly:
reads rows*W*4, writes rows*W*2) as an in-situ calibration row; the per-kernel figures are reported as corrected bytes.
Kernels are keyed by (short name incl. template args, grid size) so the GEMM's shapes stay apart."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(?:\(anonymous namespace\)::|::)?([A-Za-z_][A-Za-z0-9_]*(?:<[^>]*>)?)\(", name)
    return m.group(1) if m else name[:60]


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    keys = sorted(set(fetch) | set(write))
    out = {}
    print(f"{'kernel':44s} {'grid':>9s} {'launches':>8s} {'read MB (2x FETCH_SIZE KiB)':>28s} {'write MB (WRITE_SIZE KiB)':>26s}")
    for k in keys:
        fv, wv = fetch.get(k, []), write.get(k, [])
        rd = 2.0 * 1024.0 * (sum(fv) / len(fv)) if fv else 0.0
        wr = 1024.0 * (sum(wv) / len(wv)) if wv else 0.0
        print(f"{k[0]:44s} {k[1]:9d} {max(len(fv), len(wv)):8d} {rd / 1e6:28.2f} {wr / 1e6:26.2f}")
        out[f"{k[0]}@{k[1]}"] = {"launches": max(len(fv), len(wv)), "read_bytes": rd, "write_bytes": wr}
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(out, f, indent=1)
    else:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
