#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel name: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    for d in sys.argv[1:]:
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        print("==", d, files[:1])
        acc = defaultdict(lambda: defaultdict(list))
        for f in files:
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name", "")
                if "gemm_nt" not in name:
                    continue
                key = (name[name.find("gemm_nt"):][:40], r.get("Grid_Size", ""))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for key, ctrs in sorted(acc.items()):
            print(key)
            vals = {c: sum(v) / len(v) for c, v in ctrs.items()}
            wc = vals.get("SQ_WAVE_CYCLES", 0) or 1
            for c, v in sorted(vals.items()):
                print(f"   {c:28s} {v:16.0f}  ({v / wc:6.3f} of WAVE_CYCLES)")


if __name__ == "__main__":
    main()
