#!/usr/bin/env python
"""Per-kernel average of the FETCH_SIZE / WRITE_SIZE PMC passes (rocprofv3 --pmc, counter_collection.csv).
usage: tools/pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass>   -> table + JSON on the last line"""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    names = sorted({k[0] for k in list(fetch) + list(write)})
    out = {}
    print(f"{'kernel':62s} {'launches':>8s} {'FETCH_SIZE avg':>16s} {'WRITE_SIZE avg':>16s}")
    for n in names:
        fv = fetch.get((n, "FETCH_SIZE"), [])
        wv = write.get((n, "WRITE_SIZE"), [])
        fa = sum(fv) / len(fv) if fv else 0.0
        wa = sum(wv) / len(wv) if wv else 0.0
        print(f"{n:62s} {max(len(fv), len(wv)):8d} {fa:16.1f} {wa:16.1f}")
        out[n] = {"launches": max(len(fv), len(wv)), "FETCH_SIZE": fa, "WRITE_SIZE": wa}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
