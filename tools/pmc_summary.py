#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter_collection.csv files: python tools/pmc_summary.py <csv> [name-filter]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if filt and filt not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    for k, cs in acc.items():
        n = len(calls[k])
        print("%s  (%d dispatches; per-dispatch averages)" % (k[:100], n))
        for c, v in sorted(cs.items()):
            print("    %-32s %16.1f" % (c, v / n))


if __name__ == "__main__":
    main()
