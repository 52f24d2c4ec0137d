#!/usr/bin/env python3
"""Kernel timeline of the second-to-last training step from a rocprofv3 --kernel-trace csv."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['End_Timestamp'])
for r in rows[a + 1:b + 1]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%-44s q%-2s start %7.1f end %7.1f dur %6.1f" % (n, r['Queue_Id'], s / 1e3, e / 1e3, (e - s) / 1e3))
