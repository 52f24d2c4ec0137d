#!/usr/bin/env python3
"""GPU busy time (union of kernel intervals) vs wall span from a rocprofv3 --kernel-trace csv: is the step host-bound?"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
adam = [i for i, r in enumerate(iv) if 'adam_kernel' in r[2]]
a, b = adam[len(adam) // 2], adam[-1]
seg = iv[a + 1:b + 1]
nsteps = len(adam) - 1 - len(adam) // 2
span = seg[-1][1] - iv[a][1]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("steps %d: wall %.1f us/step, GPU busy (union) %.1f us/step = %.1f%%, kernel sum %.1f us/step" % (
    nsteps, span / nsteps / 1e3, busy / nsteps / 1e3, 100.0 * busy / span, sum(e - s for s, e, _ in seg) / nsteps / 1e3))
