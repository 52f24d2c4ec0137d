#!/usr/bin/env python3
"""All kernels of ONE steady-state step (the one whose span is the median) from a rocprofv3 --kernel-trace csv, on
every queue, times relative to the step's first kernel.   python tools/step_timeline.py gpurun_out/prof_<tag> [which]"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mainq = collections.Counter(r['Queue_Id'] for r in rows if 'gru_bwd' in r['Kernel_Name']).most_common(1)[0][0]
# a step begins with the head of the forward on the main queue: the fused gather (embed_wih_prep) or the plain one
head = 'embed_wih_prep_kernel' if any('embed_wih_prep_kernel' in r['Kernel_Name'] for r in rows) else 'embed_fwd_kernel'
starts = [i for i, r in enumerate(rows) if head in r['Kernel_Name'] and r['Queue_Id'] == mainq]
starts = starts[len(starts) // 3:]
spans = sorted((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']), a, b) for a, b in zip(starts[:-1], starts[1:]))
pick = spans[len(spans) // 2] if len(sys.argv) < 3 else spans[int(float(sys.argv[2]) * (len(spans) - 1))]
_, a, b = pick
t0 = int(rows[a]['Start_Timestamp'])
qs = {}
# include kernels that started before the next step's first kernel, plus the stragglers of this step on other queues
for r in rows[a:b + 12]:
    q = qs.setdefault(r['Queue_Id'], 'q%d' % (len(qs) + 1))
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    print("%-44s %s  start %7.1f end %7.1f dur %6.1f  wg %s" % (n, q, s, e, e - s, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])) if 'Grid_Size_X' in r else '?'))
print("span %.1f us" % (pick[0] / 1e3))
