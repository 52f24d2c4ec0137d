#!/usr/bin/env python3
"""Per-step critical-path view from a rocprofv3 --kernel-trace csv of bench.py: for every steady-state step (delimited
by the Adam launch that ends it) the span, and how the caller's queue spent it: kernel time by name, and the idle gaps
in front of each kernel (waits for other streams / launch latency).  Averages over steps.
    python tools/step_breakdown.py gpurun_out/prof_<tag>"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
# steps end with the LAST adam_kernel of a step (two per step: floated W3 bucket on its own queue + the rest)
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# the caller's queue = the queue of embed_fwd
mainq = collections.Counter(r['Queue_Id'] for r in rows if 'gru_bwd' in r['Kernel_Name']).most_common(1)[0][0]
# a step starts with the first gather on the caller's queue (a second one, for the rows of the second time range, runs
# on the side queue)
starts = [i for i, r in enumerate(rows) if 'embed_fwd_kernel' in r['Kernel_Name'] and r['Queue_Id'] == mainq]
starts = starts[len(starts) // 3:]                      # steady state
spans, busy, gaps = [], collections.defaultdict(float), collections.defaultdict(float)
n = 0
for a, b in zip(starts[:-1], starts[1:]):
    t0, t1 = int(rows[a]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
    spans.append((t1 - t0) / 1e3)
    last_end = t0
    for r in rows[a:b]:
        if r['Queue_Id'] != mainq:
            continue
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        k = short(r['Kernel_Name'])
        busy[k] += (e - s) / 1e3
        gaps[k] += max(0, s - last_end) / 1e3
        last_end = max(last_end, e)
    n += 1
print("steps %d: start-to-start span mean %.1f us (min %.1f max %.1f)" % (n, sum(spans) / n, min(spans), max(spans)))
print("caller's queue, mean per step: %-40s %8s %8s" % ("kernel", "busy us", "gap before us"))
tb = tg = 0.0
for k in sorted(busy, key=lambda k: -busy[k] - gaps[k]):
    print("   %-58s %8.1f %8.1f" % (k, busy[k] / n, gaps[k] / n))
    tb += busy[k] / n; tg += gaps[k] / n
print("   %-58s %8.1f %8.1f" % ("total", tb, tg))
