#!/usr/bin/env python3
"""Agreement between bench.py's kernel_ms (HIP events / dispatch timestamps in the replay pass) and the rocprofv3
kernel trace of the SAME command:  tools/prof_agree.py <bench.json> <kernel_stats.csv> [steps_in_trace]
Per-step milliseconds from the trace = TotalDurationNs / steps in the trace (settle + warm-up + timed + replay)."""
import csv
import json
import sys

bench = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(sys.argv[2])))
# steps in the trace: warm-up + the timed region repeated ms_per_step_regions.n times + the replay pass of the kernel
# brackets; counted from the trace itself (one optimizer launch set per step) when it holds the adam kernel
regions = bench.get("ms_per_step_regions", {}).get("n", 1)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else bench["warmup"] + (regions + 1) * bench["steps"] + bench.get("settle", {}).get("steps", 0) + \
    ((bench.get("roofline") or {}).get("alone", {}).get("steps", -2) + 2)     # the sweeps-alone pass and its 2 warm-up steps
# the fp32-class leg of the same command launches the <.., true> (split) instantiations of the same templates: not the
# timed mode, left out here
rows = [r for r in rows if "true>" not in r["Name"]]
tot = {}
for r in rows:
    tot[r["Name"]] = tot.get(r["Name"], 0.0) + float(r["TotalDurationNs"])
# one weight-gradient sweep per bf16 step (settle, cold figure, warm-up, timed regions, replay, sweeps alone): the
# trace's own step count
dw_calls = sum(int(float(r["Calls"])) for r in rows if "proj_kernel<2" in r["Name"])
if len(sys.argv) <= 3 and dw_calls:
    steps = dw_calls


def trace_ms(*subs):
    return sum(v for k, v in tot.items() if any(s in k for s in subs)) / steps * 1e-6


pairs = [("proj_fdh_fused", ("proj_kernel<1>", "proj_kernel<1,")), ("proj_dw_fused", ("proj_kernel<2>", "proj_kernel<2,")),
         ("gru_fwd", ("gru_fwd_bf16_kernel",)), ("gru_bwd", ("gru_bwd_bf16_kernel",)),
         ("wgrad", ("wgrad_kernel", "wgrad_pair_kernel")), ("gi_gemm", ("rowgemm_kernel<0",)), ("demb_gemm", ("rowgemm_kernel<1",)),
         ("mlp_fwd", ("mlp_fwd_kernel",)), ("mlp_bwd", ("mlp_bwd_kernel",))]
calls = {}
for r in rows:
    calls[r["Name"]] = calls.get(r["Name"], 0) + int(float(r["Calls"]))
fdh_calls = sum(v for k, v in calls.items() if "proj_kernel<1" in k)
print("launches of the forward + dL/dh sweep in the trace: %d (steps assumed: %d)\n" % (fdh_calls, steps))
print("| kernel (bench tag) | bench.py kernel_ms (per step) | rocprofv3 trace (per step) | ratio |")
print("|---|---|---|---|")
for tag, subs in pairs:
    if tag in bench["kernel_ms"]:
        a, b = bench["kernel_ms"][tag], trace_ms(*subs)
        print("| %s | %.4f | %.4f | %.3f |" % (tag, a, b, a / b if b else float("nan")))
print("\nsteps in the trace: %d; bench: %.0f seq/s, %.4f ms/step" % (steps, bench["value"], bench["ms_per_step"]))
