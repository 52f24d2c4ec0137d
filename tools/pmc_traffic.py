#!/usr/bin/env python3
"""HBM traffic per launch of the projection sweep kernels from the rocprofv3 --pmc passes of tools/gpu_pmc.sh.

    python tools/pmc_traffic.py gpurun_out/pmc_<tag> profiles/<name>.json

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes
(MI355X_MICROARCH.md, HBM section: 128-B requests tallied at 64 B), so reads = 2 * FETCH_SIZE KiB; WRITE_SIZE is
exact for 16-byte-per-lane stores.  All reads of these kernels are 16-byte-per-lane (LDS-DMA and fragment loads).
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    out = {k: {c: v / len(calls[k]) for c, v in cs.items()} for k, cs in acc.items()}
    for k in out:
        out[k]["_dispatches"] = float(len(calls[k]))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    merged = collections.defaultdict(dict)
    for p in ("sq1", "sq2", "fetch", "write"):
        f = sorted(glob.glob("%s/%s/*/*counter_collection.csv" % (src, p)))
        if f:
            for k, cs in per_kernel(f[-1]).items():
                merged[k].update(cs)
    out = {}
    for k, cs in merged.items():
        if "proj_kernel" not in k:
            continue
        if "true>" in k:
            continue                                   # the split (parity) instantiation is never timed
        mode = {"0": "proj_fwd", "1": "proj_fdh_fused", "2": "proj_dw_fused"}[k[k.index("<") + 1]]
        d = dict(cs)
        steps = max((c.get("_dispatches", 0.0) for kk, c in merged.items() if "proj_kernel<2" in kk and "true>" not in kk), default=0.0)
        d["launches_per_step"] = d.pop("_dispatches", 0.0) / steps if steps else 1.0
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            d["hbm_read_bytes"] = 2.0 * cs["FETCH_SIZE"] * 1024.0
            d["hbm_write_bytes"] = cs["WRITE_SIZE"] * 1024.0
            d["hbm_bytes"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
            # busy cycles summed over 1024 SIMDs / (kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs)
            d["mfma_pipe_util"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (cs["GRBM_GUI_ACTIVE"] / 8.0)
        if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
            d["l2_hit_rate"] = cs["TCC_HIT_sum"] / (cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"])
        out[mode] = d
    json.dump({"source": src, "note": "per-launch averages over a 4-step bench run; see tools/pmc_traffic.py for the corrections",
               "kernels": out}, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps({m: {k: d.get(k) for k in ("hbm_bytes", "launches_per_step", "mfma_pipe_util", "l2_hit_rate")} for m, d in out.items()}))


if __name__ == "__main__":
    main()
