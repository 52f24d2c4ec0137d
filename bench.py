#!/usr/bin/env python3
"""Benchmark of the SVAE training step on MI355X (contract: see the task's bench.py section).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--prec bf16|fp32] [--batch B] [--shape ...]

``--gpus N`` with N > 1 and no launcher around it (WORLD_SIZE unset): this process starts N rank processes itself
(one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) BEFORE it touches the GPU and
relays rank 0's JSON line; under ``python -m torch.distributed.run`` it is one of the ranks, as before.

A "step" = forward + backward + (gradient all-reduce) + Adam over one length-bucketed batch of
B users per GPU, synthetic ML-20M-shaped sequences (138k users / 20k items / <=200 items per user;
generator svae_cf_amd/synth.py, seed 98765), inputs resident in HBM before the timed region.
Metric (BASELINE.json): train sequences/sec, whole job.  Weak scaling: B per GPU is fixed.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK = {"hbm": 8000.0, "bf16": 2500.0, "fp32": 157.3}


def spawn_ranks(n):
    """Start n rank processes of this script (no torchrun needed) and relay what they print.  Nothing here touches
    the GPU: the children initialise it.  Returns the largest exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p_ in procs:
        rc = max(rc, abs(p_.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)     # warmup 8 + 48 steps: each of the 16 plans exactly 3 times
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--prec", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--shape", default="ml-20m", choices=["ml-1m", "ml-20m", "netflix", "large"],
                    help="synthetic shape (synth.SHAPES); the default is the one BASELINE.json's metric is quoted on")
    ap.add_argument("--items", type=int, default=None, help="catalogue size (default: the shape's)")
    ap.add_argument("--nplans", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle-steps", type=int, default=48,
                    help="untimed steps before the warm-up steps (an idle GPU ramps up for ~40 ms; 0 = none)")
    ap.add_argument("--no-cold-figure", action="store_true", help="skip the timed region in front of the settle phase")
    ap.add_argument("--sparse-embed", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: exchange the item-embedding gradient as (row id, row) pairs (auto: when fewer bytes)")
    ap.add_argument("--shard-w3", default="auto", choices=["auto", "on", "off"],
                    help="decoder.linear2 gradient: reduce-scatter + sharded optimizer + all-gather (auto: >= 100k items)")
    ap.add_argument("--eval-leg", action="store_true", help="also time the scoring leg (forward + masked top-100)")
    ap.add_argument("--no-kernel-times", action="store_true", help="do not bracket kernels with events (no roofline)")
    ap.add_argument("--no-fp32-class", action="store_true",
                    help="skip the short leg that times the same step at the reference's own precision (PREC_SPLIT)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions in all: the first is the contractual one (`value`), the others give the spread")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # SVAE_DIST_BACKEND=gloo is the rehearsal mode for a one-GPU box (tools/dp_rehearsal.sh): the ranks share
    # the card and the all-reduce goes through gloo; the driver's runs use the default, nccl (= RCCL over xGMI)
    # SVAE_DP_REHEARSAL=1: create the process group and keep the gradient all-reduces even at world size 1
    # (the RCCL call path, stream ordering included, on a one-GPU box)
    backend = os.environ.get("SVAE_DIST_BACKEND", "nccl")
    rehearse = os.environ.get("SVAE_DP_REHEARSAL", "0") == "1"
    dev_index = local_rank % max(1, torch.cuda.device_count()) if world > 1 else 0
    if world > 1 or rehearse:
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    import svae_cf_amd as S
    from svae_cf_amd import synth
    from svae_cf_amd.dist import GradSync, rank_batches

    _, shape_items, mu, sigma, lmax = synth.SHAPES[args.shape]
    I, B = (args.items or shape_items), args.batch
    shape_name = {"ml-1m": "ML-1M", "ml-20m": "ML-20M", "netflix": "Netflix", "large": "large-catalogue"}[args.shape]
    shape_cfg = {"ml-1m": 1, "ml-20m": 2, "netflix": 3, "large": 4}[args.shape]
    n_users = args.nplans * B * world
    users = synth.make_users(n_users, I, mu, sigma, lmax, seed=98765)
    rng = np.random.default_rng(98765)
    # every rank builds the same seeded order; a global batch (B*world users of similar length) is dealt
    # round-robin to the ranks, so rows and Tmax per step are balanced across ranks (dist.rank_batches)
    mine = rank_batches([len(u) for u in users], B, rng, rank, world)
    plans = [S.build_plan([users[i] for i in idx], 4, user_keys=idx, device=dev, total_items=I) for idx in mine]
    rows = float(np.mean([p.R for p in plans]))

    prec = S.PREC_BF16 if args.prec == "bf16" else S.PREC_FP32
    eng = S.SVAEEngine(I, device=dev, prec=prec)
    # the loop's stream, as train_batched picks it (high priority unless SVAE_TRAIN_HIPRIO=0)
    torch.cuda.set_stream(eng.training_stream())
    torch.manual_seed(0)
    sd = {}
    for name, shp in eng.layout.ref_shapes.items():
        if name == "item_embed.weight":
            sd[name] = torch.randn(shp)
        elif name.startswith("gru."):
            sd[name] = (torch.rand(shp) * 2 - 1) / (200 ** 0.5)
        elif name.endswith("weight"):
            sd[name] = torch.randn(shp) * (2.0 / (shp[0] + shp[1])) ** 0.5
        else:
            sd[name] = (torch.rand(shp) * 2 - 1) * 0.05
    eng.load_state_dict(sd)
    eng.reserve(max(p_.R for p_ in plans))         # workspace for the largest batch: no growth inside the timed region
    for p_ in plans:
        eng.prepare(p_)
    if world > 1 or rehearse:
        # SVAE_DP_NATIVE=1: the collectives through the library's own RCCL wrappers, in order on the producing streams
        # (dist.GradSync(native=True); opt-in: it has only run at world size 1)
        eng.grad_sync = GradSync(eng.grads, eng.layout.buckets(), force=rehearse,
                                 native=os.environ.get("SVAE_DP_NATIVE", "0") == "1")
        # large catalogues: the item-embedding gradient (one non-zero row per distinct input item of the batch) goes
        # over the wire as (row id, row) pairs when that is fewer bytes than the dense all-reduce of the table
        cap_rows = max(p_.R for p_ in plans)
        if args.sparse_embed == "on" or (args.sparse_embed == "auto" and GradSync.sparse_rows_pay(I, cap_rows, world)):
            from svae_cf_amd.layout import E as _E
            eng.grad_sync.enable_sparse_rows(2, I, _E, capacity=cap_rows)
        # ... and the decoder.linear2 bucket (151 I floats, dense by nature) as reduce-scatter + sharded optimizer step
        # + all-gather: the same bytes on the links, the optimizer's 28 B per parameter once per job, not per rank
        if args.shard_w3 == "on" or (args.shard_w3 == "auto" and world > 1 and I >= 100000):
            eng.grad_sync.enable_sharded(0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or rehearse:
            dist.barrier()
        torch.cuda.synchronize()

    # Device settle phase (disclosed in the JSON line as "settle"): an idle GPU needs ~40 ms of load before it runs at its
    # steady rate -- with the driver's --warmup 5 the first timed region measured 0.970 ms per step, with --warmup 16
    # 0.937, with --warmup 40 0.902, the later regions of all three 0.907-0.915 (one box, one call).  The workload's
    # metric is sustained training throughput, so --settle-steps untimed steps over the same plans run first; the W
    # warm-up steps and the K timed steps then follow exactly as asked, on the same plans as without it.
    # A COUNT of steps, the same on every rank (each step issues the data-parallel collectives: a time-based loop
    # would leave the ranks with different numbers of them).
    # ... and the figure WITHOUT that phase is printed beside `value` (`value_without_settle`): the same W warm-up and K
    # timed steps run first, on the idle GPU, exactly as a caller who passes --settle-steps 0 would get them.
    cold = None
    if args.settle_steps > 0 and not args.no_cold_figure:
        for i in range(args.warmup):
            eng.train_step(plans[i % len(plans)], 0.01, seed=1234)
        barrier()
        t0c = time.perf_counter()
        uc = 0
        for i in range(args.steps):
            p_ = plans[(args.warmup + i) % len(plans)]
            eng.train_step(p_, 0.01, seed=1234)
            uc += p_.B
        barrier()
        dtc = time.perf_counter() - t0c
        tc = torch.tensor([dtc, float(uc)], dtype=torch.float64, device=dev)
        if world > 1 or rehearse:
            tmax = tc[:1].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            usum = tc[1:].clone()
            dist.all_reduce(usum, op=dist.ReduceOp.SUM)
            dtc, uc = float(tmax[0]), float(usum[0])
        cold = dict(value=uc / dtc, ms_per_step=dtc / args.steps * 1e3,
                    note="the same --warmup / --steps on the idle GPU, before the settle phase")
    settle_steps = max(0, int(args.settle_steps))
    t_s = time.perf_counter()
    for i in range(settle_steps):
        eng.train_step(plans[i % len(plans)], 0.01, seed=1234)
    torch.cuda.synchronize()
    settle_ms = (time.perf_counter() - t_s) * 1e3
    step = 0
    for i in range(args.warmup):
        eng.train_step(plans[step % len(plans)], 0.01, seed=1234)
        step += 1
    # The timed region carries NO events.  A HIP event recorded between queued kernels (or attached to a dispatch)
    # stalls the asynchronous launch pipeline on this stack: 150-600 us of host time per record once kernels are in
    # flight, and with the two sweeps bracketed on every 4th step some processes measured 130k seq/s where the
    # unbracketed loop gives 256k (bimodal: the same command, run three times on one box, gave 249k / 136k / 130k).
    # Kernel durations are therefore measured with HIP events on the launch streams in a REPLAY of the timed steps
    # (same plans, same order, same process) right after the timed region; the sweeps use start / stop events attached
    # to the dispatch itself (hipExtLaunchKernelGGL: the kernel's own timestamps, what rocprofv3 reports).
    # Only the kernels whose bench-side timing agrees with the rocprofv3 trace of the same command to 5 % are kept (the
    # two sweeps: dispatch-attached timestamps; the recurrences: long kernels, the brackets' cost is small beside them).
    # The event-bracketed short kernels read 10-45 % off (profiles/r02p_bench_vs_rocprof.md: wgrad 1.32, mlp_fwd 1.46,
    # gi_gemm 0.66): for those the kernel trace under profiles/ is the record, not this line.
    tags_all = ["proj_fwd", "proj_fdh_fused", "proj_dw_fused", "proj_dw", "proj_dx", "softmax_nll", "gru_fwd", "gru_fwd2",
                "gru_bwd"]
    eng.prof = None
    first_step = step

    def timed_region(step):
        """EXACTLY args.steps steps between barrier + synchronize on both sides."""
        barrier()
        t0 = time.perf_counter()
        users_done = rows_done = 0
        for i in range(args.steps):
            p = plans[step % len(plans)]
            eng.train_step(p, 0.01, seed=1234)
            users_done += p.B
            rows_done += p.R
            step += 1
        t_enq = time.perf_counter() - t0
        barrier()
        return time.perf_counter() - t0, t_enq, users_done, rows_done, step

    dt, t_enq, users_done, rows_done, step = timed_region(step)
    loss = float(eng.loss_out[0])
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    u = torch.tensor([float(users_done)], dtype=torch.float64, device=dev)
    dt_ranks = [dt]
    if world > 1 or rehearse:
        gathered = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, t)
        dt_ranks = [float(g_[0]) for g_ in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    dt, total_users = float(t[0]), float(u[0])
    # the same region again (same plans: the step counter continues through the cycle): spread of the measurement
    region_ms = [dt / args.steps * 1e3]
    for _ in range(max(0, args.repeats - 1)):
        d2, _, _, _, step = timed_region(step)
        t2 = torch.tensor([d2], dtype=torch.float64, device=dev)
        if world > 1 or rehearse:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        region_ms.append(float(t2[0]) / args.steps * 1e3)

    # (per STEP: a kernel launched twice in a step -- the forward sweep and the recurrence run in two time ranges --
    # counts with the sum of its launches)
    def harvest(nsteps):
        out = {}
        for tag, evs in (eng.prof or {}).items():
            if evs and nsteps:
                out[tag] = float(np.sum([a.elapsed_time(b) for a, b in evs])) / nsteps     # ms
        if "gru_fwd2" in out:
            out["gru_fwd"] = out.get("gru_fwd", 0.0) + out.pop("gru_fwd2")
        return out
    kt, replay_steps = {}, 0
    if not args.no_kernel_times:
        replay_steps = min(args.steps, 48)
        eng.prof = {t: [] for t in tags_all}
        for i in range(replay_steps):
            eng.train_step(plans[(first_step + i) % len(plans)], 0.01, seed=1234)
        torch.cuda.synchronize()
        kt = harvest(replay_steps)
    rows_bracketed = [plans[(first_step + i) % len(plans)].R for i in range(replay_steps)]
    rows_extra, kt_extra = [], {}
    eng.prof = None
    # The sweeps ALONE: in the step they share the GPU on purpose (the forward sweep of the first time range runs on
    # 192 CUs beside the second recurrence, the weight-gradient sweep as 158 long workgroups beside the BPTT), which
    # is the step's optimum and not the kernel's.  One more pass over the plans with every stream folded into one, the
    # forward in one range, the weight-gradient sweep split to fill the idle GPU in one round: the same kernels with the GPU to themselves.
    solo = None
    if kt and eng.fused and "proj_fdh_fused" in kt and args.prec == "bf16" and world == 1 and not rehearse:
        saved = (eng.overlap, eng.time_split, eng.dw_q, eng.fdh_q0)
        saved_wgs = eng.dw_wgs
        eng.dw_wgs = eng._ncu                       # (flat work division: one range per CU of the idle GPU)
        # (weight-gradient sweep 3-way: 237 workgroups, one round on the idle GPU -- 4-way is 316, two rounds, 0.279 ms
        # against 0.209; the forward sweep keeps its nominal 4-way split: nominal 2 leaves unequal workgroups, 0.227
        # against 0.216)
        eng.overlap, eng.time_split, eng.dw_q = False, False, 3
        try:
            for p_ in plans:
                eng.prepare(p_)
            n_solo = min(replay_steps, len(plans))
            for i in range(2):
                eng.train_step(plans[(first_step + i) % len(plans)], 0.01, seed=1234)
            torch.cuda.synchronize()
            eng.prof = {"proj_fdh_fused": [], "proj_dw_fused": []}
            for i in range(n_solo):
                eng.train_step(plans[(first_step + i) % len(plans)], 0.01, seed=1234)
            torch.cuda.synchronize()
            ks = harvest(n_solo)
            rows_solo = float(np.mean([plans[(first_step + i) % len(plans)].R for i in range(n_solo)]))
            if "proj_fdh_fused" in ks and "proj_dw_fused" in ks:
                ms_s = ks["proj_fdh_fused"] + ks["proj_dw_fused"]
                fl = 300.0 * I * rows_solo
                solo = dict(avg_ms=ms_s, fdh_ms=ks["proj_fdh_fused"], dw_ms=ks["proj_dw_fused"],
                            achieved=3 * fl / (ms_s * 1e-3) / 1e12, frac=3 * fl / (ms_s * 1e-3) / 1e12 / PEAK[args.prec],
                            executed_frac=4 * fl / (ms_s * 1e-3) / 1e12 / PEAK[args.prec], steps=n_solo,
                            note="the two sweeps with nothing beside them (one stream, forward in one range, "
                                 "weight-gradient sweep as one round of 237 workgroups): the kernel's own rate; `frac` above is its rate inside "
                                 "the overlapped step")
        finally:
            eng.prof = None
            eng.overlap, eng.time_split, eng.dw_q, eng.fdh_q0 = saved
            eng.dw_wgs = saved_wgs
            torch.cuda.synchronize()
    kt_all = dict(kt_extra)
    kt_all.update(kt)
    rows_step = rows_done / args.steps
    # Algorithmic work per launch: one projection-sized contraction = 300*I flop per row; each
    # fused kernel does two (forward+dL/dh sweep: logits and P W3; weight-gradient sweep: logit recompute and
    # P^T h); the GRU recurrence 2*200*600 flop per row (forward), the same count in backward (dgh W_hh).
    # a kernel's work is priced on the rows of the steps it was bracketed in (sampled steps of the timed region;
    # the 6 extra steps for the others)
    rows_timed = float(np.mean(rows_bracketed)) if rows_bracketed else rows_step
    rows_untimed = float(np.mean(rows_extra)) if rows_extra else rows_step
    proj_flop = 300.0 * I * rows_timed
    def work_of(tag):
        rows = rows_timed if tag in kt else rows_untimed
        pf = 300.0 * I * rows
        return {
            # per kernel: forward+dL/dh sweep = 2 algorithmic contractions; weight-gradient sweep = 1 (its logit
            # recompute is executed, not algorithmic work)
            "proj_fwd": ("mfma", pf), "proj_fdh_fused": ("mfma", 2 * pf), "proj_dw_fused": ("mfma", pf),
            "proj_dw": ("mfma", pf), "proj_dx": ("mfma", pf),
            "gru_fwd": ("mfma", 240000.0 * rows), "gru_bwd": ("mfma", 240000.0 * rows),
            "softmax_nll": ("hbm", rows * I * 4 * 2 + rows * 4 * 4),
        }.get(tag)
    rooflines = {}
    for tag, ms in kt_all.items():
        if work_of(tag) is None:
            continue
        bound, amount = work_of(tag)
        if bound == "mfma":
            ach, peak, unit = amount / (ms * 1e-3) / 1e12, PEAK[args.prec], "TFLOP/s"
        else:
            ach, peak, unit = amount / (ms * 1e-3) / 1e9, PEAK["hbm"], "GB/s"
        rooflines[tag] = dict(bound=bound, kernel=tag, achieved=ach, peak=peak, unit=unit, frac=ach / peak,
                              traffic=None, avg_ms=ms)
    # HBM bytes per launch from the PMC passes (tools/gpu_pmc.sh + tools/pmc_traffic.py: FETCH_SIZE doubled, the
    # gfx950 correction of MI355X_MICROARCH.md; separate --pmc runs), committed under profiles/.
    pmc, pmc_name = {}, None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc_proj.json")))
        if cands:
            pmc_name = os.path.basename(cands[-1])          # the newest record tag (names sort by round and letter)
            with open(cands[-1]) as f:
                pmc = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        pmc = {}
    if args.shape != "ml-20m" or I != 20000:
        pmc = {}                                        # the PMC passes were taken on the default workload only
    for tag, r in rooflines.items():
        if tag in pmc and "hbm_bytes" in pmc[tag]:      # per STEP, like avg_ms: bytes per launch x launches per step
            r["traffic"] = pmc[tag]["hbm_bytes"] * pmc[tag].get("launches_per_step", 1.0)
    # The dominant kernel is the projection sweep, proj_kernel<MODE> (csrc/proj.hip): launched twice per step
    # (forward + dL/dh sweep, dL/dW sweep), together the largest share of the step.  Its line covers both launches.
    # ALGORITHMIC work: the projection is 300*I flop per row forward, x3 for training = three
    # projection-sized contractions per step (logits, dL/dh2 = P W3, dL/dW3 = P^T h2).  The kernels EXECUTE four: the
    # weight-gradient sweep recomputes the logits instead of reading them back (they are never materialised); that
    # figure is reported under "executed", it is not the roofline fraction.
    # The GRU recurrence kernels are listed in "rooflines" for completeness: they are bound by per-timestep latency.
    roofline = None
    if "proj_fdh_fused" in kt and "proj_dw_fused" in kt:
        ms = kt["proj_fdh_fused"] + kt["proj_dw_fused"]
        ach = 3 * proj_flop / (ms * 1e-3) / 1e12
        exe = 4 * proj_flop / (ms * 1e-3) / 1e12
        tr = None
        if "traffic" in rooflines["proj_fdh_fused"] and rooflines["proj_fdh_fused"]["traffic"] and \
                rooflines["proj_dw_fused"]["traffic"]:
            tr = rooflines["proj_fdh_fused"]["traffic"] + rooflines["proj_dw_fused"]["traffic"]
        roofline = dict(bound="mfma", kernel="proj_kernel (fdh + dw sweeps of one step)", achieved=ach,
                        peak=PEAK[args.prec], unit="TFLOP/s", frac=ach / PEAK[args.prec], traffic=tr, avg_ms=ms,
                        algorithmic_flop=3 * proj_flop, traffic_source=("profiles/%s" % pmc_name) if tr else None,
                        executed=dict(achieved=exe, frac=exe / PEAK[args.prec], flop=4 * proj_flop,
                                      note="incl. the logit recompute of the weight-gradient sweep"))
        if solo is not None:
            roofline["alone"] = solo
    elif rooflines:
        roofline = rooflines[max(rooflines, key=lambda k: rooflines[k]["avg_ms"])]

    # Scoring leg (evaluate()'s per-user work, nb:615-636): forward without gradients, the last timestep's logits
    # dense, seen items masked, top-100.  The ranking kernel reads the [users, I] scores once: HBM-bound.
    ev_leg = None
    if rank == 0 and (args.eval_leg or args.shape == "large"):
        from svae_cf_amd import _lib as L
        p0 = plans[0]
        for _ in range(2):
            eng.score(p0, topk=100, seed=3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nrep = 5
        e0.record()
        for _ in range(nrep):
            _, idx_k, val_k, last = eng.score(p0, topk=100, seed=3)
        e1.record()
        torch.cuda.synchronize()
        score_ms = e0.elapsed_time(e1) / nrep
        work = torch.empty(p0.B * eng.IP, device=eng.device, dtype=torch.float32)
        st = torch.cuda.current_stream().cuda_stream
        e0.record()
        for _ in range(nrep):
            L.call("svae_masked_topk", last.data_ptr(), eng.IP, p0.seen.data_ptr(), p0.seen_off.data_ptr(), p0.B, I, 100,
                   idx_k.data_ptr(), val_k.data_ptr(), work.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        topk_ms = e0.elapsed_time(e1) / nrep
        tb = float(p0.B) * I * 4
        # the way notebook_api.evaluate scores: one scoring session, batches alternating between the engine's two
        # workspaces on two streams
        cur_s = torch.cuda.current_stream()
        some = plans[:8]
        lanes_ms = None
        lanes = eng.scoring_lanes(3)
        with eng.scoring_session(sweep_cus=None if len(lanes) == 1 else eng.two_lane_sweep_cus(p0.B)):
            if len(lanes) > 1:
                for s_ in lanes[1:]:
                    s_.wait_stream(cur_s)
                for timed in (False, True):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for k_, p_ in enumerate(some * (3 if timed else 1)):
                        with torch.cuda.stream(lanes[k_ % len(lanes)]):
                            eng.score(p_, topk=100, seed=3, slot=k_ % len(lanes))
                    for s_ in lanes[1:]:
                        cur_s.wait_stream(s_)
                    torch.cuda.synchronize()
                    lanes_ms = (time.perf_counter() - t0) * 1e3 / (3 * len(some))
        ev_leg = dict(users=p0.B, rows=p0.R, score_ms=score_ms, users_per_s=p0.B / (score_ms * 1e-3),
                      lanes=(None if lanes_ms is None else dict(
                          ms_per_batch=lanes_ms, users_per_s=sum(p_.B for p_ in some) / len(some) / (lanes_ms * 1e-3),
                          note="scoring_session(): %d batches in flight on %d streams, 8 plans x 3" % (len(lanes), len(lanes)))),
                      topk=dict(kernel="svae_masked_topk (k=100)", ms=topk_ms, algorithmic_bytes=tb, bound="hbm",
                                achieved=tb / (topk_ms * 1e-3) / 1e9, peak=PEAK["hbm"], unit="GB/s",
                                frac=tb / (topk_ms * 1e-3) / 1e9 / PEAK["hbm"]))

    # The same step at the REFERENCE's precision (the notebook computes in fp32): PREC_SPLIT runs the timed kernel
    # templates with every operand as bf16 hi + mid + lo parts (fp32-class results: the parity gates run through this
    # mode).  A few steps over the same plans, same parameters; reported beside `value`, never as `value`.
    fp32_class = None
    if rank == 0 and world == 1 and not rehearse and args.prec == "bf16" and not args.no_fp32_class:
        torch.cuda.synchronize()
        eng2 = S.SVAEEngine(I, device=dev, prec=S.PREC_SPLIT)
        eng2.load_state_dict(sd)
        eng2.reserve(max(p_.R for p_ in plans))
        for p_ in plans:
            eng2.prepare(p_)
        n2w, n2 = 3, 8
        for i in range(n2w):
            eng2.train_step(plans[i % len(plans)], 0.01, seed=1234)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        u2 = 0
        for i in range(n2):
            p_ = plans[(n2w + i) % len(plans)]
            eng2.train_step(p_, 0.01, seed=1234)
            u2 += p_.B
        torch.cuda.synchronize()
        d2 = time.perf_counter() - t0
        fp32_class = dict(mode="split", value=u2 / d2, unit="sequences/s", ms_per_step=d2 / n2 * 1e3, steps=n2, warmup=n2w,
                          final_loss=float(eng2.loss_out[0]),
                          note="PREC_SPLIT: the timed kernel templates with operands as bf16 hi+mid+lo parts and six "
                               "products per MFMA (fp32-class: loss 1e-5, gradients and parameters 1e-4 against the "
                               "reference, tests/test_gpu_split_parity.py); the reference itself computes in fp32")
        del eng2
        torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import svae_torch as TP           # checker / baseline leg only
        cb = TP.time_cpu_baseline(users[:1200], I, next_k=4, warmup=3, budget_s=20.0, max_users=1000)
        cpu = dict(value=cb["value"], unit="sequences/s", cores=cb["cores"], kind="port",
                   sample="%d users of the same synthetic %s-shaped set, notebook-style B=1 fwd+bwd+Adam "
                          "(dense fp32, torch CPU), %.1f s" % (cb["users"], shape_name, cb["seconds"]))

    if rank == 0:
        out = {
            "metric": "train sequences/sec (%s shape)" % shape_name, "value": total_users / dt, "unit": "sequences/s",
            "n_gpus": world, "ranks_in_process_group": (dist.get_world_size() if (world > 1 or rehearse) else 1),
            "steps": args.steps, "warmup": args.warmup,
            "value_without_settle": cold,
            "settle": {"steps": settle_steps, "ms": settle_ms,
                       "note": "untimed steps before the warm-up: an idle GPU needs ~40 ms of load to reach its steady rate"},
            "ms_per_step": dt / args.steps * 1e3, "ms_per_step_by_rank": [d / args.steps * 1e3 for d in dt_ranks],
            "ms_per_step_regions": dict(n=len(region_ms), min=float(np.min(region_ms)),
                                        median=float(np.median(region_ms)), max=float(np.max(region_ms)),
                                        note="the timed region repeated; `value` is the first one"), "host_enqueue_ms_per_step": t_enq / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
            "config": {"workload": "%s-shaped synthetic (configs[%d]): %d items, seq<=%d, "
                                   "length-bucketed batch %d users/GPU, next_k=4, fwd+bwd+allreduce+Adam"
                                   % (shape_name, shape_cfg, I, lmax, B),
                       "items": I, "batch_per_gpu": B, "mean_rows_per_step": rows, "mean_rows_bracketed_steps": rows_timed, "parallelism": "dp%d" % world, "embedding_gradient_exchange": ("rows" if (eng.grad_sync is not None and eng.grad_sync.sparse is not None) else "dense"),
                       "caller_stream": ("high priority (SVAEEngine.training_stream, as train_batched)" if os.environ.get("SVAE_TRAIN_HIPRIO", "1") == "1" else "default"),
                       "collectives": ("library RCCL wrappers on the producing streams" if (eng.grad_sync is not None and eng.grad_sync.native) else ("torch.distributed" if eng.grad_sync is not None else None)),
                       "decoder_linear2_exchange": ("reduce-scatter + sharded optimizer + all-gather" if (eng.grad_sync is not None and eng.grad_sync.sharded is not None) else "all-reduce"),
                       "final_loss": loss, "side_stream": eng.side_stream_info},
            "kernel_ms": kt, "kernel_timing": "HIP events on the launch streams over a replay of the first %d timed steps "
                                             "(events inside the timed region stall the launch pipeline)" % replay_steps, "roofline": roofline, "rooflines": rooflines,
            "cpu_baseline": cpu, "fp32_class": fp32_class,
        }
        if ev_leg is not None:
            out["eval_leg"] = ev_leg
        print(json.dumps(out))
    if world > 1 or rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
