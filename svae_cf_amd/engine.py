"""SVAE step engine: owns the flat parameter / gradient / Adam buffers and the activation
workspace in HBM and drives the HIP kernels through the C-ABI (``_lib``).

One engine per process (= per GPU).  Every launch goes to ``torch.cuda.current_stream()``;
nothing here allocates inside a step once the workspace has reached its high-water mark,
and nothing synchronises with the host (the loss stays on the device until asked for).

Reference path replaced: Model.forward nb:752-768, VAELoss.forward nb:795-812,
loss.backward() nb:1205, optimizer.step() nb:1206, scoring in evaluate() nb:615-636.
"""
import contextlib
import math
import os
import threading

import torch

from . import _lib
from .layout import D, DP, E, G, HP, L, ParamLayout

PREC_FP32 = 0
PREC_BF16 = 1
# The fused bf16 kernels in their SPLIT instantiation: operands as bf16 hi + mid + lo parts (all 24 significant bits),
# products by their six leading terms (fp32-class results from the same kernel templates, work division, slab /
# finish structure as PREC_BF16: the projection sweeps, the MLP chain, the row-panel and weight-gradient GEMMs, and --
# since round 3 -- the persistent GRU recurrence, forward and BPTT, with exact sigmoid / tanh).  The parity gates run
# through it, on the same host path as the timed step (two time ranges, merged gate-gradient array, side streams); it
# does six times the MFMA work of PREC_BF16 and is only timed for the bench line's `fp32_class` figure.
PREC_SPLIT = 2


# Launch stream of the kernels.  torch.cuda.current_stream() costs ~8 us of host time (device-index and
# environment look-ups) and the step asked for it 33 times; forward / backward / adam_step now ask once and pin
# the raw pointer here, `_Launch` switches it (and, only when torch ops or collectives must follow, torch's own
# current stream) for the side-stream sections.
_PINNED = [None]
_PINNED_OBJ = [None]


_POOL_LOCK = threading.Lock()
# Streams are created once per process and device and shared by every engine (see SVAEEngine._side_stream):
_TRAIN_STREAMS = {}      # device index -> the high-priority stream of the training loop
_STREAM_POOL = {}        # (device index, priority) -> candidate side streams, at most 8
_SIDE_CHOICE = {}        # (device index, priority, streams to run beside) -> (chosen stream, measurement)


def _stream():
    p = _PINNED[0]
    return p if p is not None else torch.cuda.current_stream().cuda_stream


class _Launch:
    """with _Launch(stream, also_torch): kernels launched through ``_stream()`` go to `stream`."""

    def __init__(self, stream, also_torch=False):
        self.stream = stream
        self.ctx = torch.cuda.stream(stream) if also_torch else None

    def __enter__(self):
        self.prev, self.prev_obj = _PINNED[0], _PINNED_OBJ[0]
        _PINNED[0], _PINNED_OBJ[0] = self.stream.cuda_stream, self.stream
        if self.ctx is not None:
            self.ctx.__enter__()
        return self.stream

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        _PINNED[0], _PINNED_OBJ[0] = self.prev, self.prev_obj
        return False


# Tuning switches of the step's order and work division.  Every default is the measured optimum (the comments beside
# the attributes say what was measured); the environment is consulted ONLY when SVAE_TUNING=1 is set as well -- the A/B
# tools under tools/ set it -- so that a stray SVAE_* variable in a training environment cannot steer the product path.
# ``python -c "import svae_cf_amd.engine as e; print(e.ENGINE_OPTIONS)"`` after constructing an engine lists them.
ENGINE_OPTIONS = {}
_TUNING = os.environ.get("SVAE_TUNING", "0") == "1"


def _opt(name, default):
    ENGINE_OPTIONS[name] = default
    return os.environ.get(name, default) if _TUNING else default


class SVAEEngine:
    def __init__(self, total_items, device="cuda", prec=PREC_FP32, lr=1e-3, betas=(0.9, 0.999),
                 adam_eps=1e-8, weight_decay=5e-3):
        if not torch.cuda.is_available():
            raise RuntimeError("SVAEEngine needs a GPU: the HIP path has no CPU fallback")
        _lib.load()
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            _lib.call("svae_check_device")
        self._ncu = int(_opt("SVAE_PROJ_CUS", torch.cuda.get_device_properties(self.device).multi_processor_count))
        self.layout = ParamLayout(total_items)
        self.I, self.IP = self.layout.I, self.layout.IP
        self.prec = prec
        self.split = prec == PREC_SPLIT
        # PREC_SPLIT runs the recurrence through the split instantiation of the TIMED kernel template (gru.hip, prec 2)
        # since round 3; SVAE_SPLIT_GRU_F32=1 restores the generic fp32 kernels of rounds 1-2 in that mode
        self.gru_prec = PREC_FP32 if (self.split and _opt("SVAE_SPLIT_GRU_F32", "0") == "1") else prec
        self.gemm_prec = PREC_FP32 if self.split else prec         # generic GEMM (dense surface): fp32 or bf16
        self.lr, self.betas, self.adam_eps, self.weight_decay = lr, betas, adam_eps, weight_decay
        n = self.layout.total
        f32 = dict(device=self.device, dtype=torch.float32)
        self.params = torch.zeros(n, **f32)
        self.grads = torch.zeros(n, **f32)
        self.exp_avg = torch.zeros(n, **f32)
        self.exp_avg_sq = torch.zeros(n, **f32)
        self.step_count = 0
        # which optimizer optimizer.step() runs (nb:1252-1267): "adam" (the default the reference trains with), or
        # (kind, decay, eps) of svae_optimizer_step for adagrad / adadelta / rmsprop -- set by set_optimizer()
        self.optimizer = "adam"
        self._opt = None
        self._pviews, self._gviews = {}, {}
        self._gbase = self.grads.data_ptr()
        self.whh_t = torch.zeros(HP * G, **f32)
        self.loss_out = torch.zeros(4, **f32)
        self.cap = 0
        self.ws = {}
        # scoring sessions (scoring_session / score(slot=1)): a second, forward-only workspace so that two batches can be
        # scored beside each other on two streams; `_session` promises that the parameters do not change
        self._session = False
        self._session_ready = None
        self._slot = 0
        self.score_cus = None          # workgroups of the loss-only sweep of `score` (None: the CU count); see scoring_session
        self._slot_main = None         # (ws, loss_out) of slot 0 while slot 1 is current
        self._slots = {}               # slot (1, 2) -> (ws, loss_out, cap)
        # row splits of the weight-gradient GEMMs (end of round 2, tools/sweep_env.sh, ms per step on one box, two
        # rounds: 32 -> 0.868 / 0.870, 16 -> 0.867 / 0.870, 24 -> 0.868 / 0.869, 21 -> 0.881 / 0.879, 48 -> 0.883 / 0.881)
        self.nsplit = int(_opt("SVAE_NSPLIT", "32"))
        # The GRU's two weight gradients on 256 x 256 tiles (svae_wgrad_pair_big): 36 us against 48 alone at 21.8k rows, but
        # NOT in the step -- 0.8475 against 0.8460 ms, two alternating pairs: its workgroups take a whole CU (128 KB of
        # LDS, one wave per SIMD) and wait for the CUs the weight-gradient sweep holds, where the 128 x 128 kernel's
        # fit in beside the sweep's.  Off; kept as a library entry point with its test (tests/test_gpu_wgrad.py).
        self.wgrad_big = _opt("SVAE_WGRAD_BIG", "0") == "1"
        # The GRU's two weight gradients on the all-bf16 128 x 128 kernel with hand-counted waits (svae_wgrad_pair_bf16;
        # its recurrent-weight job reads a bf16 copy of the previous hidden states made beside the BPTT): 31 us against 48
        # alone, weight gradients bit-identical
        self.wgrad16 = _opt("SVAE_WGRAD16", "1") == "1"
        self.nsplit_big = 40                                       # (6 output tiles x 40 row splits = 240 workgroups)
        self.slabs = torch.zeros(max(self.nsplit, self.nsplit_big) * (G * E + G), **f32)        # largest split-K output: dW_ih
        self.slabs_side = torch.zeros(self.nsplit * (DP * HP + DP), **f32)  # side-stream weight gradients
        # dW_hh's own slabs: the paired launch of the two GRU weight gradients (and the two-range BPTT) keep them
        # pending beside dW_ih's
        self.slabs_hh = torch.zeros(max(self.nsplit, self.nsplit_big) * (G * HP + G), **f32)
        self.dh_carry = None
        # Backward runs on two streams: the data-gradient chain that leads to the GRU's BPTT (a 16-64 CU kernel,
        # latency bound) stays on the caller's stream, the decoder-side weight gradients (whole-GPU kernels that
        # nothing downstream waits for) go to a side stream and fill the CUs the recurrence leaves idle.
        self.overlap = True
        self._sides = []
        self._sides_by_caller = {}     # caller's stream -> the side streams chosen to run beside it
        self._side_for = None
        self.side_stream_info = []
        self.colparts = torch.zeros(64 * max(self.IP, G), **f32)
        self._plan = None
        self._anneal = 0.0
        # fused projection + log-softmax + NLL (never materialises logits): bf16 MFMA mode only
        self.fused = prec in (PREC_BF16, PREC_SPLIT)
        self.IPAD = (self.I + 63) // 64 * 64
        self.w3b = self.w3tb = None
        self._last_fused = False
        self._fused_grad = False
        self._prepped_grad = False
        self._dtmp_extra = None
        self._loss_deferred = False
        self._train_stream = None
        self.wgrad_pair = _opt("SVAE_WGRAD_PAIR", "1") == "1"   # the two GRU weight gradients in one launch
        # stream priorities of the side stream and of the decoder.linear2 stream ("a,b"; lower number = higher priority,
        # 0 = the default, the caller's stream is whatever the caller made it)
        self.side_prio = [int(x) for x in _opt("SVAE_SIDE_PRIO", "0,0").split(",")]
        self.time_split = _opt("SVAE_TIME_SPLIT", "1") == "1"
        self.rowgemm = _opt("SVAE_ROWGEMM", "1") == "1"
        self.merged_dg = _opt("SVAE_MERGED_DG", "1") == "1"
        self.dg_bf16 = _opt("SVAE_DG_BF16", "1") == "1"
        # train_step only: the decoder.linear2 gradient work (weight-gradient sweep, its finish, the bucket's
        # all-reduce, Adam over the bucket, the bf16 operand copies of the new weights) runs on a stream of its own
        # and the step does not wait for it -- it finishes under the next step's embedding / input GEMM / first
        # recurrence range, which leave most of the GPU idle, and the next step waits for it before its decoder side
        self.float_w3 = _opt("SVAE_FLOAT_W3", "1") == "1"
        self.fused_prep = _opt("SVAE_FUSED_PREP", "1") == "1" and not self.split
        self._in_step = False
        self._float_pending = False
        self._float_plan = None
        self._w3_event = None          # recorded on the w3 stream behind the floated work of the last train_step
        self._w3_fresh = False         # w3b / w3s / w3tb match the current parameters (made by the floated work)
        self.tail_embedding_first = _opt("SVAE_TAIL", "emb") == "emb"
        self.gru_wgrad_side = _opt("SVAE_GRU_WGRAD_SIDE", "1") == "1"
        self.gi_split = _opt("SVAE_GI_SPLIT", "1") == "1"
        self.bptt_split = _opt("SVAE_BPTT_SPLIT", "0") == "1"
        # (both split counts re-swept on the round's final step, tools/sweep_env.sh, ms per step on one box, two rounds:
        #  forward 3 -> 0.864, 4 -> 0.864, 5 -> 0.870, 6 -> 0.872, 8 -> 0.883; weight gradient 2 -> 0.866, 3 -> 0.924,
        #  4 -> 0.911)
        self.fdh_q0 = int(_opt("SVAE_FDH_Q0", "4"))  # nominal split count of the forward sweep (_proj_sched)
        self.stamps = None             # tools/step_stamps.py: dict(buf, names, step)
        self._sevs = {}                # purpose -> _lib.SyncEvent, re-recorded every step
        self._sched_cache = {}         # tile-count shape -> (device table, nwg, ns_q, ns_rem), see _proj_sched
        self.dw_q = int(_opt("SVAE_DW_Q", "2"))      # split count of the weight-gradient sweep (_proj_sched)
        # flat work division of the two sweeps (round 3, bf16 instantiation): equal-cost contiguous ranges of the
        # (x tile, y tile) space, one per workgroup; SVAE_FLAT_SCHED=0 restores the balanced q / q+1 splits
        self.flat_sched = _opt("SVAE_FLAT_SCHED", "1") == "1"
        self.flat_boundary = int(_opt("SVAE_FLAT_BOUNDARY", "5"))   # cost of a further segment, in tiles
        # workgroups of the weight-gradient sweep: it runs beside the BPTT (64 CUs) and the weight-gradient GEMMs, and
        # what it leaves them decides the step (one box, ms per step, two rounds: 144 -> 0.863 / 0.864, 158 -> 0.851 /
        # 0.855, 170 -> 0.922 / 0.921, 192 -> 0.932 / 0.930; balanced 2-way split = 158 workgroups: 0.859 / 0.858)
        # Larger catalogues take more (_dw_workgroups: 192 above 51 200 items, the CU count less 32 above 100 000;
        # 1M items: 158 -> 14.70, 224 -> 13.36 ms per step).  ML-1M shape: 96 .. 158 within 0.3 %.
        self.dw_wgs = int(_opt("SVAE_DW_WGS", "158"))
        self.range1_cus = int(_opt("SVAE_RANGE1_CUS", "0"))           # tuning: workgroups of the first range's forward sweep
        self.dw_wgs_exact = _opt("SVAE_DW_WGS_EXACT", "0") == "1"     # tuning: dw_wgs as given, whatever the shape
        # the weight-gradient sweep starts behind the forward sweep (not behind the MLP backward): from ~4 000 items on --
        # ML-20M shape with 6 000 / 10 000 / 14 000 / 20 000 items: -0.2 / -1.2 / -0.5 / -1.2 %; at the ML-1M shape (3 483
        # items) the sweep is 70 us and starting it beside the MLP backward costs 0.9 % (0.7344 against 0.7281 ms)
        self.dw_early = _opt("SVAE_DW_EARLY", "1" if self.I >= 4096 else "0") == "1"
        self.head_fused = _opt("SVAE_HEAD_FUSED", "1") == "1"
        # bf16 mode: the gathered embedding rows are kept as bf16 (their two consumers -- the input-gate GEMM and the
        # input-weight gradient -- round them to bf16 anyway: bit-identical) -- half the bytes written by the gather,
        # read by the GEMM and re-read six times from L2 by the weight-gradient kernel
        self.emb16 = _opt("SVAE_EMB16", "1") == "1"
        self.mlp2_side = _opt("SVAE_MLP2_SIDE", "1") == "1"
        self.early_embed_opt = _opt("SVAE_EARLY_EMBED_OPT", "0") == "1"
        self._early_opt = None         # train_step -> backward: (range, gscale, step number) of the embedding bucket
        self.grad_sync = None          # dist.GradSync for data parallel runs
        self.event_pool = []
        self.timer_pool = []
        self.prof = None               # dict tag -> list of (start, end) torch.cuda.Event pairs
        self._pin_pool = None
        self.generation = 0            # counts forwards: the workspace holds the activations of the LAST one only

    # ------------------------------------------------------------------ parameters
    def P(self, name):
        v = self._pviews.get(name)
        if v is None:
            v = self._pviews[name] = self.layout.view(self.params, name)
        return v

    def Gv(self, name):
        v = self._gviews.get(name)
        if v is None:
            v = self._gviews[name] = self.layout.view(self.grads, name)
        return v

    def _torch_ctx(self):
        """Side-stream sections must also switch torch's current stream when collectives are issued from inside
        them; otherwise the raw launch pointer is enough (and much cheaper)."""
        return self.grad_sync is not None

    def _settle(self):
        """Order the caller's stream behind work a train_step left floating on the w3 stream."""
        if self._w3_event is not None:
            _lib.stream_wait(torch.cuda.current_stream(), self._w3_event)
            self._w3_event = None
            self._float_plan = None

    def load_state_dict(self, sd):
        self._settle()
        self._w3_fresh = False
        self.layout.pack(sd, self.params)

    def state_dict(self):
        self._settle()
        return self.layout.unpack(self.params)

    def grad_dict(self):
        self._settle()
        return self.layout.unpack(self.grads)

    def reset_optimizer(self):
        self._settle()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_count = 0

    # ------------------------------------------------------------------ workspace
    def reserve(self, max_rows):
        """Size the workspace for batches of up to `max_rows` rows now.  It otherwise grows on demand, and growing is a
        synchronisation plus a reallocation of every buffer in the middle of a run (seen in the driver's 20-step bench
        of round 1: plans first met inside the timed region)."""
        self._ensure(int(max_rows))

    def _ensure(self, R):
        if R <= self.cap:
            return
        if self._slot != 0:
            raise RuntimeError("the workspace grows on slot 0 only (score() sizes it before it switches slots)")
        if self._w3_event is not None:     # floated kernels still read the workspace that is about to be replaced
            self._settle()
            torch.cuda.current_stream().synchronize()
        if self._session:                  # another slot's kernels may still be running on another stream
            torch.cuda.synchronize(self.device)
        cap = int(R * 1.25) + 64
        self.ws = self._new_ws(cap)
        self.cap = cap
        if self.fused:
            self.capp = (cap + 63) // 64 * 64
            if self.w3b is None:
                bf = dict(device=self.device, dtype=torch.bfloat16)
                m = 3 if self.split else 1
                self.wih_frags = [torch.zeros(m * _lib.load().svae_rowgemm_frag_elems(i), **bf) for i in (0, 1)]
                self.mlp_frags = torch.zeros(m * _lib.load().svae_mlp_frag_elems(), **bf)
                self.w3b = torch.zeros(m * self.IPAD * DP, **bf)
                self.w3s = torch.zeros(m * self.IPAD * DP, **bf)
                self.w3tb = torch.zeros(m * DP * self.IPAD, **bf) if self.split else None

    # what only backward writes or reads: a forward-only workspace (slot 1 of a scoring session) leaves them empty
    _BACKWARD_ONLY = ("dgh", "dg4", "dz", "dtmp", "dd1", "denc", "dhs", "dgi", "demb", "dg4b", "hprev16", "mpart",
                      "pslabs", "pslabs_side", "hprev", "gates4")

    def _new_ws(self, cap, forward_only=False):
        f32 = dict(device=self.device, dtype=torch.float32)
        shapes = dict(emb=E, gi=G, hs=HP, hprev=HP, gates4=4 * HP, dgh=G, dg4=4 * HP, enc=DP, tmp=2 * L, eps=L, z=L, d1=DP,
                      logits=self.IP, dz=L, dtmp=2 * L, dd1=DP, denc=DP, dhs=HP, dgi=G, demb=E,
                      kl_row=1, nll_row=1, lse=1)
        shapes["logits"] = 0                      # allocated on demand (_ensure_logits): fused mode never needs it
        ws = {k: torch.empty(cap * w, **f32) for k, w in shapes.items()}
        if self.fused:
            bf = dict(device=self.device, dtype=torch.bfloat16)
            capp = (cap + 63) // 64 * 64
            ws["dg4b"] = torch.zeros((cap + 1) * 4 * HP, **bf)
            ws["emb16"] = torch.zeros((cap + 1) * E, **bf)       # the gathered rows as bf16 (self.emb16)
            ws["hprev16"] = torch.zeros((cap + 1) * HP, **bf)    # the previous hidden states as bf16 (self.wgrad_big)
            # (split instantiation: every bf16 image is followed by its mid and lo parts, one part stride apart)
            m = 3 if self.split else 1
            ws["h2b"] = torch.zeros(m * capp * DP, **bf)
            ws["h2s"] = torch.zeros(m * capp * DP, **bf)
            # (transposed images: read by the split instantiation only; the bf16 sweeps transpose on the LDS read)
            ws["h2tb"] = torch.zeros(m * DP * capp, **bf) if self.split else None
            nsm = _lib.load().svae_proj_ns_max()
            ws["part"] = torch.zeros(nsm * capp * 2, **f32)
            ws["mpart"] = torch.zeros(nsm * capp, **f32)
            ws["pslabs"] = torch.empty(nsm * capp * DP, **f32)
            ws["pslabs_side"] = torch.empty(nsm * self.IPAD * DP, **f32)
        if forward_only:
            # (the generic fp32 recurrence always writes its saved state: SVAE_SPLIT_GRU_F32, a tuning switch)
            keep = ("hprev", "gates4") if self.gru_prec == PREC_FP32 else ()
            for k in self._BACKWARD_ONLY:
                if k in ws and k not in keep:
                    ws[k] = torch.empty(0, device=self.device, dtype=ws[k].dtype)
        return ws

    def _use_slot(self, k):
        """Make workspace `k` (0: the engine's own; 1, 2: the forward-only ones of a scoring session) current."""
        if k == self._slot:
            return
        if self._slot != 0:                            # back to the engine's own first
            self.ws, self.loss_out = self._slot_main
            self._slot_main = None
            self._slot = 0
        if k != 0:
            have = self._slots.get(k)
            if have is None or have[2] != self.cap:
                if have is not None:                   # (grown: its kernels may still be running)
                    torch.cuda.synchronize(self.device)
                have = self._slots[k] = (self._new_ws(self.cap, forward_only=True), torch.zeros_like(self.loss_out), self.cap)
            self._slot_main = (self.ws, self.loss_out)
            self.ws, self.loss_out = have[0], have[1]
            self._slot = k

    @contextlib.contextmanager
    def scoring_session(self, sweep_cus=None):
        """Scoring many batches with unchanged parameters (evaluate()).  The caller promises that nothing writes the
        parameters between enter and exit; in return everything ``score`` derives from the parameters alone -- the MLP
        and W_ih weight fragments, the bf16 images of decoder.linear2 (35 us of a 480 us call at the ML-20M shape) -- is
        made ONCE here, on the current stream, and ``score(..., slot=1)`` may run on a second stream beside slot 0:
        the slots share those read-only images and nothing else (own workspace, own loss scalars), so the recurrence of
        one batch (a quarter of the chip) runs beside the sweep of the other.  ``scoring_stream()`` names a stream that
        runs concurrently with the current one.  sweep_cus: workgroups of the loss-only sweep inside the session (default:
        one per CU) -- with two batches in flight, a sweep that leaves the other batch's recurrence its CUs (one
        workgroup per 4 users; a sweep's workgroup and a recurrence's cannot share a CU: LDS) keeps both streams moving.
        Fused modes only; elsewhere the session changes nothing."""
        if not self.fused or self._session:
            yield self
            return
        self.score_cus = None if sweep_cus is None else max(8, min(self._ncu, int(sweep_cus)))
        self._settle()
        cur = torch.cuda.current_stream()
        self._ensure(1)
        with _Launch(cur):
            self._param_prep(False)
            if self.rowgemm:
                _lib.call("svae_rowgemm_prep_prec", self.P("gru.weight_ih_l0").data_ptr(), 0, self.wih_frags[0].data_ptr(),
                          self.prec, _stream())
        self._session_ready = self._sev("scoring_session")
        self._session_ready.record(cur)
        self._session = True
        try:
            yield self
        finally:
            self._session = False
            self.score_cus = None
            self._use_slot(0)

    def two_lane_sweep_cus(self, batch_users):
        """Workgroups of the loss-only sweep when two batches are scored beside each other: the CUs the other batch's
        recurrence (one workgroup per 4 users) leaves, never under half the chip.  ML-20M shape, B = 256, ms per batch
        on two lanes: 160: 0.333, 176: 0.330, 192: 0.322, 208: 0.329, 224: 0.331, 256: 0.334."""
        return max(self._ncu // 2, self._ncu - (int(batch_users) + 3) // 4)

    def scoring_lanes(self, n=3):
        """[the current stream, side streams ...]: the streams batches of a scoring session alternate between, slot k on
        lane k (ML-20M shape, B = 256, ms per batch: one lane 0.454, two 0.310, three 0.282)."""
        lanes = [torch.cuda.current_stream()]
        for w in range(max(0, min(int(n), 3) - 1)):
            s_ = self.scoring_stream(w)
            if s_ is not None:
                lanes.append(s_)
        return lanes

    def scoring_stream(self, which=0):
        """A stream that runs concurrently with the current one (the engine's measured side streams: which = 0, 1), for
        slots 1 and 2 of a scoring session; None where the engine has none (no overlap, non-fused modes)."""
        if not (self.fused and self.overlap):
            return None
        return self._side_stream(which, cur=torch.cuda.current_stream())

    def _ps(self, which):
        """Part stride (elements) of the bf16 operand images: 'rows' (h2b / h2s / h2tb) or 'items' (w3b / w3s / w3tb);
        0 in the bf16 instantiation."""
        if not self.split:
            return 0
        return (self.capp if which == "rows" else self.IPAD) * DP

    def _ensure_logits(self):
        if self.ws["logits"].numel() < self.cap * self.IP:
            self.ws["logits"] = torch.empty(self.cap * self.IP, device=self.device, dtype=torch.float32)

    def _gemm(self, A, B, C, M, N, K, lda, ldb, ldc, al, bl, bias=None, act=0, aux=None, ldaux=0,
              nsplit=1, slab_stride=0, tag=None, colsum=None):
        ev = self._tic(tag)
        _lib.call("svae_gemm", A.data_ptr(), B.data_ptr(), C.data_ptr(), _lib.ptr(bias), _lib.ptr(aux),
                  _lib.ptr(colsum), M, N, K, lda, ldb, ldc, ldaux, al, bl, act, self.gemm_prec, nsplit, slab_stride, _stream())
        self._toc(tag, ev)

    def _event(self):
        # events are taken from a pool filled outside the timed region (creating one costs ~10 us of host time)
        return self.event_pool.pop() if self.event_pool else torch.cuda.Event(enable_timing=True)

    def _tic(self, tag):
        if self.prof is None or tag not in self.prof:
            return None
        ev = self._event()
        ev.record(_PINNED_OBJ[0]) if _PINNED_OBJ[0] is not None else ev.record()
        return ev

    def _toc(self, tag, ev):
        if ev is None:
            return
        end = self._event()
        end.record(_PINNED_OBJ[0]) if _PINNED_OBJ[0] is not None else end.record()
        self.prof[tag].append((ev, end))

    def _bucket(self, i):
        if self.grad_sync is not None:
            self.grad_sync.bucket_ready(i)

    def _stamp(self, name, sp=None):
        """Diagnostic (tools/step_stamps.py): the device clock when the stream (default: the launch stream) gets here."""
        d = self.stamps
        if d is None:
            return
        i = d["names"].setdefault(name, len(d["names"]))
        _lib.call("svae_stamp", d["buf"].data_ptr(), d["step"] * 64 + i, _stream() if sp is None else sp)

    def _sev(self, name):
        """The ordering event kept for `name` (one per purpose: a wait refers to the record before it in host order)."""
        ev = self._sevs.get(name)
        if ev is None:
            ev = self._sevs[name] = _lib.SyncEvent()
        return ev

    def training_stream(self):
        """A high-priority stream for the caller's side of the training loop (one per device and process, shared by
        every engine: streams are a finite resource of the runtime).  The engine is stream-agnostic -- a step runs on
        whatever stream is current -- but the caller's chain is the step's critical path and competes for CUs with
        the engine's own side streams: at high priority the step measured 0.24 % shorter (0.8497 -> 0.8477 ms, three
        alternating pairs).  ``SVAE_TRAIN_HIPRIO=0``: the current stream."""
        if _opt("SVAE_TRAIN_HIPRIO", "1") != "1":
            return torch.cuda.current_stream(self.device)
        if self._train_stream is None:
            key = self._dev_index()
            with _POOL_LOCK:
                if key not in _TRAIN_STREAMS:
                    _TRAIN_STREAMS[key] = torch.cuda.Stream(device=self.device, priority=-1)
            self._train_stream = _TRAIN_STREAMS[key]
        return self._train_stream

    def _dev_index(self):
        return self.device.index if self.device.index is not None else torch.cuda.current_device()

    def _side_stream(self, which=0, cur=None):
        """Stream `which` (0, 1) of the overlapped work.  HIP maps streams onto a few hardware queues (4 per
        process here) and two streams that share a queue run IN ORDER: a side stream that landed on the caller's
        queue serialises the whole backward (seen with the process group created before the engine: 1.17 -> 1.54 ms
        per step).  So the streams are chosen by measurement: candidates are tried until one runs a spin kernel
        concurrently with the caller's stream (and with the side streams chosen before it).
        The choice is kept PER CALLER STREAM (an epoch of train_batched runs on the training stream, evaluate() on the
        caller's own: switching between them must not choose again), the candidates come from a pool of at most 8
        streams per device and priority that every engine of the process shares, and what was measured for one
        engine holds for the next (the mapping of a stream to its queue does not change).  Round 3: before this, every
        switch of the caller's stream created up to 16 new streams and spun on them; after ~330 switches in one
        process (tools/convergence.py: 160 epochs, each followed by an evaluation) a spin never completed."""
        cur = torch.cuda.current_stream() if cur is None else cur
        if self._side_for != cur.cuda_stream:
            self._sides = self._sides_by_caller.setdefault(cur.cuda_stream, [])
            self._side_for = cur.cuda_stream
        while len(self._sides) <= which:
            self._sides.append(self._pick_concurrent_stream([cur] + self._sides))
        return self._sides[which]

    def _pick_concurrent_stream(self, others, tries=8, usec=200):
        """A stream that runs concurrently with every stream of `others` (others[0] = the caller's) AND whose launches
        are not held up behind theirs.  Two tests, because HIP gives a process 4 hardware queues and maps further
        streams onto the same ones (tools/probe/queue_pipe_probe.py: null, s0, s1, s2 get a queue each; s3 shares
        s2's, s4 s1's, s5 s0's, and a high-priority stream created after them the null stream's):
         1. one-workgroup spin kernels on all the streams at once finish in the time of one (streams on DIFFERENT
            queues, or on one queue without a barrier between their packets -- this test alone passes both);
         2. a launch whose DISPATCH is long (8192 one-wave workgroups, two per CU) on one stream does not delay a
            one-workgroup kernel on the other: packets of one queue are dispatched in order, so streams that share a
            queue fail this.  It is what hurts the step: a finish kernel or the optimizer (3-5 000 workgroups) on the
            floating stream held every launch of the caller's stream for its whole length -- 1.17 instead of 0.85 ms
            per step for a loop on the default stream after an epoch on the training stream (round 3,
            tools/probe/default_stream_probe.py train_first)."""
        cur = others[0]
        prio = self.side_prio[min(len(others) - 1, len(self.side_prio) - 1)]
        dev = self._dev_index()
        ckey = (dev, prio, tuple(o.cuda_stream for o in others))
        with _POOL_LOCK:
            hit = _SIDE_CHOICE.get(ckey)
        if hit is not None:
            self.side_stream_info.append(dict(hit[1], cached=True))
            return hit[0]

        def spin_all(cand):
            torch.cuda.synchronize(self.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for o in others[1:] + [cand]:
                o.wait_event(e0)
            for o in others + [cand]:
                _lib.call("svae_spin", usec, o.cuda_stream)
            for o in others[1:] + [cand]:
                cur.wait_stream(o)
            e1.record(cur)
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e3

        def held_behind(sa, sb):
            """Is a small kernel on sb held until a long dispatch on sa is through?  (fraction of that dispatch's length
            after which the small kernel finished: ~0.15 independent, ~1 held)"""
            worst = 0.0
            for _ in range(2):
                torch.cuda.synchronize(self.device)
                e0, eb, ea = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record(sa)
                sb.wait_event(e0)
                _lib.call("svae_spin_grid", 10, 8192, 65536, sa.cuda_stream)
                _lib.call("svae_spin", 10, sb.cuda_stream)
                eb.record(sb)
                ea.record(sa)
                ea.synchronize()
                eb.synchronize()
                frac = e0.elapsed_time(eb) / max(e0.elapsed_time(ea), 1e-3)
                worst = frac if _ == 0 else min(worst, frac)     # the better of two: a one-off delay is not sharing
            return worst
        with _POOL_LOCK:
            pool = _STREAM_POOL.setdefault((dev, prio), [])
        taken = set(o.cuda_stream for o in others)
        cands, times, infos, choice = [], [], [], None
        for i in range(tries):
            with _POOL_LOCK:
                if i >= len(pool):
                    pool.append(torch.cuda.Stream(device=self.device, priority=prio))
                c = pool[i]
            if c.cuda_stream in taken:
                continue
            spin_all(c)                       # first use of a stream creates its queue: not timed
            t = min(spin_all(c), spin_all(c))
            conc = t < 1.5 * usec
            held = max(max(held_behind(o, c), held_behind(c, o)) for o in others) if conc else 1.0
            info = dict(tried=len(times) + 1, all_us=t, concurrent=conc, held_frac=round(held, 3),
                        dispatch_independent=held < 0.6)
            cands.append(c)
            times.append((not conc, held >= 0.6, t))
            infos.append(info)
            if conc and held < 0.6:
                choice = (c, info)
                break
        if choice is None:
            # nothing passed both tests (more streams in use than the process has queues): the best of what there is
            best = min(range(len(times)), key=lambda i: times[i])
            choice = (cands[best], infos[best])
        with _POOL_LOCK:
            _SIDE_CHOICE[ckey] = choice
        self.side_stream_info.append(choice[1])
        return choice[0]

    def _wgrad(self, dY, X, wname, R, N, K, ldy, ldx, slabs=None, remap=None, dy_bf16=False, x_bf16=False):
        """Weight + bias gradient of y = x W^T + b in one pass: gW[N][K] = dY[R][N]^T X[R][K] and
        gb[N] = column sums of dY (folded into the GEMM).  Split-K over rows writes one slab per split
        (weights then bias), summed in a fixed order: deterministic, no atomics.  The bias gradient sits
        right behind the weight gradient in the flat buffer (layout.py keeps them adjacent)."""
        out = self._gbase + 4 * self.layout.offset(wname)
        slabs = self.slabs if slabs is None else slabs
        sp = slabs.data_ptr()
        ns = max(1, min(self.nsplit, (R + 127) // 128))
        stride = N * K + N
        if self.fused:
            ev = self._tic("wgrad")
            rf, ra = remap if remap is not None else (N if not dy_bf16 else (N + 7) // 8 * 8, 0)
            fn = "svae_wgrad_cols_bf16dy" if dy_bf16 else ("svae_wgrad_cols_prec" if self.split else "svae_wgrad_cols")
            if x_bf16:
                fn = "svae_wgrad_cols_bf16dy_bf16x"
            tail = (rf, ra, self.prec, _stream()) if fn == "svae_wgrad_cols_prec" else (rf, ra, _stream())
            if ns == 1:
                _lib.call(fn, dY.data_ptr(), X.data_ptr(), out, out + 4 * N * K, R, N, K, ldy, ldx, 1, 0, *tail)
            else:
                _lib.call(fn, dY.data_ptr(), X.data_ptr(), sp, sp + 4 * N * K, R, N, K, ldy, ldx, ns, stride, *tail)
                _lib.call("svae_reduce_slabs", sp, out, stride, ns, stride, _stream())
            self._toc("wgrad", ev)
            return
        outv = self.grads[self.layout.offset(wname):]
        if ns == 1:
            self._gemm(dY, X, outv, N, K, R, ldy, ldx, K, 1, 1, colsum=outv[N * K:])
        else:
            self._gemm(dY, X, slabs, N, K, R, ldy, ldx, K, 1, 1, nsplit=ns, slab_stride=stride,
                       colsum=slabs[N * K:])
            _lib.call("svae_reduce_slabs", sp, out, stride, ns, stride, _stream())

    def _colsum(self, A, out, R, N, ld):
        nparts = max(1, min(64, (R + 127) // 128))
        _lib.call("svae_colsum", A.data_ptr(), out.data_ptr(), self.colparts.data_ptr(), R, N, ld, nparts, _stream())

    # ------------------------------------------------------------------ forward
    def forward(self, plan, anneal=0.0, eps=None, seed=0, step=0, want_grad=True, want_loss=True, defer_loss=False):
        """Encoder GRU -> latent -> decoder -> logits (+ loss and dlogits).

        eps: optional [R, 64] fp32 device tensor in plan row order (parity mode: the reference's
        np.random.normal draws, nb:744); otherwise Philox noise keyed by (seed, step, row_key).
        After the call ``ws['logits']`` holds the logits (want_grad=False) or dlogits (want_grad=True).
        """
        if not self._in_step:
            self._settle()
        self.generation += 1
        cur = torch.cuda.current_stream()
        with _Launch(cur):
            return self._forward(cur, plan, anneal, eps, seed, step, want_grad, want_loss, defer_loss)

    def _forward(self, cur, plan, anneal, eps, seed, step, want_grad, want_loss, defer_loss):
        R = plan.R
        self._ensure(R)
        w, st = self.ws, _stream()
        self._plan, self._anneal = plan, float(anneal)
        if getattr(plan, "ready", None) is not None:      # uploaded on a copy stream (batch.PlanPrefetcher)
            _lib.stream_wait(cur, plan.ready)
            plan.ready = None
            # its buffers were allocated on that stream: tell the allocator which streams read them, or a later
            # upload could reuse the memory while this step's kernels are still running
            streams = [cur] + list(self._sides)
            for t in [plan._buf] + [v[0] for v in plan.proj_sched.values()]:
                if t is not None and t.is_cuda:
                    for s_ in streams:
                        t.record_stream(s_)
        params_ready = None
        eps_done = False
        if eps is not None:
            # the caller's noise goes in FIRST, on the caller's stream and ahead of `step_begin`: the side stream's MLP
            # chain of the second time range reads w["eps"] and is ordered behind that event only (validated before
            # anything has been enqueued)
            if eps.numel() != R * L:
                raise ValueError("eps must have %d x %d elements" % (R, L))
            w["eps"][:R * L].copy_(eps.reshape(-1))
        # scoring session: the parameter-derived operands exist already and are shared with the other slot (read-only);
        # nothing goes to the side stream -- the other slot may be running there
        sess = bool(self._session and self.fused and not want_grad)
        if self.fused and self.overlap and not sess:
            side = self._side_stream(cur=cur)
            step_begin = self._sev("step_begin")
            step_begin.record(cur)
        self._stamp("q1 step begins")
        # the caller's stream first: its kernels head the critical path, the side stream's are not needed before the
        # decoder side
        # With a two-range forward (below) only the rows of the first range's timesteps are needed by the recurrence
        # that heads the step: their gather and input gates here, the others' on the side stream beside it (+1.2 %).
        # The first recurrence reads the input gates of EVERY row of its timesteps, rows < row_off[t1] -- which is up to 63
        # rows more than plan.split_r (that cut is rounded DOWN to 64 rows for the decoder side): the head's share ends
        # at row_off[t1] rounded UP.  (Round 2 cut both at split_r: the rows in between were gathered and multiplied on
        # the side stream, usually AFTER the first recurrence had read them -- it then saw the input gates of the batch
        # before.  Invisible to every test that repeats one plan; found in round 3 with
        # tests/test_gpu_fullsize.py::test_a_step_does_not_depend_on_the_batch_before_it.)
        gi_split = 0
        if (self.fused and self.rowgemm and self.gi_split and self.overlap and self.time_split and want_loss and want_grad
                and self.gru_prec in (PREC_BF16, PREC_SPLIT) and plan.split_t > 0):
            gi_split = (int(plan._row_off_host[plan.split_t]) + 63) // 64 * 64
            if gi_split >= R:
                gi_split = 0
        head_fused = self.fused and self.rowgemm and self.prec == PREC_BF16 and self.head_fused
        # (bf16 rows need the bf16 gradient array on the other side: the weight-gradient kernel's bf16-X form reads both)
        # (the fused loss paths only: training, whose backward then has the bf16 gradient array, and scoring, which
        #  has no backward; the dense notebook surface keeps fp32 rows)
        e16 = (head_fused and self.emb16 and self.gru_prec == PREC_BF16 and self.merged_dg and self.dg_bf16
               and not self.bptt_split and want_loss)
        self._emb16_live = bool(e16)
        if sess:
            _lib.stream_wait(cur, self._session_ready)
            head_fused = False                    # (the W_ih fragments were made when the session began)
        if sess and e16:
            _lib.call("svae_embed_fwd_bf16", self.P("item_embed.weight").data_ptr(), plan.x.data_ptr(),
                      w["emb16"].data_ptr(), R, self.I, st)
        elif head_fused:
            # gather + W_ih fragments in one launch (both head the step, neither needs the other)
            _lib.call("svae_embed_fwd_wih_prep", self.P("item_embed.weight").data_ptr(), plan.x.data_ptr(),
                      (w["emb16"] if e16 else w["emb"]).data_ptr(), gi_split if gi_split else R, self.I,
                      self.P("gru.weight_ih_l0").data_ptr(), self.wih_frags[0].data_ptr(), 1 if e16 else 0, st)
        else:
            _lib.call("svae_embed_fwd", self.P("item_embed.weight").data_ptr(), plan.x.data_ptr(),
                      w["emb"].data_ptr(), gi_split if gi_split else R, self.I, st)
        if self.fused and self.rowgemm:
            # W_ih fragments are built here, on the caller's stream (the GEMM is the second kernel of the step: the
            # side stream's preparation would not be ready in time), 2 us
            if not head_fused and not sess:
                _lib.call("svae_rowgemm_prep_prec", self.P("gru.weight_ih_l0").data_ptr(), 0,
                          self.wih_frags[0].data_ptr(), self.prec, st)
            ev = self._tic("gi_gemm")
            if e16:
                _lib.call("svae_rowgemm_gi_bf16x", w["emb16"].data_ptr(), E, self.wih_frags[0].data_ptr(),
                          self.P("gru.bias_ih_l0").data_ptr(), w["gi"].data_ptr(), gi_split if gi_split else R, st)
            else:
                _lib.call("svae_rowgemm_ld_prec", w["emb"].data_ptr(), E, self.wih_frags[0].data_ptr(),
                          self.P("gru.bias_ih_l0").data_ptr(), w["gi"].data_ptr(), gi_split if gi_split else R, 0,
                          self.prec, st)
            self._toc("gi_gemm", ev)
            if gi_split:
                head_done = self._sev("head_done")
                head_done.record(cur)
        else:
            self._gemm(w["emb"], self.P("gru.weight_ih_l0"), w["gi"], R, G, E, E, E, G, 0, 0,
                       bias=self.P("gru.bias_ih_l0"))
        # Two time ranges (batch.choose_time_split): the recurrence of steps >= t1 runs on the side stream beside
        # the decoder side of the rows of steps < t1 (rows are time-major: the first r1 of them).
        side_prep = self.fused and self.overlap and not sess
        t1, r1 = (plan.split_t, plan.split_r) if (self.time_split and side_prep and want_loss
                                                  and want_grad and self.gru_prec in (PREC_BF16, PREC_SPLIT)) else (0, 0)

        # h_{t-1} and the gates are saved for the BPTT only: a forward without gradients (scoring) leaves them out
        # (bf16 / split recurrence kernels; 5.4 KB per row of HBM writes)
        save_state = bool(want_grad or self.gru_prec == PREC_FP32)

        def gru_range(a, b, tag):
            ev = self._tic(tag)
            _lib.call("svae_gru_fwd_range", w["gi"].data_ptr(), self.P("gru.weight_hh_l0").data_ptr(),
                      self.P("gru.bias_hh_l0").data_ptr(), plan.row_off.data_ptr(), plan.nact.data_ptr(),
                      plan.B, plan.Tmax, a, b, w["hs"].data_ptr(), w["hprev"].data_ptr() if save_state else None,
                      w["gates4"].data_ptr() if save_state else None, self.gru_prec, _stream())
            self._toc(tag, ev)

        # Host order = critical path first: the recurrence that heads the step is enqueued before the side stream's
        # eight launches (when the host is not far ahead of the GPU -- the first steps of a run, a profiler attached,
        # collectives in the step -- the caller's queue otherwise runs dry for as long as those take: 50-130 us in a
        # kernel trace).
        if t1 > 0:
            gru_range(0, t1, "gru_fwd")
            self._stamp("q1 recurrence of range 1 done")
            first_done = self._sev("first_done")
            first_done.record(cur)
        else:
            gru_range(0, plan.Tmax, "gru_fwd")
        second_done = None
        if side_prep:
            # everything that depends on the parameters only (bf16 operand copies, weight fragments, W_hh^T, the
            # zero fill of the embedding gradient) runs on the side stream beside the embedding / GRU kernels
            _lib.stream_wait(side, step_begin)
            with _Launch(side, self._torch_ctx()):
                self._param_prep(want_grad)
                if eps is None:               # the noise depends on the plan and the step only
                    _lib.call("svae_philox_normal", w["eps"].data_ptr(), plan.row_key.data_ptr(), R, int(seed),
                              int(step) & 0xFFFFFFFF, _stream())
                    eps_done = True
                params_ready = self._sev("params_ready")
                params_ready.record(side)
                self._stamp("side parameters prepared, noise drawn")
                if gi_split:
                    # (behind the parameter preparation: the decoder side of the first range waits for that, the
                    #  second recurrence, which needs these rows, only starts when the first one has ended)
                    if e16:
                        _lib.call("svae_embed_fwd_bf16", self.P("item_embed.weight").data_ptr(),
                                  plan.x.data_ptr() + 4 * gi_split, w["emb16"].data_ptr() + 2 * E * gi_split,
                                  R - gi_split, self.I, _stream())
                    else:
                        _lib.call("svae_embed_fwd", self.P("item_embed.weight").data_ptr(), plan.x.data_ptr() + 4 * gi_split,
                                  w["emb"].data_ptr() + 4 * E * gi_split, R - gi_split, self.I, _stream())
                    _lib.stream_wait(side, head_done)          # (the W_ih fragments)
                    if e16:
                        _lib.call("svae_rowgemm_gi_bf16x", w["emb16"].data_ptr() + 2 * E * gi_split, E,
                                  self.wih_frags[0].data_ptr(), self.P("gru.bias_ih_l0").data_ptr(),
                                  w["gi"].data_ptr() + 4 * G * gi_split, R - gi_split, _stream())
                    else:
                        _lib.call("svae_rowgemm_ld_prec", w["emb"].data_ptr() + 4 * E * gi_split, E,
                                  self.wih_frags[0].data_ptr(), self.P("gru.bias_ih_l0").data_ptr(),
                                  w["gi"].data_ptr() + 4 * G * gi_split, R - gi_split, 0, self.prec, _stream())
                if t1 > 0:
                    self._stamp("side input gates of range 2 done")
                    _lib.stream_wait(side, first_done)
                    self._stamp("side recurrence of range 2 may start")
                    gru_range(t1, plan.Tmax, "gru_fwd2")
                    self._stamp("side recurrence of range 2 done")
                    if self.mlp2_side and self._w3_event is not None:
                        # the MLP chain of the second range's rows here, behind their recurrence, on the CUs it has
                        # just freed: the caller's stream is still in the first range's sweep (the floated
                        # decoder.linear2 work of the step before read d1 / h2b, which this overwrites)
                        _lib.stream_wait(side, self._w3_event)
                    if self.mlp2_side:
                        self._decoder_rows(plan, r1, R, self._ncu, sweep=False)
                    self._stamp("side MLP chain of range 2 done")
                    second_done = self._sev("second_done")
                    second_done.record(side)
        if eps is None and not eps_done:
            _lib.call("svae_philox_normal", w["eps"].data_ptr(), plan.row_key.data_ptr(), R, int(seed),
                      int(step) & 0xFFFFFFFF, st)
        rows_prepared = False
        if self.fused:
            # encoder -> (mu | log_sigma) -> z, KL -> decoder hidden in one kernel (bf16 MFMA mode)
            if params_ready is not None:
                _lib.stream_wait(cur, params_ready)
            elif not sess:
                self._param_prep(want_grad)
            if self._w3_event is not None:
                # the last step's floated decoder.linear2 work: it read d1 / h2b / h2tb, which the kernels below
                # overwrite, and it made the operand copies of the new weights, which they read
                _lib.stream_wait(cur, self._w3_event)
                self._w3_event = None
                self._float_plan = None       # (kept alive until here: the floated kernels read its tables)
            if t1 > 0:
                self._last_fused = True
                self._stamp("q1 decoder side of range 1 may start (parameters, last step's floated work)")
                self._decoder_rows(plan, 0, r1, self._range1_cus(plan))
                self._stamp("q1 decoder side of range 1 done")
                _lib.stream_wait(cur, second_done)
                self._stamp("q1 sweep of range 2 may start")
                self._decoder_rows(plan, r1, R, self._ncu, mlp=not self.mlp2_side)
                self._stamp("q1 sweep of range 2 done")
                if self.dw_early:
                    self._sev("fwd_done").record(cur)
                self._fused_grad = True
                if defer_loss and self.overlap:
                    self._loss_deferred = True
                else:
                    _lib.call("svae_loss_reduce", w["nll_row"].data_ptr(), w["kl_row"].data_ptr(), plan.w_nll.data_ptr(),
                              plan.w_kl.data_ptr(), float(anneal), R, self.loss_out.data_ptr(), st)
                return self.loss_out
            # one time range: the MLP chain, and with it (fused_prep, when a fused sweep follows) the sweep's operand
            # copies of its last output in the same kernel -- scoring's forward takes this path at every call
            rows_prepared = bool(want_loss and self.fused_prep)
            if rows_prepared:
                # (enc / mu, log sigma / z are the backward's inputs: a forward without gradients does not write them)
                self._decoder_rows(plan, 0, R, self._ncu, sweep=False, save_acts=bool(want_grad))
            else:
                self._mlp_fwd_rows(0, R)
        else:
            self._gemm(w["hs"], self.P("encoder.linear1.weight"), w["enc"], R, DP, HP, HP, HP, DP, 0, 0,
                       bias=self.P("encoder.linear1.bias"), act=1)
            self._gemm(w["enc"], self.P("linear1.weight"), w["tmp"], R, 2 * L, DP, DP, DP, 2 * L, 0, 0,
                       bias=self.P("linear1.bias"))
            _lib.call("svae_latent_fwd", w["tmp"].data_ptr(), w["eps"].data_ptr(), w["z"].data_ptr(),
                      w["kl_row"].data_ptr(), R, st)
            self._gemm(w["z"], self.P("decoder.linear1.weight"), w["d1"], R, DP, L, L, L, DP, 0, 0,
                       bias=self.P("decoder.linear1.bias"), act=1)
        if self.fused and want_loss:
            self._last_fused = True
            if not rows_prepared:
                self._proj_prepare(R)
            if want_grad:
                # forward + hidden-side backward in one sweep over the items (dd1 is ready after this)
                args = (w["h2s"].data_ptr(), w["h2b"].data_ptr(), self.w3b.data_ptr(), _lib.ptr(self.w3tb),
                        self.IPAD, w["d1"].data_ptr(), self.P("decoder.linear2.weight").data_ptr(),
                        self.P("decoder.linear2.bias").data_ptr(), plan.tgt.data_ptr(), plan.w_nll.data_ptr(),
                        R, self.I, plan.K, w["pslabs"].data_ptr(), w["mpart"].data_ptr(), w["nll_row"].data_ptr(),
                        w["lse"].data_ptr(), w["dd1"].data_ptr()) + self._proj_sched(plan, "rows")
                self._split_call("svae_proj_nll_fwd_dh", args, ((1, "proj_fdh_fused"), (2, "proj_fdh_finish")),
                                 strides=(self._ps("rows"), self._ps("items")))
                if self.dw_early and self.overlap:
                    self._sev("fwd_done").record(cur)
            else:
                _lib.call("svae_proj_nll_fwd_split", w["h2s"].data_ptr(), self.w3b.data_ptr(), w["d1"].data_ptr(),
                          self.P("decoder.linear2.weight").data_ptr(), self.P("decoder.linear2.bias").data_ptr(),
                          plan.tgt.data_ptr(), R, self.I, plan.K, w["part"].data_ptr(), w["nll_row"].data_ptr(),
                          w["lse"].data_ptr(), *self._proj_sched(plan, "rows", ncu=self.score_cus, flat=False), self._ps("rows"),
                          self._ps("items"), st)
            self._fused_grad = bool(want_grad)
            if defer_loss and want_grad and self.overlap:
                # nothing in backward reads the scalar loss: backward() reduces it on the side stream, behind the
                # event that stream waits for anyway
                self._loss_deferred = True
            else:
                _lib.call("svae_loss_reduce", w["nll_row"].data_ptr(), w["kl_row"].data_ptr(), plan.w_nll.data_ptr(),
                          plan.w_kl.data_ptr(), float(anneal), R, self.loss_out.data_ptr(), st)
            return self.loss_out
        self._last_fused = False
        if not want_loss:
            return self.loss_out                  # callers of the dense surface use dense_logits()
        self._ensure_logits()
        self._gemm(w["d1"], self.P("decoder.linear2.weight"), w["logits"], R, self.I, DP, DP, DP, self.IP, 0, 0,
                   bias=self.P("decoder.linear2.bias"), tag="proj_fwd")
        if want_loss:
            ev = self._tic("softmax_nll")
            _lib.call("svae_softmax_nll", w["logits"].data_ptr(), plan.tgt.data_ptr(), plan.w_nll.data_ptr(),
                      w["nll_row"].data_ptr(), w["lse"].data_ptr(), R, self.I, self.IP, plan.K,
                      1 if want_grad else 0, st)
            self._toc("softmax_nll", ev)
            _lib.call("svae_loss_reduce", w["nll_row"].data_ptr(), w["kl_row"].data_ptr(), plan.w_nll.data_ptr(),
                      plan.w_kl.data_ptr(), float(anneal), R, self.loss_out.data_ptr(), st)
        return self.loss_out

    def _mlp_fwd_rows(self, r0, r1):
        w, n = self.ws, r1 - r0
        ev = self._tic("mlp_fwd")
        _lib.call("svae_mlp_fwd_prec", w["hs"].data_ptr() + 4 * HP * r0, w["eps"].data_ptr() + 4 * L * r0,
                  self.mlp_frags.data_ptr(), self.P("encoder.linear1.bias").data_ptr(),
                  self.P("linear1.bias").data_ptr(), self.P("decoder.linear1.bias").data_ptr(),
                  w["enc"].data_ptr() + 4 * DP * r0, w["tmp"].data_ptr() + 8 * L * r0, w["z"].data_ptr() + 4 * L * r0,
                  w["d1"].data_ptr() + 4 * DP * r0, w["kl_row"].data_ptr() + 4 * r0, n, self.prec, _stream())
        self._toc("mlp_fwd", ev)

    def _decoder_rows(self, plan, r0, r1, ncu, mlp=True, sweep=True, save_acts=True):
        """Decoder side of rows [r0, r1) of the plan (r0 a multiple of 64): MLP chain, bf16 operand copies (mlp),
        forward + dL/dh sweep of the projection and its finish (sweep).  Every array is row-major, so a row range is
        a pointer offset; the slab scratch of a range starts where the previous range's ends."""
        w, R, n = self.ws, plan.R, r1 - r0
        Rp = (R + 63) // 64 * 64
        if not mlp:
            pass
        elif self.fused_prep:
            # MLP chain and the sweeps' operand copies in one kernel (svae_mlp_fwd_prep)
            ev = self._tic("mlp_fwd")
            _lib.call("svae_mlp_fwd_prep", w["hs"].data_ptr(), w["eps"].data_ptr(), self.mlp_frags.data_ptr(),
                      self.P("encoder.linear1.bias").data_ptr(), self.P("linear1.bias").data_ptr(),
                      self.P("decoder.linear1.bias").data_ptr(), w["enc"].data_ptr() if save_acts else None,
                      w["tmp"].data_ptr() if save_acts else None, w["z"].data_ptr() if save_acts else None,
                      w["d1"].data_ptr(), w["kl_row"].data_ptr(), R, Rp, r0, Rp if r1 == R else r1,
                      w["h2b"].data_ptr(), w["h2s"].data_ptr(), _lib.ptr(w["h2tb"]), _stream())
            self._toc("mlp_fwd", ev)
        else:
            self._mlp_fwd_rows(r0, r1)
            _lib.call("svae_proj_prep_rows_range_split", w["d1"].data_ptr(), R, Rp, r0, Rp if r1 == R else r1,
                      w["h2b"].data_ptr(), w["h2s"].data_ptr(), _lib.ptr(w["h2tb"]), self._ps("rows"), _stream())
        if not sweep:
            return
        nsm = _lib.load().svae_proj_ns_max()
        args = (w["h2s"].data_ptr() + 2 * DP * r0, w["h2b"].data_ptr() + 2 * DP * r0, self.w3b.data_ptr(),
                _lib.ptr(self.w3tb), self.IPAD, w["d1"].data_ptr() + 4 * DP * r0,
                self.P("decoder.linear2.weight").data_ptr(), self.P("decoder.linear2.bias").data_ptr(),
                plan.tgt.data_ptr() + 4 * plan.K * r0, plan.w_nll.data_ptr() + 4 * r0, n, self.I, plan.K,
                w["pslabs"].data_ptr() + 4 * nsm * DP * r0, w["mpart"].data_ptr() + 4 * nsm * r0,
                w["nll_row"].data_ptr() + 4 * r0, w["lse"].data_ptr() + 4 * r0,
                w["dd1"].data_ptr() + 4 * DP * r0) + self._proj_sched(plan, "rows", nx=n, ncu=ncu)
        self._split_call("svae_proj_nll_fwd_dh", args, ((1, "proj_fdh_fused"), (2, "proj_fdh_finish")),
                         strides=(self._ps("rows"), self._ps("items")))

    def _split_call(self, name, args, parts, strides=None):
        """One C-ABI call covering several kernels; when profiling, one call per kernel so each is bracketed.
        strides: (row, item) part strides of the operand images in the split instantiation (then the `_split`
        entry point is called)."""
        if strides is not None and strides[0] > 0:
            _lib.call(name + "_split", *args, sum(b for b, _ in parts), *strides, _stream())
            return
        if self.prof is None:
            _lib.call(name, *args, sum(b for b, _ in parts), _stream())
            return
        for bit, tag in parts:
            if bit == 1 and tag in self.prof:
                # the sweep kernel: start / stop events ride on the dispatch itself (no marker packets, see proj.hip)
                tm = self.timer_pool.pop() if self.timer_pool else _lib.DispatchTimer()
                tm.arm()
                _lib.call(name, *args, bit, _stream())
                self.prof[tag].append((tm, tm))
                continue
            ev = self._tic(tag)
            _lib.call(name, *args, bit, _stream())
            self._toc(tag, ev)

    def _proj_sched(self, plan, side, nx=None, ncu=None, host_only=False, flat=None):
        """(device table, nwg, ns_q, ns_rem) of a projection sweep over this plan: 'rows' = rows resident, items
        streamed (forward / dL/dh sweep; nx = the number of rows when only a range of the plan is swept), 'items' =
        the weight-gradient sweep.  Built once per plan and engine.
        flat (default: the engine's setting, bf16 instantiation only): the flat work division of round 3
        (svae_proj_schedule_flat: equal-cost contiguous ranges of the tile space, one per workgroup) instead of the
        balanced one (every x tile cut into q or q + 1 pieces); the loss-only sweep of `score` always takes the latter."""
        ncu = self._ncu if ncu is None else max(8, int(ncu))
        flat = (self.flat_sched and not self.split) if flat is None else bool(flat)
        if flat:
            nwg_want = self._dw_workgroups(plan.R, ncu) if side == "items" else ncu
            key = (side, self.I, nwg_want, nx, "flat", self.flat_boundary)
            shape = (("rows-flat", ((plan.R if nx is None else nx) + 255) // 256, nwg_want, self.flat_boundary) if side == "rows"
                     else ("items-flat", (plan.R + 63) // 64, nwg_want, self.flat_boundary))
        else:
            key = (side, self.I, ncu, nx, self.dw_q if side == "items" else self.fdh_q0)
            shape = (("rows", ((plan.R if nx is None else nx) + 255) // 256, ncu, self.fdh_q0) if side == "rows"
                     else ("items", (plan.R + 63) // 64, self.dw_q))
        cache = plan.proj_sched
        # A table depends on the plan only through its tile counts: x tiles of 256 rows (the forward sweep of nx
        # rows) or y tiles of 64 rows (the weight-gradient sweep of R rows).  The engine keeps every table it has
        # uploaded under that shape: over an epoch most plans find theirs here, and `prepare` -- which runs on the
        # prefetcher's worker threads and holds the interpreter lock for most of its 0.15-0.25 ms -- becomes a few
        # dictionary look-ups.  (Inserted only after its upload has been enqueued: whoever finds a table there orders
        # its own work behind that copy -- the prefetcher's workers share one copy stream, the others use the caller's.)
        if key not in cache and shape in self._sched_cache:
            cache[key] = self._sched_cache[shape]
        if host_only:
            if key in cache:
                return None
        if key not in cache or host_only:
            if flat:
                # (what was measured for the workgroup counts and the boundary cost: DESIGN.md section 4, "flat schedule")
                if side == "rows":
                    tab, n, q, rem = _lib.proj_schedule_flat(plan.R if nx is None else nx, self.I, nwg_want, self.flat_boundary)
                else:
                    tab, n, q, rem = _lib.proj_schedule_flat(self.I, plan.R, nwg_want, self.flat_boundary)
            elif side == "rows":
                # nominal 4-way split (the workgroup count is the next whole multiple of the CU count): fewer, longer
                # workgroups than the 8-way nominal write and re-read fewer slabs -- measured 3: 243.8k, 4: 245.0k,
                # 5: 243.2k, 6: 242.0k, 8: 240.1k seq/s
                tab, n, q, rem = _lib.proj_schedule(plan.R if nx is None else nx, self.I, ncu,
                                                    uniform=-self.fdh_q0)
            else:
                # uniform 2-way split.  With the sweep on its own stream past the end of the step, fewer and longer
                # workgroups win (slab traffic, and CUs left to the kernels beside it).  Round 1: 2: 249.4k, 3: 248.9k,
                # 4: 249.0k, 5: 247.6k, 6: 248.4k, 7: 245.1k, 8: 244.5k seq/s (before the sweep floated, 8 was best).
                # Round 2, pipelined kernel (median ms per step over repeated regions): 1: 1.003, 2: 0.934, 3: 0.946,
                # 4: 0.945, 5: 0.959, 6: 0.949, 8: 0.967 -- at 2 the 158 workgroups leave 98 CUs to the BPTT and the
                # weight-gradient GEMMs (0.36 -> 0.27 ms) and write half the slabs of the 4-way split.
                fq = self.dw_q
                tab, n, q, rem = _lib.proj_schedule(self.I, plan.R, ncu, uniform=1 + fq)   # (ncu unused: uniform)
            if host_only:
                return key, tab, (n, q, rem), shape
            cache[key] = self._sched_cache[shape] = (torch.from_numpy(tab).to(self.device), n, q, rem)
        t, n, q, rem = cache[key]
        return (t.data_ptr(), n, q, rem)

    def _dw_workgroups(self, R, ncu):
        """Workgroups of the flat weight-gradient sweep (see `dw_wgs`): by catalogue size, from the measured optima --
        the sweep grows with rows x items and what runs beside it only with rows, so its share of the chip grows with
        the catalogue, but not smoothly (the step time has cliffs where the caller's chain stops finding CUs).  ms per
        step at B = 256 (tools/ab_multi.sh with SVAE_DW_WGS_EXACT=1, one box per row):
          20 000 items   152: 0.863   158: 0.857   160: 0.856   170: 0.922   192: 0.932
          40 000         158: 1.237   168: 1.283   192: 1.280   200: 1.343
          65 536         176: 1.826   192: 1.813   208: 1.916   224: 1.898   232: 1.964
          131 072        192: 3.385   208: 3.393   216: 3.372   224: 3.363
          1 000 000      158: 14.70   192: 13.84   208: 13.70   224: 13.36   237: 14.17   256: 13.98
                         (B = 1024: 158: 51.1, 192: 47.7, 208: 47.1, 224: 46.0, 256: 46.2)"""
        if self.dw_wgs_exact:
            return min(ncu, self.dw_wgs)
        want = self.dw_wgs if self.I <= 51200 else max(self.dw_wgs, 192 if self.I <= 100000 else ncu - 32)
        return max(8, min(ncu, want))

    def _range1_cus(self, plan):
        """Workgroups of the first time range's forward sweep: the CUs the recurrence of the second range (one workgroup
        per 4 users) leaves free -- but never under half the chip: with the flat schedule this IS the workgroup count,
        and at B = 1024 the difference is 0 (8 workgroups swept the whole range: 125 ms instead of 21 at 1M items).
        Above 51 200 items the whole chip: the sweep then lasts many times the recurrence, which finds its CUs as the
        sweep's workgroups end (1M items, ms per step at B = 256: 192 -> 13.41, 224 -> 13.04, 256 -> 12.88; at
        B = 1024: 128 -> 45.5, 192 -> 43.4, 256 -> 43.0; ML-20M shape at B = 1024: 128 / 192 / 256 within 0.5 %)."""
        if self.range1_cus > 0:
            return min(self._ncu, self.range1_cus)
        if self.I > 51200:
            return self._ncu
        return max(self._ncu // 2, self._ncu - (plan.B + 3) // 4)

    def prepare(self, plan, scoring=False):
        """Build and upload everything the fused step derives from a plan (the work division tables of the
        projection sweeps) BEFORE the step is launched.  Built lazily instead, the upload is a synchronous copy on
        the caller's stream in the middle of backward: the host waits for the GRU's BPTT and the weight-gradient
        sweep starts after it rather than beside it (seen in a kernel trace: +0.2-0.4 ms on a plan's first step).
        scoring: the plan will only be scored (``score``: the loss-only sweep over all rows)."""
        return self.prepare_upload(plan, self.prepare_host(plan, scoring))

    def prepare_host(self, plan, scoring=False):
        """First half of ``prepare``: the tables this engine does not have yet, built on the host (no device call; a
        prefetcher's builder threads run it).  Returns what ``prepare_upload`` takes."""
        if not self.fused:
            return []
        want = [("rows", None, self.score_cus, False)] if scoring else [("rows", None, None, None), ("items", None, None, None)]
        if not scoring and self.time_split and plan.split_t > 0:
            want += [("rows", plan.split_r, self._range1_cus(plan), None),
                     ("rows", plan.R - plan.split_r, self._ncu, None)]
        return [m for m in (self._proj_sched(plan, sd, nx=nx, ncu=ncu, host_only=True, flat=fl) for sd, nx, ncu, fl in want)
                if m is not None]

    def prepare_upload(self, plan, made):
        """Second half of ``prepare``: one asynchronous copy of the new tables on the current stream."""
        # (a table another plan has uploaded since is found by shape when the sweep asks for it)
        made = [m for m in made if m[0] not in plan.proj_sched and m[3] not in self._sched_cache]
        if not made:
            return plan
        # all tables through ONE pinned buffer and one asynchronous copy on the caller's current stream: a pageable
        # source makes the copy synchronous, and a prefetcher's worker then waits for whatever the GPU has queued on
        # the hardware queue its copy stream shares (measured: the training loop blocked on plans 0.3-0.8 ms a step)
        if self._pin_pool is None:
            from .batch import _PinnedPool
            with _POOL_LOCK:
                if self._pin_pool is None:
                    self._pin_pool = _PinnedPool()
        n = sum(int(t.shape[0]) for _, t, _, _ in made)
        host = self._pin_pool.take(16 * n)
        hv = host[:16 * n].view(torch.int32).view(n, 4)
        o = 0
        for _, t, _, _ in made:
            hv[o:o + t.shape[0]] = torch.from_numpy(t)
            o += int(t.shape[0])
        dev = hv.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pin_pool.give(host, ev)
        o = 0
        for key, t, (nwg, q, rem), shape in made:
            plan.proj_sched[key] = self._sched_cache[shape] = (dev[o:o + t.shape[0]], nwg, q, rem)
            o += int(t.shape[0])
        return plan

    def _param_prep(self, want_grad):
        """Per-step work that depends on the parameters only (they change at every optimizer step)."""
        st = _stream()
        _lib.call("svae_mlp_prep_prec", self.P("encoder.linear1.weight").data_ptr(), self.P("linear1.weight").data_ptr(),
                  self.P("decoder.linear1.weight").data_ptr(), self.mlp_frags.data_ptr(), self.prec, st)
        if not self._w3_fresh:
            self._proj_prep_items(st)
        if want_grad and self.rowgemm:
            _lib.call("svae_rowgemm_prep_prec", self.P("gru.weight_ih_l0").data_ptr(), 1, self.wih_frags[1].data_ptr(),
                      self.prec, st)
        if want_grad:
            _lib.call("svae_transpose", self.P("gru.weight_hh_l0").data_ptr(), self.whh_t.data_ptr(), G, HP, st)
            e = self.layout.entries["item_embed.weight"]
            _lib.call("svae_zero", self._gbase + 4 * e.offset, 4 * e.size, st)
        self._prepped_grad = bool(want_grad)

    def _proj_prep_items(self, st):
        _lib.call("svae_proj_prep_items_split", self.P("decoder.linear2.weight").data_ptr(),
                  self.P("decoder.linear2.bias").data_ptr(), self.I, self.IPAD, self.w3b.data_ptr(),
                  self.w3s.data_ptr(), _lib.ptr(self.w3tb), self._ps("items"), st)

    def _proj_prepare(self, R):
        """bf16 operand copies of the decoder hidden rows for the fused projection kernels."""
        w, st = self.ws, _stream()
        Rp = (R + 63) // 64 * 64
        _lib.call("svae_proj_prep_rows_range_split", w["d1"].data_ptr(), R, Rp, 0, Rp, w["h2b"].data_ptr(),
                  w["h2s"].data_ptr(), _lib.ptr(w["h2tb"]), self._ps("rows"), st)

    def dense_logits(self, rows_idx=None):
        """Dense logits [n, I] for all rows of the last forward (rows_idx None) or a gathered subset
        (int32 device tensor).  This is the Model.forward / evaluate() surface; training never calls it."""
        self._settle()
        plan, st = self._plan, _stream()
        f32 = dict(device=self.device, dtype=torch.float32)
        if rows_idx is None:
            n, src = plan.R, self.ws["d1"]
        else:
            n = int(rows_idx.numel())
            src = torch.empty(n * DP, **f32)
            _lib.call("svae_gather_rows", self.ws["d1"].data_ptr(), rows_idx.data_ptr(), src.data_ptr(),
                      n, DP, DP, DP, st)
        out = torch.empty(n * self.IP, **f32)
        _lib.call("svae_logits_fwd", src.data_ptr(), self.P("decoder.linear2.weight").data_ptr(),
                  self.P("decoder.linear2.bias").data_ptr(), out.data_ptr(), n, self.I, self.IP, self.gemm_prec, st)
        return out.view(n, self.IP)

    def logits_rows(self, R):
        return self.ws["logits"][:R * self.IP].view(R, self.IP)[:, :self.I]

    def mu_logsigma(self, R):
        t = self.ws["tmp"][:R * 2 * L].view(R, 2 * L)
        return t[:, :L], t[:, L:]

    # ------------------------------------------------------------------ backward
    def backward(self, dlogits_ready=True, dmu=None, dls=None):
        """Gradients of the batched loss w.r.t. every parameter into ``self.grads``.

        Expects ``forward(..., want_grad=True)`` before it (ws['logits'] holds dlogits)."""
        if not self._in_step:
            self._settle()
        cur = torch.cuda.current_stream()
        with _Launch(cur):
            self._backward(cur)

    def _backward(self, cur):
        plan, anneal = self._plan, self._anneal
        R, I, IP = plan.R, self.I, self.IP
        w, st = self.ws, _stream()
        lay = self.layout
        side = None
        if self._last_fused:
            if plan.occ_off is None:
                raise ValueError("fused mode needs build_plan(..., total_items=I)")
            if not self._fused_grad:
                raise ValueError("backward() needs forward(..., want_grad=True)")
            Rp = (R + 63) // 64 * 64

            def proj_dw(slabs):
                args = (self.w3s.data_ptr(), w["h2b"].data_ptr(), _lib.ptr(w["h2tb"]), R, Rp, I, IP,
                        w["d1"].data_ptr(), plan.w_nll.data_ptr(), plan.occ_off.data_ptr(), plan.occ_row.data_ptr(),
                        plan.occ_items.data_ptr(), plan.nocc_long, slabs.data_ptr(),
                        self.Gv("decoder.linear2.weight").data_ptr(),
                        self.Gv("decoder.linear2.bias").data_ptr()) + self._proj_sched(plan, "items")
                self._split_call("svae_proj_nll_bwd_dw", args, ((1, "proj_dw_fused"), (2, "proj_dw_finish")),
                                 strides=(self._ps("rows"), self._ps("items")))

            if self.overlap:
                side = self._side_stream(cur=cur)
            else:
                proj_dw(w["pslabs_side"])
                self._bucket(0)
        else:
            dl = w["logits"]
            self._gemm(dl, w["d1"], self.Gv("decoder.linear2.weight"), IP, DP, R, IP, DP, DP, 1, 1, tag="proj_dw")
            self._colsum(dl, self.Gv("decoder.linear2.bias"), R, I, IP)
            self._bucket(0)
            self._gemm(dl, self.P("decoder.linear2.weight"), w["dd1"], R, DP, IP, IP, DP, DP, 0, 1,
                       act=2, aux=w["d1"], ldaux=DP, tag="proj_dx")
        # data-gradient chain: decoder.linear1 -> latent -> linear1 -> encoder.linear1
        if self.fused and self._dtmp_extra is None:
            ev = self._tic("mlp_bwd")
            _lib.call("svae_mlp_bwd_prec", w["dd1"].data_ptr(), w["tmp"].data_ptr(), w["eps"].data_ptr(),
                      w["enc"].data_ptr(), plan.w_kl.data_ptr(), self.mlp_frags.data_ptr(), float(anneal),
                      w["dtmp"].data_ptr(), w["denc"].data_ptr(), w["dhs"].data_ptr(), R, self.prec, st)
            self._toc("mlp_bwd", ev)
            self._stamp("q1 MLP backward done")
        else:
            self._gemm(w["dd1"], self.P("decoder.linear1.weight"), w["dz"], R, L, DP, DP, L, L, 0, 1)
            _lib.call("svae_latent_bwd", w["tmp"].data_ptr(), w["eps"].data_ptr(), w["dz"].data_ptr(),
                      plan.w_kl.data_ptr(), float(anneal), w["dtmp"].data_ptr(), R, st)
            if self._dtmp_extra is not None:          # dense surface: dL/dmu, dL/dlog_sigma arriving from VAELoss
                w["dtmp"][:R * 2 * L].add_(self._dtmp_extra.reshape(-1))
            self._gemm(w["dtmp"], self.P("linear1.weight"), w["denc"], R, DP, 2 * L, 2 * L, DP, DP, 0, 1,
                       act=2, aux=w["enc"], ldaux=DP)
            self._gemm(w["denc"], self.P("encoder.linear1.weight"), w["dhs"], R, HP, DP, DP, HP, HP, 0, 1)

        def decoder_side_wgrads(slabs):
            self._wgrad(w["dd1"], w["z"], "decoder.linear1.weight", R, DP, L, DP, L, slabs=slabs)
            self._wgrad(w["dtmp"], w["enc"], "linear1.weight", R, 2 * L, DP, 2 * L, DP, slabs=slabs)
            self._wgrad(w["denc"], w["hs"], "encoder.linear1.weight", R, DP, HP, DP, HP, slabs=slabs)

        if side is None:
            decoder_side_wgrads(self.slabs)
        else:
            chain_done = self._sev("chain_done")
            chain_done.record(cur)
        # GRU (BPTT)
        prepped = self.fused and self._prepped_grad
        if not prepped:
            _lib.call("svae_transpose", self.P("gru.weight_hh_l0").data_ptr(), self.whh_t.data_ptr(), G, HP, st)
        ev = self._tic("gru_bwd")
        # bf16 kernels: ONE gradient array [R][896] = (dr, dz, dn, dn*r) instead of dgi and dgh (they differ in the third
        # gate only): a third fewer stores in every BPTT step
        merged = self.fused and self.rowgemm and self.gru_prec in (PREC_BF16, PREC_SPLIT) and self.merged_dg
        # ... and as bf16 elements: its three MFMA consumers round it to bf16 anyway (RNE there as here), so the weight
        # and embedding gradients are bit-identical and the array is written once and read three times at half the
        # bytes; only the two GRU bias gradients (column sums of the array) see the rounding, 2e-4 relative
        # (tests/test_gpu_fullsize.py::test_gru_gradient_array_forms_agree)
        dg16 = merged and self.dg_bf16 and prepped and self.gru_prec == PREC_BF16     # (split: fp32 elements)
        dgi, ldg = (w["dg4b"] if dg16 else w["dg4"], 4 * HP) if merged else (w["dgi"], G)
        dgh, hh_remap = (dgi, (2 * HP, HP)) if merged else (w["dgh"], None)
        # Two-range BPTT (as the forward's two ranges, the other way round): the recurrence runs from the last timestep
        # down, so the rows of the timesteps >= tb are final first -- their share of the GRU weight gradients and of
        # dL/demb runs on the side stream while the BPTT of the steps < tb is still going.  OFF by default
        # (SVAE_BPTT_SPLIT=1): correct (tests/test_gpu_fullsize.py), and 5 % slower, 0.897 -> 0.947 ms per step -- a
        # kernel trace shows why: two BPTT launches cost 131 us where one costs 108 (each loads the 300 KB of W_hh
        # fragments and ramps up), the side stream is still busy with the dense-layer gradients when the late rows
        # become final, and four half-size weight-gradient launches + their reduces take longer than two whole ones.
        tb = rb = 0
        if (dg16 and side is not None and self.bptt_split and self.gru_wgrad_side and plan.split_t > 0
                and self.fused and self.rowgemm and prepped):
            tb = int(plan.split_t)
            rb = (int(plan._row_off_host[tb]) + 63) // 64 * 64
            if rb < 2048 or R - rb < 2048:
                tb = rb = 0
        if tb:
            if self.dh_carry is None or self.dh_carry.numel() < plan.B * HP:
                self.dh_carry = torch.zeros(max(plan.B, 256) * HP, device=self.device, dtype=torch.float32)
            gargs = (w["dhs"].data_ptr(), w["hprev"].data_ptr(), w["gates4"].data_ptr(), self.whh_t.data_ptr(),
                     plan.row_off.data_ptr(), plan.nact.data_ptr(), plan.B, plan.Tmax)
            _lib.call("svae_gru_bwd_range_bf16out", *gargs, tb, plan.Tmax, dgi.data_ptr(), self.dh_carry.data_ptr(), st)
            late_done = self._sev("late_done")
            late_done.record(cur)
            _lib.call("svae_gru_bwd_range_bf16out", *gargs, 0, tb, dgi.data_ptr(), self.dh_carry.data_ptr(), st)
        elif dg16:
            _lib.call("svae_gru_bwd_bf16out", w["dhs"].data_ptr(), w["hprev"].data_ptr(), w["gates4"].data_ptr(),
                      self.whh_t.data_ptr(), plan.row_off.data_ptr(), plan.nact.data_ptr(), plan.B, plan.Tmax,
                      dgi.data_ptr(), st)
        else:
            _lib.call("svae_gru_bwd", w["dhs"].data_ptr(), w["hprev"].data_ptr(), w["gates4"].data_ptr(),
                      self.whh_t.data_ptr(), plan.row_off.data_ptr(), plan.nact.data_ptr(), plan.B, plan.Tmax,
                      dgi.data_ptr(), None if merged else dgh.data_ptr(), self.gru_prec, st)
        self._toc("gru_bwd", ev)
        self._stamp("q1 BPTT done")
        if side is not None:
            # gated on the chain above, enqueued after the recurrence so that its few workgroups get their CUs
            # first; the weight-gradient kernels then take whatever the recurrence leaves idle
            _lib.stream_wait(side, chain_done)
            floating = self._in_step and self.float_w3 and self._last_fused
            if floating:
                w3 = self._side_stream(1, cur=cur)
                # the sweep recomputes the logits from d1, W3 and the row statistics of the forward sweep: it needs
                # nothing of the MLP backward in front of `chain_done`
                _lib.stream_wait(w3, self._sev("fwd_done") if self.dw_early else chain_done)
                with _Launch(w3, self._torch_ctx()):
                    self._stamp("w3 weight-gradient sweep may start")
                    proj_dw(w["pslabs_side"])
                    self._stamp("w3 weight-gradient sweep + finish done")
                    # (its bucket's all-reduce is issued by train_step AFTER the other buckets': collectives run in
                    # issue order on RCCL's one stream, and this one, which waits for the sweep, would hold up the
                    # embedding bucket that is ready a hundred microseconds earlier)
                self._float_pending = True
            with _Launch(side, self._torch_ctx()):
                if not floating:
                    proj_dw(w["pslabs_side"])
                    self._bucket(0)
                self._stamp("side dense weight gradients may start")
                if self._bf16_wgrads(dg16):
                    # the bf16 copy of the previous hidden states for the all-bf16 weight-gradient kernels: here, beside
                    # the BPTT, where this stream has time (4 us; the kernels round to bf16 anyway: same products)
                    _lib.call("svae_f32_to_bf16", w["hprev"].data_ptr(), w["hprev16"].data_ptr(), R * HP, _stream())
                decoder_side_wgrads(self.slabs_side)
                self._stamp("side dense weight gradients done")
                if self._loss_deferred:
                    # last on this stream: a one-workgroup kernel that waited 40 us for a free CU beside the sweep
                    # when it came first, with the weight gradients queued behind it.  (Later still -- behind the
                    # side stream's join point, the caller's stream waiting for it only at the end of train_step, so
                    # that it no longer sits in front of the GRU weight gradients -- measured 0.7 % SLOWER, 0.8728 vs
                    # 0.8666 ms per step over three alternating pairs (tools/ab_env.sh): a kernel trace, where this
                    # kernel shows as 8 - 25 us on the chain into the optimizer, is no guide to the untraced step.
                    #  Round 3, on the caller's stream behind the scatter-add, where that stream waits ~20 us for this one
                    #  anyway: 0.6 % slower again, 0.8434 vs 0.8386 ms, two alternating pairs.)
                    _lib.call("svae_loss_reduce", w["nll_row"].data_ptr(), w["kl_row"].data_ptr(),
                              plan.w_nll.data_ptr(), plan.w_kl.data_ptr(), float(anneal), R,
                              self.loss_out.data_ptr(), _stream())
        # Order of the tail: the embedding gradient first.  Its bucket is the largest of the step (256 I floats) and
        # its all-reduce then runs under the GRU weight-gradient kernels; the GRU bucket that ends the step is 1.3 MB.
        def embedding_grad():
            if self.fused and self.rowgemm and prepped:
                ev = self._tic("demb_gemm")
                if dg16:
                    _lib.call("svae_rowgemm_bf16x", dgi.data_ptr(), ldg, self.wih_frags[1].data_ptr(),
                              w["demb"].data_ptr(), R, st)
                else:
                    _lib.call("svae_rowgemm_ld_prec", dgi.data_ptr(), ldg, self.wih_frags[1].data_ptr(), None,
                              w["demb"].data_ptr(), R, 1, self.prec, st)
                self._toc("demb_gemm", ev)
            else:
                self._gemm(dgi, self.P("gru.weight_ih_l0"), w["demb"], R, E, G, ldg, E, E, 0, 1)
            # embedding: dense gradient, zero + deterministic segmented scatter-add
            ge = self.Gv("item_embed.weight")
            if not prepped:
                _lib.call("svae_zero", ge.data_ptr(), 4 * ge.numel(), st)
            self._prepped_grad = False
            _lib.call("svae_embed_bwd", w["demb"].data_ptr(), plan.perm.data_ptr(), plan.seg_off.data_ptr(),
                      plan.seg_item.data_ptr(), ge.data_ptr(), plan.nseg, plan.nseg_long, 1.0, st)
            gs = self.grad_sync
            if gs is not None and gs.sparse is not None and gs.sparse["bucket"] == 2:
                # large catalogues: the gradient's non-zero rows (one per distinct input item) as (id, row) pairs
                # (torch ops: this stream is torch's current stream here)
                gs.rows_ready(plan.seg_item, plan.nseg)
            else:
                self._bucket(2)

        def gru_wgrads():
            # (Round 1: these two on a THIRD stream beside dL/demb -> scatter-add were 2.5 % slower, 1.153 vs 1.123 ms per
            # step.  Round 2: on the existing side stream, behind the dense-layer gradients, +1.4 %, 0.925 -> 0.911 ms: a
            # kernel trace shows the caller's queue busy 898 of a step's 971 us -- it is the critical path, and by the
            # end of the BPTT the side stream has been idle for ~70 us.)
            ns = max(1, min(self.nsplit, (R + 127) // 128))
            if self._bf16_wgrads(dg16) and not self.wgrad_big and side is not None and not tb and ns > 1:
                o_hh = self._gbase + 4 * self.layout.offset("gru.weight_hh_l0")
                o_ih = self._gbase + 4 * self.layout.offset("gru.weight_ih_l0")
                s_hh, s_ih = self.slabs_hh.data_ptr(), self.slabs.data_ptr()
                st_hh, st_ih = G * HP + G, G * E + G
                ev = self._tic("wgrad")
                _lib.call("svae_wgrad_pair_bf16", dgi.data_ptr(), R, G, ldg, ns,
                          w["hprev16"].data_ptr(), HP, HP, s_hh, s_hh + 4 * G * HP, st_hh, hh_remap[0], hh_remap[1],
                          w["emb16"].data_ptr(), E, E, s_ih, s_ih + 4 * G * E, st_ih, (G + 7) // 8 * 8, 0, _stream())
                _lib.call("svae_reduce_slabs_pair", s_hh, o_hh, st_hh, st_hh, s_ih, o_ih, st_ih, st_ih, ns, _stream())
                self._toc("wgrad", ev)
                return
            if self._bf16_wgrads(dg16) and self.wgrad_big and side is not None and not tb:
                # both on 256 x 256 tiles, all operands bf16 (svae_wgrad_pair_big: 36 us against 48 alone at 21.8k rows)
                nb = max(2, min(self.nsplit_big, (R + 255) // 256))
                o_hh = self._gbase + 4 * self.layout.offset("gru.weight_hh_l0")
                o_ih = self._gbase + 4 * self.layout.offset("gru.weight_ih_l0")
                s_hh, s_ih = self.slabs_hh.data_ptr(), self.slabs.data_ptr()
                st_hh, st_ih = G * HP + G, G * E + G
                ev = self._tic("wgrad")
                _lib.call("svae_wgrad_pair_big", dgi.data_ptr(), R, G, ldg, nb,
                          w["hprev16"].data_ptr(), HP, HP, s_hh, s_hh + 4 * G * HP, st_hh, hh_remap[0], hh_remap[1],
                          w["emb16"].data_ptr(), E, E, s_ih, s_ih + 4 * G * E, st_ih, (G + 7) // 8 * 8, 0, _stream())
                _lib.call("svae_reduce_slabs_pair", s_hh, o_hh, st_hh, st_hh, s_ih, o_ih, st_ih, st_ih, nb, _stream())
                self._toc("wgrad", ev)
                return
            if self.wgrad_pair and dg16 and self.fused and ns > 1:
                # both in ONE launch and their slabs summed by ONE reduce (svae_wgrad_pair_bf16dy): the same workgroups
                # writing the same slabs, one kernel boundary and one reduce less on the tail's side chain
                e16 = bool(self._emb16_live)
                xi = w["emb16"] if e16 else w["emb"]
                o_hh = self._gbase + 4 * self.layout.offset("gru.weight_hh_l0")
                o_ih = self._gbase + 4 * self.layout.offset("gru.weight_ih_l0")
                s_hh, s_ih = self.slabs_hh.data_ptr(), self.slabs.data_ptr()
                st_hh, st_ih = G * HP + G, G * E + G
                ev = self._tic("wgrad")
                _lib.call("svae_wgrad_pair_bf16dy", dgi.data_ptr(), R, G, ldg, ns,
                          w["hprev"].data_ptr(), HP, HP, s_hh, s_hh + 4 * G * HP, st_hh, hh_remap[0], hh_remap[1],
                          xi.data_ptr(), int(e16), E, E, s_ih, s_ih + 4 * G * E, st_ih, (G + 7) // 8 * 8, 0, _stream())
                _lib.call("svae_reduce_slabs_pair", s_hh, o_hh, st_hh, st_hh, s_ih, o_ih, st_ih, st_ih, ns, _stream())
                self._toc("wgrad", ev)
                return
            self._wgrad(dgh, w["hprev"], "gru.weight_hh_l0", R, G, HP, ldg, HP, remap=hh_remap, dy_bf16=dg16)
            if self._emb16_live:
                if not dg16:
                    raise ValueError("bf16 embedding rows need the bf16 gradient array (forward with want_grad)")
                self._wgrad(dgi, w["emb16"], "gru.weight_ih_l0", R, G, E, ldg, E, dy_bf16=True, x_bf16=True)
            else:
                self._wgrad(dgi, w["emb"], "gru.weight_ih_l0", R, G, E, ldg, E, dy_bf16=dg16)

        if tb:
            self._tail_two_ranges(plan, cur, side, late_done, dgi, ldg, rb, hh_remap)
        elif side is not None and self.gru_wgrad_side:
            # the two GRU weight gradients behind the dense-layer ones on the side stream (idle by the time the BPTT
            # ends), beside dL/demb -> scatter-add on the caller's: the caller's queue is the step's critical path
            # (host order: the caller's kernels first -- its queue must not run dry while the side launches are made)
            bptt_done = self._sev("bptt_done")
            bptt_done.record(cur)
            embedding_grad()
            if self._early_opt is not None:
                # train_step, one process: the optimizer over the embedding table (60 % of the parameters, 26 us of
                # HBM traffic) here, under the side stream's GRU weight gradients, instead of behind the join with
                # them -- the side stream is the longer of the two by then (tools/skip_sensitivity.sh: the GRU weight
                # gradients hold 56 us of the step, dL/demb + scatter-add 14)
                (s_, e_), gscale_, count_ = self._early_opt
                self._adam_range(s_, e_, gscale_, st, count=count_)
                self._early_opt = False          # (done: train_step leaves this bucket out)
            self._stamp("q1 dL/demb + scatter-add done")
            _lib.stream_wait(side, bptt_done)
            with _Launch(side, self._torch_ctx()):
                gru_wgrads()          # (one of the two here and one on the caller's stream: no better, 0.922 vs 0.918;
                                      #  round 3, both on a THIRD stream so that they do not queue behind the dense-layer
                                      #  gradients and the loss reduce, which end ~20 us after the BPTT: 0.8278 vs 0.8285,
                                      #  nothing -- the tail is short of CUs, not of ordering freedom)
                self._stamp("side GRU weight gradients done")
        elif self.tail_embedding_first:
            embedding_grad()
            gru_wgrads()
        else:
            gru_wgrads()
            embedding_grad()
        if side is not None:
            joined = self._sev("side_joined")
            joined.record(side)
            _lib.stream_wait(cur, joined)
            self._stamp("q1 joined the side stream")
        elif self._loss_deferred:
            _lib.call("svae_loss_reduce", w["nll_row"].data_ptr(), w["kl_row"].data_ptr(), plan.w_nll.data_ptr(),
                      plan.w_kl.data_ptr(), float(anneal), R, self.loss_out.data_ptr(), st)
        self._loss_deferred = False
        self._bucket(1)      # dense layers (side stream) + GRU (this stream): behind the join

    def _bf16_wgrads(self, dg16):
        """The GRU's weight gradients take an all-bf16 kernel (svae_wgrad_pair_bf16, or svae_wgrad_pair_big on request):
        bf16 gradient array, bf16 embedding rows, the overlapped step (the bf16 copy of the hidden states is made on the
        side stream beside the BPTT)."""
        return bool((self.wgrad16 or self.wgrad_big) and dg16 and self._emb16_live and self.fused and self.overlap
                    and self.wgrad_pair and self.gru_wgrad_side and not self.bptt_split)

    def _tail_two_ranges(self, plan, cur, side, late_done, dg, ldg, rb, hh_remap):
        """Tail of backward with a two-range BPTT (bf16 gradient array dg [R][ldg]): rows >= rb were final at
        `late_done` (recorded on `cur` between the two BPTT launches), the others are when `cur` reaches this point.
        Row ranges of one weight gradient write disjoint slab sets that ONE reduce sums at the end."""
        w, R = self.ws, plan.R
        ns = max(2, min(self.nsplit, (R + 127) // 128))
        ns_a = max(1, min(ns - 1, int(round(ns * (R - rb) / float(R)))))
        if self.slabs_hh is None:
            self.slabs_hh = torch.zeros(self.nsplit * (G * HP + G), device=self.device, dtype=torch.float32)
        jobs = (("gru.weight_hh_l0", w["hprev"], HP, self.slabs_hh, hh_remap),
                ("gru.weight_ih_l0", w["emb"], E, self.slabs, ((G + 7) // 8 * 8, 0)))

        def wgrads(r0, r1, z0, nz):
            for name, X, K, slabs, (rf, ra) in jobs:
                stride = G * K + G
                sp = slabs.data_ptr() + 4 * stride * z0
                ev = self._tic("wgrad")
                _lib.call("svae_wgrad_cols_bf16dy", dg.data_ptr() + 2 * ldg * r0, X.data_ptr() + 4 * K * r0, sp,
                          sp + 4 * G * K, r1 - r0, G, K, ldg, K, nz, stride, rf, ra, _stream())
                self._toc("wgrad", ev)

        def demb(r0, r1):
            ev = self._tic("demb_gemm")
            _lib.call("svae_rowgemm_bf16x", dg.data_ptr() + 2 * ldg * r0, ldg, self.wih_frags[1].data_ptr(),
                      w["demb"].data_ptr() + 4 * E * r0, r1 - r0, _stream())
            self._toc("demb_gemm", ev)

        early_done = self._sev("early_done")
        early_done.record(cur)
        demb(0, rb)                                   # the caller's stream: rows of the early timesteps
        _lib.stream_wait(side, late_done)
        with _Launch(side, self._torch_ctx()):
            demb(rb, R)
            late_demb = self._sev("late_demb")
            late_demb.record(side)
            wgrads(rb, R, 0, ns_a)
        _lib.stream_wait(cur, late_demb)
        ge = self.Gv("item_embed.weight")
        self._prepped_grad = False
        _lib.call("svae_embed_bwd", w["demb"].data_ptr(), plan.perm.data_ptr(), plan.seg_off.data_ptr(),
                  plan.seg_item.data_ptr(), ge.data_ptr(), plan.nseg, plan.nseg_long, 1.0, _stream())
        gs = self.grad_sync
        if gs is not None and gs.sparse is not None and gs.sparse["bucket"] == 2:
            gs.rows_ready(plan.seg_item, plan.nseg)
        else:
            self._bucket(2)
        _lib.stream_wait(side, early_done)
        with _Launch(side, self._torch_ctx()):
            wgrads(0, rb, ns_a, ns - ns_a)
            for name, X, K, slabs, _ in jobs:
                stride = G * K + G
                _lib.call("svae_reduce_slabs", slabs.data_ptr(), self._gbase + 4 * self.layout.offset(name), stride, ns,
                          stride, _stream())

    # ------------------------------------------------------------------ dense (notebook) surface
    def dense_loss(self, logits, mu, ls, tgt, anneal, batch_size=1.0):
        """VAELoss.forward (nb:795-812) on dense logits [R, I]: returns (loss 0-dim tensor,
        dL/dlogits [R, I], dL/d(mu|log_sigma) [R, 128]).  w_nll = 1/(batch_size*num_ones) with num_ones =
        the first row's target count (nb:803), KL = mean over rows (nb:797)."""
        R, I, IP, st = logits.shape[0], self.I, self.IP, _stream()
        f32 = dict(device=self.device, dtype=torch.float32)
        buf = torch.zeros(R * IP, **f32)
        buf.view(R, IP)[:, :I].copy_(logits)
        tgt = tgt.contiguous()
        K = tgt.shape[1]
        n0 = (tgt[0] >= 0).sum().to(torch.float32)
        w_nll = (1.0 / (float(batch_size) * n0)).expand(R).contiguous()
        w_kl = torch.full((R,), 1.0 / R, **f32)
        tmp = torch.cat([mu, ls], dim=1).contiguous()
        nll_row, kl_row, lse = torch.empty(R, **f32), torch.empty(R, **f32), torch.empty(R, **f32)
        zeros = torch.zeros(R * L, **f32)
        scratch = torch.empty(R * L, **f32)
        dtmp = torch.empty(R, 2 * L, **f32)
        out = torch.zeros(4, **f32)
        _lib.call("svae_softmax_nll", buf.data_ptr(), tgt.data_ptr(), w_nll.data_ptr(), nll_row.data_ptr(),
                  lse.data_ptr(), R, I, IP, K, 1, st)
        _lib.call("svae_latent_fwd", tmp.data_ptr(), zeros.data_ptr(), scratch.data_ptr(), kl_row.data_ptr(), R, st)
        _lib.call("svae_loss_reduce", nll_row.data_ptr(), kl_row.data_ptr(), w_nll.data_ptr(), w_kl.data_ptr(),
                  float(anneal), R, out.data_ptr(), st)
        _lib.call("svae_latent_bwd", tmp.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), w_kl.data_ptr(),
                  float(anneal), dtmp.data_ptr(), R, st)
        return out[0].clone(), buf.view(R, IP)[:, :I], dtmp

    def backward_dense(self, plan, dlogits, dmu, dls):
        """Backward of Model.forward given dense dL/d dec_out [R, I] and dL/dmu, dL/dlog_sigma [R, 64]
        (what autograd hands to the notebook's Model at loss.backward(), nb:1205)."""
        R = plan.R
        self._ensure_logits()
        buf = self.ws["logits"][:R * self.IP].view(R, self.IP)
        buf.zero_()
        buf[:, :self.I].copy_(dlogits)
        self._plan, self._anneal, self._last_fused = plan, 0.0, False
        self._dtmp_extra = torch.cat([dmu, dls], dim=1).contiguous()
        try:
            self.backward()
        finally:
            self._dtmp_extra = None

    # ------------------------------------------------------------------ optimizer
    def adam_step(self, gscale=1.0):
        self._settle()
        self._w3_fresh = False
        self.step_count += 1
        self._adam_range(0, self.layout.total, gscale, _stream())

    OPTIMIZERS = {"adam": None, "adagrad": 0, "adadelta": 1, "rmsprop": 2}

    def set_optimizer(self, kind, lr, weight_decay, decay=0.0, eps=1e-8, betas=(0.9, 0.999)):
        """Selects what optimizer.step() runs (nb:1252-1267) and clears its state.  decay = Adagrad's lr_decay,
        Adadelta's rho, RMSprop's alpha; exp_avg_sq holds state_sum / square_avg, exp_avg holds acc_delta."""
        if kind not in self.OPTIMIZERS:
            raise ValueError("unknown optimizer %r (the notebook knows adagrad, adadelta, adam, rmsprop)" % (kind,))
        self.reset_optimizer()
        self.optimizer, self.lr, self.weight_decay = kind, float(lr), float(weight_decay)
        if kind == "adam":
            self.betas, self.adam_eps, self._opt = betas, float(eps), None
        else:
            self._opt = (self.OPTIMIZERS[kind], float(decay), float(eps))

    def _adam_range(self, s, e, gscale, st, count=None):
        """The optimizer over elements [s, e) of the flat buffers (Adam unless set_optimizer chose another); count =
        the optimizer step's number (default: step_count, which the caller has advanced)."""
        count = self.step_count if count is None else count
        if self._opt is not None:
            kind, decay, eps = self._opt
            _lib.call("svae_optimizer_step", kind, self.params.data_ptr() + 4 * s, self.grads.data_ptr() + 4 * s,
                      self.exp_avg_sq.data_ptr() + 4 * s, self.exp_avg.data_ptr() + 4 * s, e - s, count,
                      self.lr, decay, eps, self.weight_decay, float(gscale), st)
            return
        _lib.call("svae_adam_step", self.params.data_ptr() + 4 * s, self.grads.data_ptr() + 4 * s,
                  self.exp_avg.data_ptr() + 4 * s, self.exp_avg_sq.data_ptr() + 4 * s, e - s, count, self.lr,
                  self.betas[0], self.betas[1], self.adam_eps, self.weight_decay, float(gscale), st)

    def train_step(self, plan, anneal, eps=None, seed=0, step=None):
        """forward + backward + Adam for one batch.  Returns the device tensor holding
        [loss, likelihood, KL]; reading it synchronises, so do that only when logging."""
        self._in_step = True
        gs = self.grad_sync
        gscale = gs.gscale if gs is not None else 1.0
        emb_bucket = self.layout.buckets()[2]
        # one process: backward() runs the optimizer over the embedding bucket as soon as that gradient is complete
        self._early_opt = ((emb_bucket, gscale, self.step_count + 1)
                           if gs is None and self.early_embed_opt and self.float_w3 else None)
        try:
            self.forward(plan, anneal, eps=eps, seed=seed, step=self.step_count if step is None else step,
                         defer_loss=True)
            self.backward()
        finally:
            self._in_step = False
        early_done = self._early_opt is False
        self._early_opt = None
        sharded = gs is not None and gs.active and gs.sharded is not None and gs.sharded["bucket"] == 0
        if not self._float_pending:
            if gs is not None:
                gs.wait()
            if sharded:
                # decoder.linear2 bucket: the optimizer over this rank's shard only, the parameters all-gathered
                self._settle()
                self._w3_fresh = False
                self.step_count += 1
                (s0, e0), a_b = self.layout.buckets()[0], gs.shard_range()
                self._adam_range(a_b[0], a_b[1], gscale, _stream())
                self._adam_range(e0, self.layout.total, gscale, _stream())
                wk = gs.gather_params(self.params)
                if wk is not None:
                    wk.wait()
            elif early_done:
                self._settle()
                self._w3_fresh = False
                self.step_count += 1
                self._adam_range(0, emb_bucket[0], gscale, _stream())
            else:
                self.adam_step(gscale)
            return self.loss_out
        # floated decoder.linear2 bucket: Adam over the other buckets here, over that one on its own stream
        self._float_pending = False
        self.step_count += 1
        (s0, e0), rest = self.layout.buckets()[0], self.layout.buckets()[1:]
        cur = torch.cuda.current_stream()
        w3 = self._side_stream(1, cur=cur)
        if gs is not None:
            with _Launch(w3, True):
                gs.bucket_ready(0)
            # the embedding bucket's all-reduce was issued first: its optimizer step runs under the dense + GRU
            # bucket's collective
            gs.wait(which=[2])
            self._adam_range(rest[1][0], rest[1][1], gscale, cur.cuda_stream)
            gs.wait(which=[1])
            self._adam_range(rest[0][0], rest[0][1], gscale, cur.cuda_stream)
        elif early_done:
            self._adam_range(rest[0][0], rest[0][1], gscale, cur.cuda_stream)
        else:
            self._adam_range(rest[0][0], rest[-1][1], gscale, cur.cuda_stream)
            self._stamp("q1 optimizer done", cur.cuda_stream)
        with _Launch(w3, gs is not None):
            if gs is not None:
                gs.wait(which=[0])
            if sharded:
                a_, b_ = gs.shard_range()
                self._adam_range(a_, b_, gscale, _stream())
                wk = gs.gather_params(self.params)
                if wk is not None:
                    wk.wait()
            else:
                self._adam_range(s0, e0, gscale, _stream())
            self._proj_prep_items(_stream())
            self._stamp("w3 optimizer + operand copies of decoder.linear2 done")
        self._w3_event = self._sev("w3")
        self._w3_event.record(w3)
        self._w3_fresh = True
        self._float_plan = plan
        return self.loss_out

    # ------------------------------------------------------------------ scoring
    def score(self, plan, topk=100, eps=None, seed=0, step=0, slot=0):
        """evaluate()'s per-user work (nb:615-636): loss at anneal 0.2 over all rows, then the LAST
        timestep's logits with the user's input items masked, ranked top-k.
        Returns (loss_out, idx [B, k] int32, val [B, k]) in the plan's SORTED user order.
        slot = 1 (inside ``scoring_session`` only): the call works in the engine's second workspace, so that it may be
        issued on another stream while a slot-0 call is still running; its loss_out is the slot's own."""
        if slot not in (0, 1, 2) or (slot != 0 and not self._session):
            raise ValueError("score(slot=1 | 2) needs an open scoring_session()")
        self._use_slot(0)
        self._ensure(plan.R)                      # (the workspace grows on slot 0; slot 1 follows its size)
        self._use_slot(slot)
        try:
            return self._score(plan, topk, eps, seed, step)
        finally:
            self._use_slot(0)

    def _score(self, plan, topk, eps, seed, step):
        self.forward(plan, 0.2, eps=eps, seed=seed, step=step, want_grad=False)
        B, st = plan.B, _stream()
        k = int(min(topk, self.I))
        f32 = dict(device=self.device, dtype=torch.float32)
        last = self.dense_logits(plan.last_row)          # [B, IP]: only the ranked rows are projected densely
        work = torch.empty(B * self.IP, **f32)
        idx = torch.empty(B * k, device=self.device, dtype=torch.int32)
        val = torch.empty(B * k, **f32)
        _lib.call("svae_masked_topk", last.data_ptr(), self.IP, plan.seen.data_ptr(), plan.seen_off.data_ptr(),
                  B, self.I, k, idx.data_ptr(), val.data_ptr(), work.data_ptr(), st)
        return self.loss_out, idx.view(B, k), val.view(B, k), last[:, :self.I]
