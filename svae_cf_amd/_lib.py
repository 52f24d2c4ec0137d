"""ctypes binding of the C-ABI library ``csrc/libsvae_hip.so`` (declared in include/svae_hip.h).

The product path has NO CPU fallback: if the HIP library is missing, or a call
returns non-zero, this module raises.  Nothing under ``oracle/`` is imported here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsvae_hip.so")

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_F = _c.c_float
_LL = _c.c_longlong
_ULL = _c.c_ulonglong
_U = _c.c_uint

# name -> argtypes (all return int: 0 ok, <0 argument error, >0 hipError_t)
SIGNATURES = {
    "svae_gemm": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _LL, _P],
    "svae_logits_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "svae_reduce_slabs": [_P, _P, _LL, _I, _LL, _P],
    "svae_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _LL, _P],
    "svae_gru_fwd": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P],
    "svae_gru_fwd_range": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P],
    "svae_proj_prep_rows_range": [_P, _I, _I, _I, _I, _P, _P, _P, _P],
    "svae_gru_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P],
    "svae_embed_fwd": [_P, _P, _P, _I, _I, _P],
    "svae_embed_fwd_wih_prep": [_P, _P, _P, _I, _I, _P, _P, _I, _P],
    "svae_embed_fwd_bf16": [_P, _P, _P, _I, _I, _P],
    "svae_rowgemm_gi_bf16x": [_P, _I, _P, _P, _P, _I, _P],
    "svae_wgrad_cols_bf16dy_bf16x": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _LL, _I, _I, _P],
    "svae_wgrad_pair_big": [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _LL, _I, _I, _P, _I, _I, _P, _P, _LL, _I, _I, _P],
    "svae_f32_to_bf16": [_P, _P, _LL, _P],
    "svae_wgrad_pair_bf16": [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _LL, _I, _I, _P, _I, _I, _P, _P, _LL, _I, _I, _P],
    "svae_wgrad_pair_bf16dy": [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _LL, _I, _I, _P, _I, _I, _I, _P, _P, _LL, _I, _I, _P],
    "svae_reduce_slabs_pair": [_P, _P, _LL, _LL, _P, _P, _LL, _LL, _I, _P],
    "svae_embed_bwd": [_P, _P, _P, _P, _P, _I, _I, _F, _P],
    "svae_embed_long_rows": [],
    "svae_proj_occ_long": [],
    "svae_latent_fwd": [_P, _P, _P, _P, _I, _P],
    "svae_latent_bwd": [_P, _P, _P, _P, _F, _P, _I, _P],
    "svae_softmax_nll": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "svae_adam_step": [_P, _P, _P, _P, _LL, _I, _F, _F, _F, _F, _F, _F, _P],
    "svae_optimizer_step": [_I, _P, _P, _P, _P, _LL, _I, _F, _c.c_double, _F, _F, _F, _P],
    "svae_colsum": [_P, _P, _P, _I, _I, _I, _I, _P],
    "svae_transpose": [_P, _P, _I, _I, _P],
    "svae_loss_reduce": [_P, _P, _P, _P, _F, _I, _P, _P],
    "svae_gather_rows": [_P, _P, _P, _I, _I, _I, _I, _P],
    "svae_masked_topk": [_P, _I, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "svae_masked_topk_static": [_P, _I, _P, _P, _I, _I, _I, _P, _P, _P],
    "svae_sparse_rows_rebuild": [_P, _LL, _I, _P, _P, _I, _I, _P, _I, _P],
    "svae_heldout_ranks": [_P, _I, _I, _P, _P, _P, _P, _LL, _P, _P],
    "svae_philox_normal": [_P, _P, _I, _ULL, _U, _P],
    "svae_proj_prep_rows": [_P, _I, _I, _P, _P, _P, _P],
    "svae_proj_prep_items": [_P, _P, _I, _I, _P, _P, _P, _P],
    "svae_proj_nll_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _P],
    "svae_proj_nll_fwd_dh": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I,
                             _I, _P],
    "svae_proj_nll_bwd_dw": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "svae_proj_ns_max": [],
    "svae_mlp_frag_elems": [],
    "svae_mlp_prep": [_P, _P, _P, _P, _P],
    "svae_mlp_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "svae_mlp_fwd_prep": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "svae_mlp_bwd": [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _P],
    "svae_spin": [_I, _P],
    "svae_spin_grid": [_I, _I, _I, _P],
    "svae_stamp": [_P, _I, _P],
    "svae_rccl_unique_id": [_P],
    "svae_rccl_comm_create": [_P, _I, _I, _P],
    "svae_rccl_comm_destroy": [_P],
    "svae_allreduce_sum_f32": [_P, _P, _LL, _P],
    "svae_reduce_scatter_sum_f32": [_P, _P, _LL, _I, _P],
    "svae_allgather_f32": [_P, _P, _LL, _I, _P],
    "svae_event_create": [_P],
    "svae_sync_event_create": [_P],
    "svae_event_record": [_P, _P],
    "svae_stream_wait_event": [_P, _P],
    "svae_event_destroy": [_P],
    "svae_event_elapsed_ms": [_P, _P, _P],
    "svae_proj_time_next": [_P, _P],
    "svae_rowgemm_frag_elems": [_I],
    "svae_rowgemm_prep": [_P, _I, _P, _P],
    "svae_rowgemm": [_P, _P, _P, _P, _I, _I, _P],
    "svae_rowgemm_bf16x": [_P, _I, _P, _P, _I, _P],
    "svae_wgrad_cols_bf16dy": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _LL, _I, _I, _P],
    "svae_gru_bwd_bf16out": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P],
    "svae_gru_bwd_range_bf16out": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "svae_rowgemm_ld": [_P, _I, _P, _P, _P, _I, _I, _P],
    "svae_wgrad_cols": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _LL, _I, _I, _P],
    "svae_zero": [_P, _LL, _P],
    # split instantiations (prec 2) of the fused kernels
    "svae_wgrad_cols_prec": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _LL, _I, _I, _I, _P],
    "svae_rowgemm_prep_prec": [_P, _I, _P, _I, _P],
    "svae_rowgemm_ld_prec": [_P, _I, _P, _P, _P, _I, _I, _I, _P],
    "svae_mlp_prep_prec": [_P, _P, _P, _P, _I, _P],
    "svae_mlp_fwd_prec": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "svae_mlp_bwd_prec": [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _P],
    "svae_proj_prep_rows_range_split": [_P, _I, _I, _I, _I, _P, _P, _P, _LL, _P],
    "svae_proj_prep_items_split": [_P, _P, _I, _I, _P, _P, _P, _LL, _P],
    "svae_proj_nll_fwd_split": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _LL, _LL, _P],
    "svae_proj_nll_fwd_dh_split": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I,
                                   _I, _I, _LL, _LL, _P],
    "svae_proj_nll_bwd_dw_split": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I,
                                   _LL, _LL, _P],
}

_lib = None


class SvaeHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SVAE_HIP_LIB", LIB_PATH)      # explicit library path (kernel experiments)
    if not os.path.exists(path):
        raise SvaeHipError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C svae_cf_amd/csrc`).  There is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _I
    lib.svae_last_error.argtypes = []
    lib.svae_last_error.restype = _c.c_char_p
    # The same library once more through a handle that KEEPS the interpreter lock during a call (`call` below).
    global _lib_locked
    locked = ctypes.PyDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(locked, name)
        fn.argtypes = argtypes
        fn.restype = _I
    _lib_locked = locked
    _lib = lib
    return lib


_lib_locked = None
_fns = {}
# `call` is the launch path of the engine: entry points that enqueue a kernel or an event and return in a few
# microseconds.  Through ctypes.CDLL every one of them drops the interpreter lock and has to win it back -- with plan
# builder threads running beside the loop (batch.PlanPrefetcher) each of the ~50 launches of a step or ~17 of a scoring
# call then waits for a builder's Python stretch: measured in evaluate(), a scoring call's host time went from 0.12 ms
# alone to 0.2-0.4 ms with two or three builders (tools/probe/evaluate_workers.py), and the loop became host-bound.
# So launches keep the lock (ctypes.PyDLL); entry points that can block on the device or on other ranks do not.
_UNLOCKED_PREFIXES = ("svae_rccl_", "svae_allreduce_", "svae_allgather_", "svae_reduce_scatter_", "svae_event_elapsed",
                      "svae_plan_", "svae_proj_schedule")


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        lib = load()
        fn = _fns[name] = getattr(lib if name.startswith(_UNLOCKED_PREFIXES) else _lib_locked, name)
    rc = fn(*args)
    if rc != 0:
        msg = load().svae_last_error().decode("utf-8", "replace")
        raise SvaeHipError("%s failed (rc=%d): %s" % (name, rc, msg))


def install_skip(names):
    """Timing experiments only (tools/skip_sensitivity.sh, tools/bench_skip.py): replaces ``call`` by a variant that
    leaves the named entry points out, so that the step's sensitivity to each kernel can be read off the step time.
    Results are then wrong by construction -- the product path never calls this, and no environment variable does."""
    global call
    names, real = frozenset(names), call

    def call_skipping(name, *args):
        if name in names:
            return None
        return real(name, *args)
    import sys
    sys.stderr.write("svae_cf_amd._lib: SKIPPING %s -- timing experiment, results are wrong\n" % sorted(names))
    call = call_skipping


def proj_schedule(nx, ny, ncu, uniform=False):
    """Work division of one projection sweep (svae_proj_schedule): returns (int32 numpy array [nwg, 4], nwg, ns_q,
    ns_rem)."""
    import numpy as np
    lib = load()
    cap = lib.svae_proj_ns_max() * ((nx + 255) // 256)
    buf = np.zeros((cap, 4), dtype=np.int32)
    q, rem = _I(0), _I(0)
    fn = lib.svae_proj_schedule
    fn.argtypes = [_I, _I, _I, _I, _P, _I, _c.POINTER(_I), _c.POINTER(_I)]
    fn.restype = _I
    n = fn(nx, ny, ncu, int(uniform), buf.ctypes.data, cap, _c.byref(q), _c.byref(rem))
    if n <= 0:
        raise SvaeHipError("svae_proj_schedule failed (rc=%d): %s" % (n, lib.svae_last_error().decode("utf-8", "replace")))
    return buf[:n].copy(), n, q.value, rem.value


def proj_schedule_flat(nx, ny, nwg, boundary=5):
    """Flat work division (svae_proj_schedule_flat): returns (int32 numpy array [n4, 4] -- segments, then the
    workgroups' segment offsets, then the slab counts --, nwg, 0, nseg): what the sweeps take as (table, nwg, ns_q,
    ns_rem)."""
    import numpy as np
    lib = load()
    xt = (nx + 255) // 256
    cap = (int(nwg) + xt) + (int(nwg) + 4) // 4 + (xt + 3) // 4 + 4       # segments + the two tables
    buf = np.zeros((cap, 4), dtype=np.int32)
    w, nseg = _I(0), _I(0)
    fn = lib.svae_proj_schedule_flat
    fn.argtypes = [_I, _I, _I, _I, _P, _I, _c.POINTER(_I), _c.POINTER(_I)]
    fn.restype = _I
    n = fn(nx, ny, int(nwg), int(boundary), buf.ctypes.data, cap, _c.byref(w), _c.byref(nseg))
    if n <= 0:
        raise SvaeHipError("svae_proj_schedule_flat failed (rc=%d): %s" % (n, lib.svae_last_error().decode("utf-8", "replace")))
    return buf[:n].copy(), w.value, 0, nseg.value



class PlanInfo(_c.Structure):
    """struct svae_plan_info (include/svae_hip.h)."""
    _fields_ = [(n, _I) for n in ("B", "R", "Tmax", "K", "nseg", "nseg_long", "nocc", "nocc_long", "split_t", "split_r",
                                  "total_items", "reserved")] + [("total_bytes", _LL), ("off", _LL * 16)]


_plan_fns = None


def plan_fns():
    """(create, pack, destroy) of the host-side plan builder, argtypes set; ctypes drops the interpreter lock for the
    duration of each call, so worker threads build plans side by side."""
    global _plan_fns
    if _plan_fns is None:
        lib = load()
        cr, pk, ds = lib.svae_plan_create, lib.svae_plan_pack, lib.svae_plan_destroy
        cr.argtypes, cr.restype = [_P, _P, _P, _P, _I, _I, _I, _c.POINTER(PlanInfo)], _c.c_void_p
        pk.argtypes, pk.restype = [_c.c_void_p, _P, _LL, _P, _P, _P], _I
        ds.argtypes, ds.restype = [_c.c_void_p], None
        _plan_fns = (cr, pk, ds)
    return _plan_fns


def last_error():
    return load().svae_last_error().decode("utf-8", "replace")


class SyncEvent:
    """An ordering event (no timing) recorded and waited for through the library's thin HIP wrappers; the engine keeps
    one per purpose and re-records it every step (a wait refers to the record that precedes it in host order)."""
    __slots__ = ("h",)

    def __init__(self):
        a = _c.c_void_p()
        call("svae_sync_event_create", _c.byref(a))
        self.h = a.value

    def record(self, stream):
        """stream: a torch.cuda.Stream (or anything with .cuda_stream)."""
        call("svae_event_record", self.h, stream.cuda_stream)


def stream_wait(stream, ev):
    """Order `stream` behind `ev` (a SyncEvent or a torch.cuda.Event)."""
    if type(ev) is SyncEvent:
        call("svae_stream_wait_event", stream.cuda_stream, ev.h)
    else:
        stream.wait_event(ev)


class DispatchTimer:
    """A start / stop event pair attached to ONE kernel dispatch (svae_proj_time_next); quacks like the
    (start, stop) pairs of torch events the engine's profiling lists hold: ``pair[0].elapsed_time(pair[1])``."""

    def __init__(self):
        a, b = _c.c_void_p(), _c.c_void_p()
        call("svae_event_create", _c.byref(a))
        call("svae_event_create", _c.byref(b))
        self.start, self.stop = a.value, b.value

    def arm(self):
        call("svae_proj_time_next", self.start, self.stop)

    def elapsed_time(self, other=None):
        ms = _c.c_float()
        call("svae_event_elapsed_ms", self.start, self.stop, _c.byref(ms))
        return float(ms.value)


def ptr(t):
    """Raw device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
