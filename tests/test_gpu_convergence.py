"""Does TRAINING in the timed mode (PREC_BF16) reach the metrics of the fp32-class mode?  (north_star: "Recall/NDCG
within 0.5 % absolute ... matching reference loss and metrics"; round 2's review: every metric test scored GIVEN
parameters, none trained.)

Both modes train the same model (reference init, nb:685-732) from the same parameters over the same length-bucketed
batches with the same Philox noise, through the notebook's loop as this build mirrors it -- `train_batched` per epoch,
`evaluate` on 1 536 held-out users after every epoch (nb:1274-1307, nb:594-666) -- on ML-1M-shaped synthetic data
(3 483 items, 5 120 training users, batches of 64: 80 optimizer steps per epoch, 6 400 steps in all).  PREC_SPLIT is the
comparator: the same kernel templates at fp32-class precision, pinned to the reference at the fp32 gates
(test_gpu_split_parity.py), and over 320 steps within 2e-5 (parameters) of the exact-fp32 kernels
(tools/convergence.py --modes bf16,split,fp32; profiles/r03a_convergence.md).

What can be asked of two trajectories.  Measured with this set-up (profiles/r03a_convergence.md): the validation
metrics of ONE run move by +-0.5 NDCG@100 / +-0.7 Rec@100 from epoch to epoch on the plateau (standard deviation over 49
evaluations); two runs of the fp32-class mode that differ only in the seed of the latent noise sit 0.1 NDCG@100 /
0.9 Rec@100 apart in their plateau MEANS; and two fp32-class implementations that agree to 1e-6 per step (PREC_SPLIT with
the GRU through the split template, and with the generic fp32 GRU kernels) end 0.04 / 0.02 apart, their validation loss
1.1e-3.  A gate of 0.5 on single evaluations would fail between two fp32 runs.  So:
  * early phase (the first 320 steps, where the two trajectories still coincide): per-step training loss within
    2e-2 relative (measured 7e-3), validation loss within 3e-3 relative (1.4e-3), NDCG@100 and Rec@100 within 0.5
    absolute (0.08 / 0.015);
  * plateau (steps 2 560 .. 6 400, evaluated every epoch): the plateau means of bf16 and split within 1.0 absolute for
    each seed (the seed-to-seed spread of the fp32-class mode itself; measured +0.32 / +0.46 at seed 5, -0.14 / -0.05
    at seed 6) and within 0.5 absolute on the average of the two seeds (measured 0.09 / 0.21); validation loss within
    5e-3 relative (3.8e-3, 2.9e-3).
The kernels are bit-deterministic, so these figures reproduce exactly on any MI355X.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

STEPS, PLATEAU_FROM, I = 6400, 2560, 3483
_CACHE = {}


def _pair(seed):
    """(bf16 run, split run) with training-noise seed `seed`; computed once per process."""
    if seed not in _CACHE:
        import convergence as C
        import svae_cf_amd as S
        if "readers" not in _CACHE:
            hp = C.hyper_params(I)
            _CACHE["readers"] = (hp,) + C.make_readers(5120, 1536, I, hp)
        hp, tr, va = _CACHE["readers"]
        _CACHE[seed] = tuple(C.run_mode(prec, tr, va, hp, STEPS, 64, noise_seed=seed)
                             for prec in (S.PREC_BF16, S.PREC_SPLIT))
    return _CACHE[seed]


def _plateau_diff(seed):
    import convergence as C
    a, b = (C.plateau(r, PLATEAU_FROM) for r in _pair(seed))
    return {k: a[k]["mean"] - b[k]["mean"] for k in a}, a, b


def test_bf16_training_tracks_the_fp32_class_mode_early_and_reaches_its_plateau():
    bf, sp = _pair(5)
    la, lb = np.asarray(bf["step_loss"][:320]), np.asarray(sp["step_loss"][:320])
    assert np.all(np.isfinite(bf["step_loss"])) and np.all(np.isfinite(sp["step_loss"]))
    assert float(np.max(np.abs(la - lb) / np.abs(lb))) <= 2e-2
    for ea, eb in zip(bf["epochs"][:4], sp["epochs"][:4]):
        assert ea["step"] == eb["step"] <= 320
        assert abs(ea["loss"] - eb["loss"]) <= 3e-3 * eb["loss"], (ea, eb)
        assert abs(ea["Rec@100"] - eb["Rec@100"]) <= 0.5, (ea, eb)
    e4a, e4b = bf["epochs"][3], sp["epochs"][3]
    assert abs(e4a["NDCG@100"] - e4b["NDCG@100"]) <= 0.5, (e4a, e4b)
    # the model has learnt something by the plateau: Rec@100 27 (popularity level of the first epochs) -> 55
    d, a, b = _plateau_diff(5)
    assert a["Rec@100"]["mean"] > 50 and b["Rec@100"]["mean"] > 50, (a, b)
    assert a["NDCG@100"]["n"] == 49
    assert abs(d["NDCG@100"]) <= 1.0 and abs(d["Rec@100"]) <= 1.0, d
    assert abs(d["loss"]) <= 5e-3 * b["loss"]["mean"], d


def test_bf16_plateau_within_half_a_point_on_the_average_of_two_noise_seeds():
    d5, _, _ = _plateau_diff(5)
    d6, a6, b6 = _plateau_diff(6)
    assert abs(d6["NDCG@100"]) <= 1.0 and abs(d6["Rec@100"]) <= 1.0, d6
    assert abs(d6["loss"]) <= 5e-3 * b6["loss"]["mean"], d6
    for k in ("NDCG@100", "Rec@100", "NDCG@10", "Rec@10"):
        assert abs(0.5 * (d5[k] + d6[k])) <= 0.5, (k, d5[k], d6[k])        # north_star: 0.5 absolute
