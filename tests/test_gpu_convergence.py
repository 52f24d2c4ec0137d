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
evaluations); two runs of ONE mode that differ only in the seed of the latent noise sit up to 1.0 NDCG@100 / 1.9 Rec@100
apart in their plateau MEANS; and two fp32-class implementations that agree to 1e-6 per step (PREC_SPLIT with the GRU
through the split template, and with the generic fp32 GRU kernels) end 0.04 / 0.02 apart, their validation loss 1.1e-3.
Every change of a summation order anywhere in the step (a work division, a time-range cut) moves the per-seed offset
between the two modes by a few tenths: over four code states of round 3 the offset at seed 5 was +0.50, +0.32, +0.59
NDCG@100 and +0.61, +0.46, +0.99 Rec@100.  A gate of 0.5 on single evaluations, or on one seed's plateau, would fail
between two fp32 runs.  So:
  * early phase (the first 320 steps, where the two trajectories still coincide): per-step training loss within
    2e-2 relative (measured 7e-3), validation loss within 3e-3 relative (1.4e-3), NDCG@100 and Rec@100 within 0.5
    absolute (0.08 / 0.015);
  * plateau (steps 2 560 .. 6 400, evaluated every epoch), per noise seed: the plateau means of bf16 and split within
    2.0 absolute (what two noise seeds of one mode differ by) and the validation loss within 1.5e-2 relative;
  * plateau, MEAN OVER FOUR NOISE SEEDS (5, 6, 7, 8): within 0.5 absolute (north_star) for NDCG@10/100 and Rec@10/100
    -- final code: +0.03 NDCG@100, +0.13 Rec@100; per seed +0.59 / -0.15 / -0.14 / -0.20 and +0.99 / +0.10 / -0.58 / -0.01
    -- and the validation loss within 5e-3 relative on that mean (per seed 6.6e-3, 4e-4, 1.5e-3, 2e-4).
The kernels are bit-deterministic, so these figures reproduce exactly on any MI355X for a given state of the code.
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
    assert abs(d["NDCG@100"]) <= 2.0 and abs(d["Rec@100"]) <= 2.0, d
    assert abs(d["loss"]) <= 1.5e-2 * b["loss"]["mean"], d


def test_bf16_plateau_within_half_a_point_on_the_mean_of_four_noise_seeds():
    ds, losses = [], []
    for seed in (5, 6, 7, 8):
        d, a, b = _plateau_diff(seed)
        assert abs(d["NDCG@100"]) <= 2.0 and abs(d["Rec@100"]) <= 2.0, (seed, d)
        assert abs(d["loss"]) <= 1.5e-2 * b["loss"]["mean"], (seed, d)
        ds.append(d)
        losses.append(b["loss"]["mean"])
    for k in ("NDCG@100", "Rec@100", "NDCG@10", "Rec@10"):
        m = float(np.mean([d[k] for d in ds]))
        assert abs(m) <= 0.5, (k, m, [round(d[k], 3) for d in ds])        # north_star: 0.5 absolute
    assert abs(float(np.mean([d["loss"] for d in ds]))) <= 5e-3 * float(np.mean(losses))
