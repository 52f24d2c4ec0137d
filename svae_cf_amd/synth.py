"""Synthetic user sequences shaped like the reference's datasets (no real data offline).

Lengths: clip(round(lognormal(mu, sigma)), 5, Lmax) items per user.  Items: drawn from a Zipf(1.0)
popularity over the catalogue mixed with a per-user latent cluster (64 clusters, each boosting a random
2% of the items), duplicates removed keeping first occurrence = chronological order.
"""
import numpy as np

SHAPES = {
    # name: (users, items, mu, sigma, Lmax)
    "ml-1m": (6040, 3483, 4.0, 1.0, 1000),
    "ml-20m": (138000, 20000, 4.2, 0.9, 200),
    "netflix": (480000, 17770, 4.6, 1.1, 1000),
    "large": (1000000, 1000000, 3.5, 0.8, 200),
}


def make_users(n_users, total_items, mu, sigma, lmax, seed=98765, n_clusters=64):
    rng = np.random.default_rng(seed)
    lens = np.clip(np.rint(rng.lognormal(mu, sigma, n_users)), 5, min(lmax, total_items // 2)).astype(np.int64)
    pop = 1.0 / np.arange(1, total_items + 1, dtype=np.float64)
    cdf = np.cumsum(pop / pop.sum())
    item_of_rank = rng.permutation(total_items)
    csize = max(8, total_items // 50)
    clusters = [rng.choice(total_items, csize, replace=False) for _ in range(n_clusters)]
    users = []
    for n in lens:
        c = clusters[int(rng.integers(n_clusters))]
        out = np.empty(0, dtype=np.int64)
        while len(out) < n:
            m = int(2 * n + 16)
            z = item_of_rank[np.minimum(np.searchsorted(cdf, rng.random(m)), total_items - 1)]
            cl = c[rng.integers(csize, size=m)]
            draw = np.where(rng.random(m) < 0.5, cl, z)
            out = np.concatenate((out, draw))
            _, first = np.unique(out, return_index=True)
            out = out[np.sort(first)]
        users.append(out[:n].astype(np.int32))
    return users


def bucketed_batches(lengths, batch_size, rng, window=16, world=1):
    """Length-bucketed batches: shuffle users, sort inside windows of window*batch_size*world users,
    cut into global batches of batch_size*world users, shuffle the global batches.

    world == 1: returns a list of index arrays (one per step).
    world > 1 : returns a list of steps, each a list of ``world`` index arrays (rank r's batch of that
    step).  A global batch holds users of similar length in descending order and is dealt round-robin
    (rank r takes positions r, r+world, ...), so at every step all ranks have the same Tmax and the same
    number of rows to within (longest - shortest user of the global batch): the gradient all-reduce of a
    step never waits for a rank that drew a longer batch.  A tail that does not fill every rank with the
    same number of users is dropped (collectives stay matched)."""
    n = len(lengths)
    lengths = np.asarray(lengths)
    perm = rng.permutation(n)
    out = []
    gb = batch_size * world
    w = window * gb
    for s in range(0, n, w):
        idx = perm[s:s + w]
        idx = idx[np.argsort(-lengths[idx], kind="stable")]
        for b in range(0, len(idx), gb):
            out.append(idx[b:b + gb])
    order = rng.permutation(len(out))
    out = [out[i] for i in order]
    if world == 1:
        return out
    steps = []
    for g in out:
        m = len(g) // world * world
        if m == 0:
            continue
        steps.append([g[r:m:world] for r in range(world)])
    return steps
