"""ratings.dat -> pro_sg/*.csv, the notebook's preprocessing (nb:129-283) without pandas.

Host-side data preparation either side of the hot path: numpy only, one pass of sorts and
bincounts instead of per-user DataFrame groups (the reference takes minutes on ML-1M, mostly in groupby.apply and
in 1.5k small concats; this takes about a second).  Writes the same six files, byte for byte, that
``load_data`` (nb:406-437, notebook_api.load_data) reads:

    train.csv validation_tr.csv validation_te.csv test_tr.csv test_te.csv   'uid,sid,rating' + one line per event
    unique_sid.txt                                                          raw movie ids, train order

Steps, with the reference lines they follow:
  1. read ``userId::movieId::rating::timestamp``                                     nb:132-134
  2. keep ratings > 3.5                                                               nb:140
  3. drop users with more than ``max_seq_len`` kept events                            nb:143
  4. order events by user id, then timestamp                                          nb:146
  5. keep users with >= ``min_uc`` events (filter_triplets, min_sc = 0)               nb:155-170, nb:215
  6. np.random.seed(98765); permute the sorted user ids; the last 2 x n_heldout are validation / test  nb:224-235
  7. item ids = order of first appearance in the TRAIN users' events                  nb:237-241
  8. validation / test events restricted to train items, each user's last 20 % held out
     (users left with < 5 events keep everything in the fold-in part)                 nb:250-258, nb:172-212
  9. uid = position in the permuted user list, sid = item id                          nb:214-220

Not reproduced: the order of events with EQUAL timestamps inside one user (the reference sorts with an unstable
quicksort, nb:146; this module is stable: file order decides) -- see oracle/make_golden_prep.py.
"""
import os

import numpy as np

__all__ = ["preprocess", "read_ratings"]


def read_ratings(path, sep="::"):
    """(user, movie, rating, timestamp) int64 columns of a MovieLens ``ratings.dat``."""
    with open(path, "rb") as f:
        raw = f.read()
    if not raw.strip():
        raise ValueError("%s holds no ratings" % path)
    flat = np.array(raw.replace(sep.encode(), b" ").split(), dtype=np.int64)
    if flat.size % 4:
        raise ValueError("%s: expected 4 '%s'-separated integer fields per line" % (path, sep))
    t = flat.reshape(-1, 4)
    return t[:, 0].copy(), t[:, 1].copy(), t[:, 2].copy(), t[:, 3].copy()


def _split_tail(users, test_prop):
    """Boolean mask of the held-out events: per user (events already grouped and in time order), the events from
    position int((1.0 - test_prop) * n) on when n >= 5, none otherwise (split_train_test_proportion nb:172-212)."""
    held = np.zeros(len(users), dtype=bool)
    if len(users) == 0:
        return held
    starts = np.flatnonzero(np.r_[True, users[1:] != users[:-1]])
    ends = np.r_[starts[1:], len(users)]
    for s, e in zip(starts, ends):
        n = e - s
        if n >= 5:
            held[s + int((1.0 - test_prop) * n):e] = True
    return held


def _write_csv(path, uid, sid, rating):
    with open(path, "w") as f:
        f.write("uid,sid,rating\n")
        f.write("".join("%d,%d,%d\n" % r for r in zip(uid.tolist(), sid.tolist(), rating.tolist())))


def preprocess(data_dir, out_dir=None, max_seq_len=1000, n_heldout_users=750, min_uc=5, test_prop=0.2,
               seed=98765, ratings_file="ratings.dat"):
    """Writes ``out_dir`` (default ``data_dir/pro_sg``, as nb:125-126) and returns a summary dict.  Like the notebook
    (nb:128) it does nothing when the directory already exists."""
    out_dir = os.path.join(data_dir, "pro_sg") if out_dir is None else out_dir
    if os.path.isdir(out_dir):
        return dict(skipped=True, out_dir=out_dir)
    user, movie, rating, stamp = read_ratings(os.path.join(data_dir, ratings_file))
    keep = rating > 3.5
    user, movie, rating, stamp = user[keep], movie[keep], rating[keep], stamp[keep]
    ids, inv, cnt = np.unique(user, return_inverse=True, return_counts=True)
    keep = (cnt <= max_seq_len)[inv]
    user, movie, rating, stamp = user[keep], movie[keep], rating[keep], stamp[keep]
    order = np.lexsort((stamp, user))                         # stable: by user, then time, then file order
    user, movie, rating = user[order], movie[order], rating[order]
    ids, inv, cnt = np.unique(user, return_inverse=True, return_counts=True)
    if min_uc > 0:
        keep = (cnt >= min_uc)[inv]
        user, movie, rating = user[keep], movie[keep], rating[keep]
    unique_uid = np.unique(user)                              # usercount.index: ascending ids
    n_items_all = len(np.unique(movie))
    n_users = unique_uid.size
    if n_users <= 2 * n_heldout_users:
        raise ValueError("%d users after filtering, but %d are to be held out" % (n_users, 2 * n_heldout_users))
    np.random.seed(seed)                                      # the reference's own stream and seed (nb:224-225)
    unique_uid = unique_uid[np.random.permutation(n_users)]
    profile2id = np.full(int(unique_uid.max()) + 1, -1, dtype=np.int64)
    profile2id[unique_uid] = np.arange(n_users)
    pos = profile2id[user]                                    # position of each event's user in the permuted list
    n_tr = n_users - 2 * n_heldout_users
    is_tr, is_vd = pos < n_tr, (pos >= n_tr) & (pos < n_tr + n_heldout_users)
    is_te = pos >= n_tr + n_heldout_users
    # item ids: first appearance among the train users' events (pd.unique keeps that order)
    tr_movies = movie[is_tr]
    _, first = np.unique(tr_movies, return_index=True)
    unique_sid = tr_movies[np.sort(first)]
    show2id = np.full(int(movie.max()) + 1, -1, dtype=np.int64)
    show2id[unique_sid] = np.arange(unique_sid.size)
    sid = show2id[movie]
    os.makedirs(out_dir)
    with open(os.path.join(out_dir, "unique_sid.txt"), "w") as f:
        f.write("".join("%s\n" % s for s in unique_sid.tolist()))
    _write_csv(os.path.join(out_dir, "train.csv"), pos[is_tr], sid[is_tr], rating[is_tr])
    summary = dict(skipped=False, out_dir=out_dir, n_events=int(len(user)), n_users=int(n_users),
                   n_items=int(n_items_all), n_train_items=int(unique_sid.size), n_train_users=int(n_tr))
    for name, part in (("validation", is_vd), ("test", is_te)):
        m = part & (sid >= 0)                                 # only items the train users have (nb:251, nb:256)
        u, p, s, r = user[m], pos[m], sid[m], rating[m]
        held = _split_tail(u, test_prop)
        _write_csv(os.path.join(out_dir, name + "_tr.csv"), p[~held], s[~held], r[~held])
        _write_csv(os.path.join(out_dir, name + "_te.csv"), p[held], s[held], r[held])
        summary["n_%s_events" % name] = (int((~held).sum()), int(held.sum()))
    return summary
