"""Downstream evaluation of ``final_embed`` (the step after the hot path):
``jhyexp.py:20-86`` restated -- KNN (k=5) macro/micro-F1 over train fractions
0.2/0.4/0.6/0.8 x 10 shuffles, and KMeans NMI/ARI x 10.  CPU, scikit-learn, as in
the reference (``ex_acm3025.py:279-291``); it returns the scores instead of only
printing them.

Random streams: the reference draws its shuffles from NumPy's GLOBAL legacy generator
(``np.random.permutation``, jhyexp.py:33) and lets scikit-learn's KMeans fall back on the
same global generator (``random_state=None``, jhyexp.py:62).  Here both use ONE
``np.random.RandomState(seed)`` -- the same bit stream as ``np.random.seed(seed)`` followed
by the reference's calls -- so a seeded run reproduces the reference's numbers exactly
(tests/golden/jhyexp_ref.npz holds outputs of the reference's own functions)."""
from __future__ import annotations

import numpy as np


def my_KNN(x, y, k=5, split_list=(0.2, 0.4, 0.6, 0.8), time=10, shuffle=True, seed=None, verbose=True):
    """jhyexp.py:20-51.  x (n,d) embeddings, y (n,) labels or one-hot (n,c).
    Returns {split: (macro_f1, micro_f1)} averaged over `time` repetitions."""
    from sklearn.metrics import f1_score
    from sklearn.neighbors import KNeighborsClassifier
    rng = np.random.RandomState(seed) if not isinstance(seed, np.random.RandomState) else seed
    x = np.squeeze(np.array(x))
    y = np.array(y)
    if y.ndim > 1:
        y = np.argmax(y, axis=1)
    out = {}
    for ss in split_list:
        split = int(x.shape[0] * ss)
        macro, micro = [], []
        for _ in range(time):
            if shuffle:                      # the reference re-permutes the SAME arrays each time (:32-35)
                perm = rng.permutation(x.shape[0])
                x, y = x[perm, :], y[perm]
            est = KNeighborsClassifier(n_neighbors=k).fit(x[:split], y[:split])
            pred = est.predict(x[split:])
            macro.append(f1_score(y[split:], pred, average="macro"))
            micro.append(f1_score(y[split:], pred, average="micro"))
        out[ss] = (float(np.mean(macro)), float(np.mean(micro)))
        if verbose:
            print("KNN({}avg, split:{}, k={}) f1_macro: {:.4f}, f1_micro: {:.4f}".format(
                time, ss, k, out[ss][0], out[ss][1]))
    return out


def my_Kmeans(x, y, k=4, time=10, seed=None, verbose=True):
    """jhyexp.py:54-86.  Returns (NMI, ARI) averaged over `time` fits."""
    from sklearn.cluster import KMeans
    from sklearn.metrics import adjusted_rand_score, normalized_mutual_info_score
    x = np.squeeze(np.array(x))
    y = np.array(y)
    if y.ndim > 1:
        y = np.argmax(y, axis=1)
    nmi, ari = [], []
    rs = np.random.RandomState(seed) if not isinstance(seed, np.random.RandomState) else seed
    est = KMeans(n_clusters=k, random_state=rs)          # ONE estimator, library defaults (jhyexp.py:62)
    for i in range(time):
        pred = est.fit(x, y).predict(x)                  # re-fitted `time` times (jhyexp.py:67-68)
        nmi.append(normalized_mutual_info_score(y, pred))
        ari.append(adjusted_rand_score(y, pred))
    res = float(np.mean(nmi)), float(np.mean(ari))
    if verbose:
        print("NMI (10 avg): {:.4f} , ARI (10avg): {:.4f}".format(*res))
    return res
