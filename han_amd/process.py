"""``utils/process.py`` surface used by the HAN script: :func:`adj_to_bias`, plus
the direct adjacency -> CSR route that replaces the dense mask altogether and
the ``.mat`` loader of ``ex_acm3025.py:57-87``.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .graph import CSRGraph


def adj_to_bias(adj, sizes, nhood=1):
    """utils/process.py:14-25, vectorised (the reference's O(N^2) Python double
    loop at :21-24 takes ~9 s per meta-path at N = 3025).

    adj (G,N,N) numpy array.  mt = (adj + I)^nhood; entries > 0 inside
    [0:sizes[g]]^2 become 1; returns -1e9 * (1 - mt), float64, same as the
    reference (entries outside the sizes[g] square keep their raw value)."""
    adj = np.asarray(adj)
    nb_graphs, n = adj.shape[0], adj.shape[1]
    mt = np.empty(adj.shape)
    eye = np.eye(n)
    for g in range(nb_graphs):
        m = eye.copy()
        for _ in range(nhood):
            m = np.matmul(m, adj[g] + eye)
        s = sizes[g]
        blk = m[:s, :s]
        blk[blk > 0.0] = 1.0
        mt[g] = m
    return -1e9 * (1.0 - mt)


def adj_to_graph(adj, nhood=1, device=None) -> CSRGraph:
    """Adjacency (N,N) dense / scipy sparse -> CSRGraph of ((adj + I)^nhood > 0),
    i.e. the edge set adj_to_bias encodes, without the N x N mask.  Valid for
    non-negative adjacency entries (as every meta-path count matrix is)."""
    a = sp.csr_matrix(adj[0] if getattr(adj, "ndim", 2) == 3 else adj)
    if a.nnz and a.data.min() < 0:
        raise ValueError("adjacency entries must be non-negative")
    n = a.shape[0]
    a = ((a != 0).astype(np.int64) + sp.identity(n, dtype=np.int64, format="csr"))
    a = (a != 0).astype(np.int64)
    m = a.copy()
    for _ in range(nhood - 1):
        m = ((m @ a) != 0).astype(np.int64)
    m = sp.csr_matrix(m)
    m.sort_indices()
    return CSRGraph.from_arrays(m.indptr, m.indices, n, device=device)


def sample_mask(idx, l):
    """ex_acm3025.py:50-54."""
    mask = np.zeros(l)
    mask[np.asarray(idx).ravel()] = 1
    return mask.astype(bool)


def load_data_mat(path, metapaths=("PAP", "PLP")):
    """ex_acm3025.py:57-87 (`load_data_dblp`): the ACM3025.mat layout
    (keys label, feature, PAP, PLP, train_idx, val_idx, test_idx).  Returns
    (rownetworks, truefeatures_list, y_train, y_val, y_test, train_mask,
    val_mask, test_mask) exactly as the reference, including the ``- I`` on the
    meta-path matrices (:61) that adj_to_bias later re-adds."""
    import scipy.io as sio
    data = sio.loadmat(path)
    truelabels, truefeatures = data["label"], data["feature"].astype(float)
    n = truefeatures.shape[0]
    rownetworks = [np.asarray(data[k]) - np.eye(n) for k in metapaths]
    y = truelabels
    train_mask = sample_mask(data["train_idx"], y.shape[0])
    val_mask = sample_mask(data["val_idx"], y.shape[0])
    test_mask = sample_mask(data["test_idx"], y.shape[0])
    y_train, y_val, y_test = np.zeros(y.shape), np.zeros(y.shape), np.zeros(y.shape)
    y_train[train_mask, :] = y[train_mask, :]
    y_val[val_mask, :] = y[val_mask, :]
    y_test[test_mask, :] = y[test_mask, :]
    truefeatures_list = [truefeatures] * (len(metapaths) + 1)      # :86 (3 copies for 2 meta-paths)
    return rownetworks, truefeatures_list, y_train, y_val, y_test, train_mask, val_mask, test_mask
