"""Generates the committed golden fixtures.  Run from the repo root:

    python tests/golden/gen_fixtures.py

1. adj_to_bias_*.npz -- produced by EXECUTING the reference's own
   ``adj_to_bias`` (``/root/reference/utils/process.py:14-25``: NumPy only; the
   module itself cannot be imported because of a removed SciPy import at :5, so
   the function's source lines are exec'd from the file where it lies).  This
   is the one piece of the reference that runs in this container; its outputs
   pin ``oracle.han_oracle.adj_to_bias`` and ``han_amd.process.adj_to_bias``.
   The fixture holds inputs and outputs only, no reference source.
3. jhyexp_ref.npz -- outputs of the reference's own ``jhyexp.my_KNN`` / ``my_Kmeans``
   (``/root/reference/jhyexp.py:20-86``, imported from where it lies; scikit-learn only)
   on seeded synthetic embeddings with ``np.random.seed(seed)``: pins
   ``han_amd.evaluate``.  my_KNN only prints, so its stdout lines are parsed.
2. han_forward_n64.npz -- inputs, parameters and float64 outputs of the oracle
   restatement of HeteGAT_multi.inference (PARITY UNPINNED vs TensorFlow: TF1 is
   not installable here; see oracle/han_oracle.py).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import han_oracle as ho            # noqa: E402
from oracle import han_oracle_torch as ht      # noqa: E402

REF = "/root/reference/utils/process.py"


def reference_adj_to_bias():
    lines = open(REF).read().split("\n")
    src = "\n".join(lines[13:25])                 # def adj_to_bias ... return (lines 14-25)
    ns = {"np": np}
    exec(compile(src, REF, "exec"), ns)
    return ns["adj_to_bias"]


def gen_adj_to_bias():
    fn = reference_adj_to_bias()
    rng = np.random.default_rng(0)
    cases = {}
    a3 = np.array([[[0., 1., 0.], [1., 0., 0.], [0., 0., 0.]]])
    cases["k3"] = (a3, [3], 1)
    a = (rng.random((1, 12, 12)) < 0.15).astype(float)
    cases["n12_h1"] = (a, [12], 1)
    cases["n12_h2"] = (a, [12], 2)
    a2 = (rng.random((2, 9, 9)) < 0.2).astype(float)
    cases["g2_n9_sizes"] = (a2, [9, 6], 1)        # second graph: only a 6x6 block binarised
    w = rng.integers(0, 4, size=(1, 10, 10)).astype(float)   # meta-path COUNT matrix (non-binary)
    cases["n10_counts"] = (w - np.eye(10) * np.diag(w[0]), [10], 1)
    out = {}
    for k, (adj, sizes, nh) in cases.items():
        out[k + "_adj"] = adj
        out[k + "_sizes"] = np.array(sizes)
        out[k + "_nhood"] = np.array(nh)
        out[k + "_bias"] = fn(adj.copy(), sizes, nhood=nh)
    np.savez_compressed(os.path.join(HERE, "adj_to_bias_ref.npz"), **out)
    print("adj_to_bias_ref.npz:", sorted(cases))


def gen_forward():
    rng = np.random.default_rng(2024)
    N, F, P, C = 64, 24, 2, 3
    x = rng.standard_normal((1, N, F))
    params = ho.init_params(rng, P, F, C, nonzero_biases=True)
    out = {"N": N, "F": F, "P": P, "C": C, "x": x[0]}
    biases = []
    for p, dens in enumerate((0.05, 0.4)):
        a = (rng.random((N, N)) < dens).astype(float)
        a = np.maximum(a, a.T)
        np.fill_diagonal(a, 0)
        a[3, :] = a[:, 3] = 0                       # self-loop-only row
        b = ho.adj_to_bias(a[None], [N], 1)
        biases.append(b)
        rp, ci = ho.bias_to_csr(b)
        out[f"rowptr_{p}"], out[f"colidx_{p}"] = rp, ci
    lg, fe, att = ho.hetegat_multi_inference([x] * P, C, N, False, 0.0, 0.0, biases, [8], [8, 1], params)
    bp = ht.to_batched(params)
    for k in ht.PARAM_ORDER:
        out["param_" + k] = bp[k].numpy()
    out["logits"], out["final_embed"], out["att_val"] = lg[0], fe, att
    np.savez_compressed(os.path.join(HERE, "han_forward_n64.npz"), **out)
    print("han_forward_n64.npz: logits", lg.shape)


def gen_jhyexp():
    import contextlib
    import importlib.util
    import io
    import re
    spec = importlib.util.spec_from_file_location("jhyexp_ref", "/root/reference/jhyexp.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(3)
    n, c = 240, 4
    y = rng.integers(0, c, n)
    centers = rng.standard_normal((c, 16)) * 0.55
    x = centers[y] + rng.standard_normal((n, 16))          # overlapping clusters: scores well below 1
    out = {"x": x, "y": y, "seed": np.array(11), "time": np.array(3), "k_knn": np.array(5),
           "k_means": np.array(c)}
    np.random.seed(11)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref.my_KNN(x, np.eye(c)[y], k=5, time=3)
    rows = re.findall(r"split:([0-9.]+), k=5\) f1_macro: ([0-9.]+), f1_micro: ([0-9.]+)", buf.getvalue())
    assert len(rows) == 4, buf.getvalue()
    out["knn"] = np.array([[float(v) for v in r] for r in rows])        # (4,3): split, macro, micro (4 decimals)
    np.random.seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        nmi, ari = ref.my_Kmeans(x, y, k=c, time=3, return_NMI=True)
    out["kmeans"] = np.array([nmi, ari])
    import sklearn
    out["sklearn_version"] = np.array(sklearn.__version__)
    np.savez_compressed(os.path.join(HERE, "jhyexp_ref.npz"), **out)
    print("jhyexp_ref.npz:", out["knn"].tolist(), out["kmeans"].tolist())


if __name__ == "__main__":
    which = sys.argv[1:] or ["adj", "forward", "jhyexp"]
    if "adj" in which:
        gen_adj_to_bias()
    if "forward" in which:
        gen_forward()
    if "jhyexp" in which:
        gen_jhyexp()
