import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import han_oracle as ho, han_oracle_torch as ht
from tests import rng_ref
from tests.helpers import load_params, make_problem, gpu_inputs, rel_err
from tests.test_gpu_parity import _oracle_grads, _t
from han_amd import rng as hrng, layers, ops
from han_amd.gat import HeteGAT_multi
dev = torch.device("cuda:0")
P = 8; drop = 0.6
for tdt in (torch.float32, torch.bfloat16):
    prob = make_problem(91, 200, 32, P, 3, [0.03, 0.3, 0.1, 0.01])
    xb = torch.tensor(prob["x"][0], dtype=torch.float32).to(torch.bfloat16)
    prob["x"] = xb.to(torch.float32).numpy().astype(np.float64)[None]
    bp = ht.to_batched(prob["params"])
    model = HeteGAT_multi().build(P, 32, 3, device=dev, table_dtype=tdt)
    load_params(model, bp)
    hrng.manual_seed(31)
    seeds = [hrng.next_seed() for _ in range(P)]
    hrng.manual_seed(31)
    keep = rng_ref.keep_prob32(drop)
    masks = []
    for q in range(P):
        rp, ci = ho.bias_to_csr(prob["biases"][q])
        masks.append({"seq": torch.tensor(rng_ref.seq_mask(seeds[q], 200, 32, 8, drop)),
                      "coef": torch.tensor(rng_ref.coef_mask_csr(seeds[q], rp, ci, 8, drop)),
                      "fts": torch.tensor(rng_ref.fts_mask(seeds[q], 200, 64, drop))})
    loss_ref, gref, lg_ref = _oracle_grads(prob, bp, masks=masks, keep=keep, dense=False)
    _, graphs = gpu_inputs(prob, dev)
    labels = _t(prob["labels"], dev, torch.int32)
    mask = _t(prob["mask"].astype(np.uint8), dev, torch.uint8)
    model.zero_grad_flat()
    xg = xb.to(dev) if tdt == torch.bfloat16 else xb.float().to(dev)
    M = model.node_level([xg] * P, graphs, drop, drop, True, ops.ACT_ELU)
    Z, _ = layers.SemanticAttention.apply(M, model.w_omega, model.b_omega, model.u_omega)
    loss, acc, logits = layers.ClassifierLoss.apply(Z, model.Wc, model.bc, labels, mask, 1.0 / int(prob["mask"].sum()))
    loss.backward()
    print(tdt, "logits", rel_err(logits.cpu().numpy(), lg_ref), "loss", float(loss), loss_ref)
    for k in ht.PARAM_ORDER:
        g = getattr(model, k).grad.cpu().numpy(); r = gref[k]
        per = [float(np.abs(g[q] - r[q]).max() / (np.abs(r).max())) for q in range(g.shape[0])] if g.shape[0] == P else None
        print("  ", k, "rel", rel_err(g, r), "max|ref|", float(np.abs(r).max()), "per-p", np.round(per, 4) if per else "")
