#!/usr/bin/env python3
"""The reference's experiment (``ex_acm3025.py``) on han_amd: load the ACM3025-style
``.mat`` (or, with no file, a synthetic graph of the same shape), train the 8-head HAN
full-graph with early stopping, restore the best weights, report the test metrics and the
KNN / KMeans scores of ``final_embed``.

    python examples/ex_acm3025.py [--mat ACM3025.mat] [--epochs 200] [--graph]

Hyper-parameters are the reference's (ex_acm3025.py:16-31): lr 0.005, l2 0.001,
hid_units [8], n_heads [8, 1], dropout 0.6/0.6, patience 100, mp_att_size 128.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from han_amd import evaluate, process, synth  # noqa: E402
from han_amd.gat import HeteGAT_multi  # noqa: E402
from han_amd.trainer import HANTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mat", default=None, help="ACM3025.mat-style file (keys label, feature, PAP, PLP, *_idx)")
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--patience", type=int, default=100)
    ap.add_argument("--graph", action="store_true", help="replay the epoch from a hipGraph")
    ap.add_argument("--overlap-eval", action="store_true",
                    help="with --graph: validate the parameters of epoch k - 1 beside the training step of epoch k "
                         "(HANTrainer(overlap_eval=True)); the early-stopping rule then sees each pair one call later")
    ap.add_argument("--planted", action="store_true",
                    help="without --mat: a synthetic task WITH structure (communities) instead of the ACM-shaped "
                         "random-label workload, to watch the model learn")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(args.seed)

    if args.mat:
        adj_list, fea_list, y_train, y_val, y_test, train_mask, val_mask, test_mask = process.load_data_mat(args.mat)
        graphs = [process.adj_to_graph(a, nhood=1, device=dev) for a in adj_list]      # == adj_to_bias's edge set
        xs = [torch.tensor(f, dtype=torch.float32, device=dev) for f in fea_list[:len(graphs)]]
        y = y_train + y_val + y_test
        labels = torch.tensor(y.argmax(1), dtype=torch.int32, device=dev)
        masks = [torch.tensor(m, device=dev) for m in (train_mask, val_mask, test_mask)]
        nb_classes = y.shape[1]
    elif args.planted:
        wl = synth.planted_partition(3025, 3, 2, 64, deg_in=8, deg_out=2, noise=1.0, seed=args.seed, device=dev)
        graphs, xs = wl["graphs"], [wl["x"]] * wl["p"]
        labels, nb_classes = wl["labels"], wl["c"]
        masks = [wl["train_mask"], wl["val_mask"], wl["test_mask"]]
    else:
        wl = synth.make_workload("acm-like", device=dev)
        graphs, xs = wl["graphs"], [wl["x"]] * wl["p"]
        labels, nb_classes = wl["labels"], wl["c"]
        test = ~(wl["train_mask"].bool() | wl["val_mask"].bool())
        masks = [wl["train_mask"], wl["val_mask"], test]
    n, ft = xs[0].shape
    print(f"nodes {n}, features {ft}, classes {nb_classes}, meta-paths {len(graphs)}, "
          f"edges {[g.nnz for g in graphs]}")

    model = HeteGAT_multi().build(len(graphs), ft, nb_classes, (8,), (8, 1), 128, device=dev)
    tr = HANTrainer(model, xs, graphs, labels, masks[0], masks[1], lr=0.005, l2_coef=0.001,
                    attn_drop=0.6, ffd_drop=0.6, patience=args.patience, use_graph=args.graph,
                    overlap_eval=args.graph and args.overlap_eval)
    t0 = time.perf_counter()
    for epoch in range(args.epochs):
        tl, ta, vl, va = (float(v) for v in tr.epoch())
        if tr.overlap_eval and epoch == 0:
            continue                         # that pair belongs to the initial parameters (ex_acm3025.py has no such line)
        if epoch % 10 == 0:
            print(f"epoch {epoch:4d}  train loss {tl:.5f} acc {ta:.5f} | val loss {vl:.5f} acc {va:.5f}")
        if tr.early_stopping(vl, va):
            print(f"early stop at epoch {epoch}: min val loss {tr.vlss_mn:.5f}, max val acc {tr.vacc_mx:.5f}")
            break
    else:
        if tr.overlap_eval:                  # the validation pair of the last epoch
            tr.early_stopping(*(float(v) for v in tr.flush_eval()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{epoch + 1} epochs in {dt:.2f} s ({(epoch + 1) / dt:.1f} epochs/s)")
    tr.restore_best()
    w_test = 1.0 / max(int(masks[2].sum()), 1)
    tl, ta = tr.eval_step(masks[2].to(torch.uint8).contiguous(), w_test)
    print(f"test loss {float(tl):.5f}  test accuracy {float(ta):.5f}")
    with torch.no_grad():
        _, final_embed, att = model.inference(xs, nb_classes, n, False, 0.0, 0.0, graphs, [8], [8, 1])
    print("mean meta-path attention:", att.mean(0).tolist())
    sel = masks[2].bool().cpu().numpy()
    emb, lab = final_embed.cpu().numpy()[sel], labels.cpu().numpy()[sel]
    evaluate.my_KNN(emb, lab, seed=args.seed)                     # ex_acm3025.py:288
    evaluate.my_Kmeans(emb, lab, k=nb_classes, seed=args.seed)    # :289


if __name__ == "__main__":
    main()
