import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from han_amd import synth, rng as hrng
from han_amd.gat import HeteGAT_multi
from han_amd.trainer import HANTrainer
dev = torch.device("cuda:0")
for seed in (1, 2, 3):
    hrng.manual_seed(seed)
    wl = synth.make_workload("tiny", device="cpu")
    model = HeteGAT_multi().build(wl["p"], wl["f"], wl["c"], device=dev)
    x = wl["x"].to(dev)
    labels = x[:, :3].argmax(1).to(torch.int32)
    graphs = [g.to(dev) for g in wl["graphs"]]
    tr = HANTrainer(model, [x] * wl["p"], graphs, labels, wl["train_mask"] | 1, wl["val_mask"])
    out = []
    for ep in range(60):
        tl, ta, vl, va = tr.epoch()
        if ep % 10 == 0 or ep == 59:
            out.append((round(float(tl), 4), round(float(vl), 4), round(float(va), 3)))
    print(seed, out, "val rows", int(wl["val_mask"].sum()))
