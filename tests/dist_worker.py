"""Worker for the 2-rank tests: runs `epochs` reference epochs of a node-partitioned
HANTrainer and dumps (flat params, per-epoch metrics) for rank 0 to compare with
the single-process run.  Backend gloo (host-staged collectives) so that it runs
both on CPU (with tests.cpu_backend patched in) and on ONE GPU shared by the ranks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_problem(dev, n=257, f=24, p=2, c=3, seed=5):
    from han_amd.graph import CSRGraph
    rng = np.random.default_rng(seed)
    x = torch.tensor(rng.standard_normal((n, f)), dtype=torch.float32, device=dev)
    graphs = []
    band = os.environ.get("HAN_TEST_GRAPH") == "band"
    for q in range(p):
        if band:      # locality: |i - j| <= 3 + q, directed, so the halo is a handful of boundary rows
            ii, jj = np.indices((n, n))
            a = (np.abs(ii - jj) <= 3 + q) & (rng.random((n, n)) < 0.8)
        else:
            a = rng.random((n, n)) < (0.02 if q == 0 else 0.15)     # directed: exercises the CSC build
        np.fill_diagonal(a, True)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(a.sum(1), out=rowptr[1:])
        g = CSRGraph.from_arrays(rowptr, np.nonzero(a)[1].astype(np.int32), n, device=dev)
        if os.environ.get("HAN_TEST_WEIGHTED") == "1":   # sp_attn_head values that scale the logits
            vals = torch.tensor(rng.uniform(-1.0, 2.0, size=g.nnz), dtype=torch.float32, device=dev)
            g = CSRGraph(g.rowptr, g.colidx, n, values=vals)
        graphs.append(g)
    labels = torch.tensor(rng.integers(0, c, n), dtype=torch.int32, device=dev)
    u = rng.random(n)
    tm = torch.tensor((u < 0.4).astype(np.uint8), device=dev)
    vm = torch.tensor((u >= 0.6).astype(np.uint8), device=dev)
    return x, graphs, labels, tm, vm, (n, f, p, c)


def run(rank, world, device, epochs, drop, out_path, port, use_cpu_backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if use_cpu_backend:
        from tests import cpu_backend
        cpu_backend.install()
    backend = os.environ.get("HAN_TEST_BACKEND", "gloo")
    forced = os.environ.get("HAN_FORCE_COLLECTIVES") == "1"
    if world > 1 or forced:
        if backend == "nccl":
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from han_amd import rng as hrng
    from han_amd.dist import NodePartition
    from han_amd.gat import HeteGAT_multi
    from han_amd.trainer import HANTrainer
    dev = torch.device(device)
    x, graphs, labels, tm, vm, (n, f, p, c) = build_problem(dev)
    hrng.manual_seed(77)
    gen = torch.Generator().manual_seed(3)
    bf16 = os.environ.get("HAN_TEST_BF16") == "1"            # configs[4] storage: X, H and g tables in bf16
    hid = int(os.environ.get("HAN_TEST_HID", "8"))       # e.g. 96: heads wider than 64 columns (layers.WideHeadAttention)
    heads = (8, 1) if hid == 8 else (2, 1)
    model = HeteGAT_multi().build(p, f, c, hid_units=(hid,), n_heads=heads, device=dev, generator=gen,
                                  table_dtype=torch.bfloat16 if bf16 else torch.float32)
    if bf16:
        x = x.to(torch.bfloat16)
    part = NodePartition(n, rank, world) if (world > 1 or forced) else None
    loc = (lambda t: part.local_rows(t).contiguous()) if part is not None else (lambda t: t)
    local = part is not None and os.environ.get("HAN_TEST_LOCAL") == "1"
    if local:      # the scalable form: a rank only ever sees its own destination rows (global column ids)
        from han_amd.dist import _row_block
        graphs = [_row_block(g, part.row_start, part.row_end, n) for g in graphs]
    # HAN_TEST_ALLGATHER=1: a negative halo threshold sends every meta-path down the all-gather path
    # HAN_TEST_REPLICATE=all|eval|auto: every rank holds the features of all rows and projects the whole
    # table itself in the named forward passes instead of exchanging it
    rep = os.environ.get("HAN_TEST_REPLICATE")
    tr = HANTrainer(model, [loc(x)] * p, graphs, loc(labels), loc(tm), loc(vm), attn_drop=drop,
                    ffd_drop=drop, part=part, graphs_local=local,
                    xs_full=[x] * p if (rep and part is not None) else None, replicate=rep or "auto",
                    max_halo_fraction=-1.0 if os.environ.get("HAN_TEST_ALLGATHER") == "1" else 0.6,
                    masked_backward=os.environ.get("HAN_TEST_MASKED_BWD") == "1")
    comm = None
    if part is not None and os.environ.get("HAN_TEST_COMM") == "1":      # what bench.py attaches for its N > 1 line
        from han_amd.dist import CommStats
        comm = part.comm = CommStats()
    hist = []
    for _ in range(epochs):
        hist.append(tr.reduce_metrics(*tr.epoch()))
    if rank == 0:
        pf, pb = model.halo_plans
        halo = 0 if pf is None else sum(x is not None for x in pf) + sum(x is not None for x in pb)
        halo_rows = 0 if pf is None else sum(x.n_halo for x in list(pf) + list(pb) if x is not None)
        extra = {}
        if comm is not None:
            extra = dict(comm_bytes=comm.bytes_received, comm_exchanges=comm.exchanges,
                         comm_allreduce=comm.allreduce_bytes, comm_wait_ms=comm.wait_ms(),
                         shard=part.shard, halo_rows=halo_rows, n_params=model.flat.numel())
        np.savez(out_path, flat=model.flat.detach().cpu().numpy(), hist=np.array(hist), halo_plans=halo,
                 replicate=",".join(sorted(tr.replicate)), **extra)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    # python tests/dist_worker.py rank world device epochs drop out port cpu_backend
    a = sys.argv[1:]
    run(int(a[0]), int(a[1]), a[2], int(a[3]), float(a[4]), a[5], int(a[6]), a[7] == "1")
