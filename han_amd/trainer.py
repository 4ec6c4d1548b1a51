"""Full-graph training loop of the reference (``ex_acm3025.py:171-245``) on the
HIP kernels: one epoch = ONE fwd+bwd+Adam step on the whole graph with dropout
0.6/0.6 on the train mask, then ONE eval forward on the val mask (both `while`
loops of the script run exactly once because batch_size = 1 = number of graphs).

`sess.run([train_op, loss, accuracy], feed_dict)` becomes :meth:`train_step`,
`sess.run([loss, accuracy], feed_dict)` becomes :meth:`eval_step`.  Under a
:class:`~han_amd.dist.NodePartition` every rank owns a block of rows; the only
collectives are the table all-gathers inside the node-attention op and one
all-reduce of the flat gradient buffer.
"""
from __future__ import annotations

import os

import torch

from . import layers, ops, rng
from .base_gattn import TFAdam
from .dist import NodePartition
from .gat import HeteGAT_multi
from .graph import as_graph


class _EvalBranch:
    """HANTrainer(overlap_eval=True): the eval forward of the parameters an epoch starts with, cut in the two pieces
    that fit the training step's fork / join sections: node_level() (K1 + K2 of every meta-path -> M) runs beside
    the training forward's per-meta-path chains, head() (K3, classifier, loss; the copy of the parameters) beside
    the backward's.  Each piece forks `stream` from the current stream, runs as one chain with scratch buffers of
    its own (ops.WS_SUFFIX) and is joined by the section's own join (`join()`); with stream None the pieces run in
    place."""

    def __init__(self, trainer, stream):
        self.tr, self.stream = trainer, stream
        self.M = self.vl = self.va = None

    def _run(self, fn):
        m = self.tr.model
        streams, m.path_streams = m.path_streams, None        # one chain: K1 of every meta-path in one launch
        branch, m.overlap_branch = getattr(m, "overlap_branch", None), None
        prev, ops.WS_SUFFIX = ops.WS_SUFFIX, ops.WS_SUFFIX + "@eval"
        try:
            with torch.no_grad():
                if self.stream is None:
                    fn()
                else:
                    self.stream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self.stream):
                        fn()
        finally:
            m.path_streams, m.overlap_branch, ops.WS_SUFFIX = streams, branch, prev

    def join(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def node_level(self):
        tr = self.tr

        def fn():
            self.M = tr.model.node_level(tr.xs, tr.graphs, 0.0, 0.0, False, ops.ACT_ELU, graphs_t=tr.graphs_t)
        self._run(fn)

    def head(self):
        tr = self.tr

        def fn():
            Z, _ = tr.model.semantic(self.M)
            self.vl, self.va, _ = tr.model.classifier_loss(Z, tr.labels, tr.val_mask, tr.w_val)
            tr._flat_prev.copy_(tr.model.flat)
            self.M = None
        self._run(fn)


class HANTrainer:
    def __init__(self, model: HeteGAT_multi, xs, graphs, labels, train_mask, val_mask=None,
                 lr=0.005, l2_coef=0.001, attn_drop=0.6, ffd_drop=0.6,
                 part: NodePartition | None = None, patience=100, max_halo_fraction=0.6,
                 use_graph=False, graphs_local=False, xs_full=None, replicate="auto", masked_backward=False,
                 side_stream=False, overlap_eval=False):
        """xs: list of P (N_local,F) feature tensors (this rank's rows);
        graphs: list of P CSRGraph (or dense masks / CSR tuples).  Under a partition (`part`) either
        the GLOBAL graphs (graphs_local=False: each rank keeps its row block; small data sets) or --
        graphs_local=True, the scalable form -- THIS RANK'S destination rows [row_start, row_end) with
        global column ids, so that no rank ever holds a global graph; the transposed shards are then
        built with an all-to-all-v of the edges (NodePartition.shard_local_graph);
        labels int32 (N_local,) class ids; masks uint8/bool (N_local,).
        side_stream: False | True | "auto" -- the eager training step of a single process runs the backward's dW of
        meta-path p on a second stream beside the transposed-graph gather of meta-path p + 1 (layers._on_side): +1.3-2.5 %
        epochs/s at SYN-1M, bit-equal results.  Off by default: the gather's launch time then includes that company and
        no longer measures the gather.  "auto" = on from 262 144 table rows, where a gather is long enough to hide a dW.
        use_graph: capture one whole epoch (train step + eval forward, ~60 launches) into a
        hipGraph on its second call and replay it afterwards -- for the launch-bound small
        graphs (ACM / DBLP sizes).  The per-step dropout seed and Adam's step count then live
        in a 2-word device state that the graph itself advances (han_hip.h "Seeds").
        Single-process only.
        overlap_eval (with use_graph; False | True = "branch" | "sections"): the eval forward needs the parameters one
        training step produced and nothing else of it, and the next step's forward and backward only READ those
        parameters -- so the captured epoch runs
        the eval forward of the parameters it STARTS with as a second branch of the graph beside its own training
        forward and backward, and only Adam waits for both.  Every number is the one the plain epoch computes, the
        validation pair arrives one call later: call k returns (train loss / accuracy of step k, validation loss /
        accuracy of the parameters BEFORE step k, i.e. of epoch k - 1; the first call evaluates the initial
        parameters).  `flush_eval()` evaluates the current parameters (the validation pair of the last epoch), and
        `early_stopping` snapshots the parameters the validation pair belongs to.  For the launch-bound small
        graphs, where the eval chain (about a fifth of the epoch's dependent launches) then hides under the
        training step; on graphs that fill the machine it buys nothing.  "branch": ONE branch from the epoch's first
        node to Adam (one box, 400 replays: ACM-like 0.394 -> 0.325 ms per epoch, DBLP-like 0.859 -> 0.828 with the
        per-meta-path chains issued largest graph first, layers._path_order); "sections": the eval forward in two
        pieces inside the training step's own fork / join sections -- K1 + K2 beside the training forward's
        per-meta-path chains, K3 + classifier beside the backward's (ACM-like 0.384, DBLP-like 0.858 on another box
        whose plain epochs took 0.398 / 0.891).
        xs_full: under a partition, optionally the features of ALL rows (P tensors (N,F), the same on
        every rank): the forward passes named by `replicate` ("auto" = dist.replication_policy,
        "all", "eval", "none") then project the whole table on every rank instead of exchanging it;
        `xs` may then be None (the local rows are views of xs_full).
        masked_backward: OPT-IN (never the default, never the headline measurement): with a single node-attention
        layer the backward row g_i of every destination outside `train_mask` is identically zero, so the
        transposed-graph pass skips those destinations in place (bit-identical results, dist.MaskedBackwardPlan)
        and, under a partition, only the live rows of the backward table travel.  The reference evaluates every
        row in every step (sess.run, ex_acm3025.py:190); this mode computes the same numbers with less work."""
        if not model._built:
            raise RuntimeError("build the model first (model.build(...))")
        self.model = model
        model.direct_grads(True)      # gradients land in model.flat_grad without copy/accumulate launches
        self.part = part if (part is not None and part.active) else None
        model.partition = self.part
        dev = model.flat.device
        self.xs_full, self.replicate, self.replicate_info = None, frozenset(), None
        self._replicate_arg = replicate
        if xs_full is not None and self.part is not None:
            from .dist import replication_policy
            pt = self.part
            if any(x.shape[0] != pt.n_global for x in xs_full):
                raise ValueError("xs_full must hold the features of all %d rows" % pt.n_global)
            if xs is None:
                xs = [x[pt.row_start:pt.row_end] for x in xs_full]
            # the exchanged tables have world * shard rows (the last shard may be short): pad once
            pad = {}
            for x in xs_full:      # the P meta-paths usually share one feature tensor
                if id(x) not in pad:
                    xc = x.contiguous()
                    if pt.n_table != pt.n_global:
                        xc = torch.cat([xc, xc.new_zeros((pt.n_table - pt.n_global, xc.shape[1]))])
                    pad[id(x)] = xc
            self.xs_full = tuple(pad[id(x)] for x in xs_full)
            self.replicate = replication_policy(pt.world, replicate)
            self.replicate_info = {"policy": "static (han_amd.dist.replication_policy)"}
        self.xs = [x.contiguous() for x in xs]
        graphs = [as_graph(g, dev) for g in graphs]     # dense masks / (rowptr, colidx) accepted
        if self.part is not None:
            sharded = [self.part.shard_local_graph(g) if graphs_local else self.part.shard_graph(g)
                       for g in graphs]
            self.graphs = [s[0] for s in sharded]
            self.graphs_t = [s[1] for s in sharded]
            # halo exchange where the graph has locality, all-gather where it has none
            model.halo_plans = ([self.part.plan_exchange(g, max_halo_fraction) for g in self.graphs],
                                [self.part.plan_exchange(g, max_halo_fraction) for g in self.graphs_t])
        else:
            self.graphs = list(graphs)
            self.graphs_t = [g.transpose() for g in graphs]
            model.halo_plans = (None, None)
        if self.xs_full is not None and self._replicate_arg == "auto" and "HAN_REPLICATE" not in os.environ:
            self._calibrate_replication(ffd_drop)
        self.labels = labels.to(device=dev, dtype=torch.int32).contiguous()
        self.train_mask = train_mask.to(device=dev, dtype=torch.uint8).contiguous()
        self._masked_plans = None
        self.set_masked_backward(masked_backward)
        self.val_mask = (val_mask if val_mask is not None else train_mask).to(
            device=dev, dtype=torch.uint8).contiguous()
        # mean(loss * mask / mean(mask)) == sum(mask * loss) / count(mask): the
        # normaliser is a setup-time constant, summed over ranks once.
        self.w_train = 1.0 / max(self._global_count(self.train_mask), 1)
        self.w_val = 1.0 / max(self._global_count(self.val_mask), 1)
        self.opt = TFAdam(model.flat, model.flat_grad, lr=lr, l2_coef=l2_coef)
        # d loss = 1 handed to backward() instead of the ones_like() autograd would launch per step: in a captured epoch
        # of the small graphs that 1-element fill is a dependent 5-us launch between the loss and the K3 backward
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        self.attn_drop, self.ffd_drop = attn_drop, ffd_drop
        self.patience = patience
        self.vlss_mn, self.vacc_mx, self.curr_step = float("inf"), 0.0, 0
        self.best_state = None
        self.use_graph = bool(use_graph)
        if overlap_eval not in (False, True, "branch", "sections"):
            raise ValueError(f"overlap_eval = {overlap_eval!r}: expected False, True, 'branch' or 'sections'")
        self.overlap_eval = bool(overlap_eval)
        self.overlap_form = "sections" if overlap_eval == "sections" else "branch"
        if self.overlap_eval and not self.use_graph:
            raise ValueError("overlap_eval reorders the captured epoch: it needs use_graph=True")
        self._eval_stream = None
        self._force_segments = False  # tests: run the two-piece flow of the captured epoch in place (any backend)
        self._flat_prev = None        # overlap_eval: the parameters the returned validation pair was computed with
        self._capture = True          # tests switch this off to run the same device-state flow eagerly
        self._graph = None
        self._static_out = None
        self._graph_calls = 0
        n_rows = int(self.graphs[0].n_rows) if len(self.graphs) else 0
        want_side = (n_rows >= 262144) if side_stream == "auto" else bool(side_stream)
        if want_side and not self.use_graph and self.part is None and dev.type == "cuda" and len(self.graphs) > 1:
            model.side_stream = torch.cuda.Stream(device=dev)
        if self.use_graph:
            if self.part is not None:
                raise NotImplementedError("use_graph is for single-process training")
            # [seed word, Adam step count]; advanced by the first op of every epoch
            self.step_state = torch.zeros(2, dtype=torch.int64, device=dev)
            self.step_state[1] = self.opt.t
            self._step_inc = torch.tensor([-0x61C8864680B583EB, 1], dtype=torch.int64, device=dev)  # 0x9E37...15
            model.step_seed_dev = self.step_state[0:1]
            self.opt.step_dev = self.step_state[1:2]
            # one stream per meta-path: inside the captured epoch the per-meta-path chains (K1 -> K2, and the four
            # backward kernels) become parallel branches of the graph -- at these sizes every kernel is a few
            # microseconds on a few CUs (layers._on_path); HAN_PATH_STREAMS=0 keeps the single chain
            if dev.type == "cuda" and os.environ.get("HAN_PATH_STREAMS", "1") != "0" and len(self.graphs) > 1:
                model.path_streams = [torch.cuda.Stream(device=dev) for _ in self.graphs]
            if self.overlap_eval:
                model.path_order = "heavy"
                self._flat_prev = torch.empty_like(model.flat)
                if dev.type == "cuda":
                    self._eval_stream = torch.cuda.Stream(device=dev)

    def set_masked_backward(self, flag: bool):
        """Switch the opt-in masked backward on or off (plans are built on first use; collective under a partition)."""
        self.masked_backward = bool(flag)
        if not self.masked_backward:
            self.model.masked_bwd = None
            return
        if getattr(self, "use_graph", False) and self._graph is not None:
            raise RuntimeError("the epoch is already captured; choose masked_backward before the first epoch")
        if self.model.extra:
            raise NotImplementedError("masked_backward needs a single node-attention layer: with more, the rows "
                                      "outside the mask feed the rows inside it")
        if self._masked_plans is None:
            from .dist import MaskedBackwardPlan
            live_global = self.part.all_gather_rows(self.train_mask) if self.part is not None else None
            self._masked_plans = [MaskedBackwardPlan(self.part, gt, self.train_mask, live_global)
                                  for gt in self.graphs_t]
        self.model.masked_bwd = self._masked_plans

    def _calibrate_replication(self, ffd_drop, reps=3):
        """replicate="auto" over RCCL: MEASURE, on this machine and this shape, what a forward table costs to
        exchange (one all-gather of a rank's projected rows) against what it costs to recompute (K1 over all
        rows minus K1 over the local rows, with and without dropout), take the maximum over the ranks and
        replicate exactly the forward passes whose projection is cheaper than their exchange.  Other
        backends (the gloo rehearsals, whose timings mean nothing) keep the static policy."""
        import torch.distributed as dist
        pt = self.part
        if not (dist.is_available() and dist.is_initialized() and dist.get_backend(pt.group) == "nccl"):
            return
        if all(pl is not None for pl in self.model.halo_plans[0]):
            return                                  # every meta-path has a halo plan: nothing to decide
        m, dev = self.model, self.model.flat.device
        Xf, Xl = self.xs_full[0], self.xs[0]
        F = Xf.shape[1]
        W = torch.zeros((F, ops.D), device=dev)     # K1's time does not depend on the values
        a, b = torch.zeros((8, 8), device=dev), torch.zeros(8, device=dev)

        def ms(fn):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return sorted(ts)[len(ts) // 2]

        def k1(X, drop, off):
            return ops.project_fwd(X, W, a, a, b, b, in_drop=drop, fts_drop=drop, seed=1, row_offset=off,
                                   table_dtype=m.table_dtype)

        with torch.no_grad():
            t = [ms(lambda: k1(Xl, ffd_drop, pt.row_start)), ms(lambda: k1(Xf, ffd_drop, 0)),
                 ms(lambda: k1(Xl, 0.0, pt.row_start)), ms(lambda: k1(Xf, 0.0, 0))]
            H_loc = k1(Xl, 0.0, pt.row_start)[0]
            t.append(ms(lambda: pt.all_gather_rows_async(H_loc, ("calib",)).wait()))
        tt = torch.tensor(t, device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=pt.group)
        loc_tr, full_tr, loc_ev, full_ev, gather = (float(v) for v in tt.tolist())
        rep = set()
        if full_tr - loc_tr < gather:
            rep.add("train")
        if full_ev - loc_ev < gather:
            rep.add("eval")
        self.replicate = frozenset(rep)
        self.replicate_info = {"policy": "measured at set-up (max over ranks, ms)",
                               "all_gather_of_one_forward_table": round(gather, 3),
                               "extra_projection_training": round(full_tr - loc_tr, 3),
                               "extra_projection_eval": round(full_ev - loc_ev, 3)}
        pt.drop_buffers(("calib",))

    def _global_count(self, mask):
        c = mask.sum().to(torch.float32).reshape(1)
        if self.part is not None:
            self.part.all_reduce_sum_(c)
        return int(c.item())

    def _forward(self, train: bool, mask, weight):
        m = self.model
        with torch.set_grad_enabled(train):
            M = m.node_level(self.xs, self.graphs, self.attn_drop if train else 0.0,
                             self.ffd_drop if train else 0.0, train, ops.ACT_ELU,
                             graphs_t=self.graphs_t,
                             xs_full=self.xs_full if ("train" if train else "eval") in self.replicate else None)
            Z, _ = m.semantic(M)
            loss, acc, _ = m.classifier_loss(Z, self.labels, mask, weight)
        return loss, acc

    def train_step(self):
        """sess.run([train_op, loss, accuracy]) with drop = 0.6 (ex_acm3025.py:178-193).
        Returns device scalars (this rank's share of) loss and accuracy."""
        self.model.zero_grad_flat()
        loss, acc = self._forward(True, self.train_mask, self.w_train)
        loss.backward(self._one)
        if self.part is not None:
            self.part.all_reduce_sum_(self.model.flat_grad)
        self.opt.step()
        return loss.detach(), acc

    def eval_step(self, mask=None, weight=None):
        """sess.run([loss, accuracy]) with drop = 0.0 (ex_acm3025.py:199-218)."""
        mask = self.val_mask if mask is None else mask
        weight = self.w_val if weight is None else weight
        loss, acc = self._forward(False, mask, weight)
        return loss, acc

    def epoch(self):
        """One reference epoch; returns device tensors (no host sync).  With use_graph the
        tensors are the graph's static outputs: the next epoch() overwrites them."""
        if self.use_graph:
            return self._epoch_graph()
        tl, ta = self.train_step()
        vl, va = self.eval_step()
        return tl, ta, vl, va

    def _epoch_body(self):
        self.step_state.add_(self._step_inc)
        if self.overlap_eval:
            return self._epoch_body_overlapped()
        tl, ta = self.train_step()
        vl, va = self.eval_step()
        return tl, ta, vl, va

    def _eval_branch(self):
        """The eval forward of the CURRENT parameters as ONE chain (launched eagerly: the warm-up epoch, the tests'
        eager flow, the CPU stand-in backend), and the copy of those parameters that early_stopping checkpoints."""
        br = _EvalBranch(self, None)
        br.node_level()
        br.head()
        return br.vl, br.va

    def _epoch_body_overlapped(self):
        if not self._force_segments and (self._eval_stream is None or not torch.cuda.is_current_stream_capturing()):
            # launched eagerly (the warm-up epoch, which also BUILDS the graphs' cached row lists and transposes on the
            # stream it runs on): one chain, same order of values
            vl, va = self._eval_branch()
            tl, ta = self.train_step()
            return tl, ta, vl, va
        if self.overlap_form == "branch" and not self._force_segments:
            # one branch from the epoch's first node to Adam, beside the whole training forward and backward
            br = _EvalBranch(self, self._eval_stream)
            br.node_level()
            br.head()
            self.model.zero_grad_flat()
            loss, acc = self._forward(True, self.train_mask, self.w_train)
            loss.backward(self._one)
            br.join()                                         # Adam overwrites what the branch reads
            self.opt.step()
            return loss.detach(), acc, br.vl, br.va
        # "sections": the eval forward rides in the training step's own fork / join sections (layers.NodeLevelAttention:
        # K1 + K2 of all meta-paths beside the training forward's per-meta-path chains, K3 + classifier beside the
        # backward's) -- a series-parallel graph like the plain epoch's
        br = _EvalBranch(self, None if self._force_segments else self._eval_stream)
        m = self.model
        m.overlap_branch = br
        try:
            tl, ta = self.train_step()
        finally:
            m.overlap_branch = None
        if br.vl is None:
            raise NotImplementedError("overlap_eval: the first node-attention layer did not run through "
                                      "layers.NodeLevelAttention (head shapes beyond 8 x 8 columns)")
        return tl, ta, br.vl, br.va

    def flush_eval(self):
        """overlap_eval: the validation pair of the CURRENT parameters (the one the next epoch() would return),
        launched eagerly; a no-op alias of eval_step() otherwise."""
        if self.overlap_eval and self._flat_prev is not None:
            vl, va = self.eval_step()
            self._flat_prev.copy_(self.model.flat)
            return vl, va
        return self.eval_step()

    def _epoch_graph(self):
        self._graph_calls += 1
        if self._graph is not None:
            self.opt.t += 1                      # the replayed han_adam_step reads the device count
            self._graph.replay()
            return self._static_out
        if self._graph_calls == 1 or not self._capture or not self.model.flat.is_cuda:
            # warm-up epoch (a real one) on a side stream: sizes every workspace, builds the
            # cached transposes / row splits, lets autograd set up its buffers
            if not self.model.flat.is_cuda:
                return self._epoch_body()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = self._epoch_body()
            torch.cuda.current_stream().wait_stream(side)
            return out
        from . import _lib
        g = torch.cuda.CUDAGraph()
        calls0 = _lib.CALLS
        with torch.cuda.graph(g):
            self._static_out = self._epoch_body()   # recorded, not run (opt.t advanced by step())
        self.graph_abi_calls = _lib.CALLS - calls0   # C-ABI calls recorded into the epoch (each is 1-3 kernel nodes)
        self._graph = g
        g.replay()
        return self._static_out

    def reduce_metrics(self, *vals):
        """Sum per-rank partial losses/accuracies (they are already weighted by
        the global mask count) -- logging only, not needed for training."""
        t = torch.stack([v.reshape(()) for v in vals])
        if self.part is not None:
            self.part.all_reduce_sum_(t)
        return [float(x) for x in t.tolist()]

    def early_stopping(self, val_loss: float, val_acc: float) -> bool:
        """ex_acm3025.py:225-240: checkpoint when val acc >= best AND val loss <= best;
        stop after `patience` epochs without either improving.  Returns True to stop."""
        if val_acc >= self.vacc_mx or val_loss <= self.vlss_mn:
            if val_acc >= self.vacc_mx and val_loss <= self.vlss_mn:
                # overlap_eval: the pair belongs to the parameters before the step that ran beside it
                src = self._flat_prev if self.overlap_eval else self.model.flat
                self.best_state = src.detach().clone()
            self.vacc_mx = max(val_acc, self.vacc_mx)
            self.vlss_mn = min(val_loss, self.vlss_mn)
            self.curr_step = 0
            return False
        self.curr_step += 1
        return self.curr_step == self.patience

    def _signature(self):
        m = self.model
        return {"param_shapes": [(n, list(s)) for n, s in m.param_shapes()], "residual": bool(m.residual),
                "table_dtype": str(m.table_dtype), "use_graph": self.use_graph}

    def save_checkpoint(self, path):
        """saver.save(sess, checkpt_file) (ex_acm3025.py:229) -- plus what the reference does not keep
        and a bit-exact resume needs: the optimiser state, the dropout seed stream (host counter, or the
        device step word of a captured epoch), and the early-stopping bookkeeping incl. the best weights."""
        ck = {"flat": self.model.flat.detach().cpu(),
              "opt": {"t": self.opt.t, "m": self.opt.m.cpu(), "v": self.opt.v.cpu()},
              "signature": self._signature(),
              "rng": dict(rng._state),
              "fixed_seeds": {f"{k[0]}:{k[1]}": list(v) for k, v in self.model._fixed_seeds.items()},
              "step_state": self.step_state.cpu() if self.use_graph else None,
              "early_stop": {"vlss_mn": self.vlss_mn, "vacc_mx": self.vacc_mx, "curr_step": self.curr_step},
              "best_state": self.best_state.cpu() if self.best_state is not None else None}
        torch.save(ck, path)

    def load_checkpoint(self, path):
        ck = torch.load(path, weights_only=True)
        sig = self._signature()
        if ck.get("signature") != sig:
            raise ValueError("checkpoint was written for a different model / trainer configuration: "
                             f"{ck.get('signature')} vs {sig}")
        m = self.model
        m.flat.copy_(ck["flat"])
        self.opt.load_state_dict(ck["opt"])
        rng._state.update(ck["rng"])
        fixed = {(int(k.split(":")[0]), int(k.split(":")[1])): tuple(v) for k, v in ck["fixed_seeds"].items()}
        if self._graph is not None and fixed != m._fixed_seeds:
            raise RuntimeError("this trainer has already captured its epoch with other per-meta-path seed "
                               "constants baked in; load the checkpoint before the first epoch")
        m._fixed_seeds = fixed
        if self.use_graph:
            self.step_state.copy_(ck["step_state"])
            self.step_state[1] = self.opt.t
        es = ck["early_stop"]
        self.vlss_mn, self.vacc_mx, self.curr_step = float(es["vlss_mn"]), float(es["vacc_mx"]), int(es["curr_step"])
        self.best_state = ck["best_state"].to(m.flat.device) if ck["best_state"] is not None else None

    def restore_best(self):
        """saver.restore(sess, checkpt_file) (ex_acm3025.py:247)."""
        if self.best_state is not None:
            self.model.flat.copy_(self.best_state)
