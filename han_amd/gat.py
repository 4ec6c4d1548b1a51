"""``models/gat.py`` surface: :class:`HeteGAT_multi`.

The reference builds a TF graph by calling ``HeteGAT_multi.inference(...)`` on
the CLASS (it is defined without ``self``, ``models/gat.py:35``; called as
``model.inference(...)`` with ``model = HeteGAT_multi``, ``ex_acm3025.py:31,139``);
variables are created implicitly on that first call.  Here the same call works
on the class (a process-wide default instance plays the role of TF's default
graph) or on an instance; parameters are created on the first call from the
shapes of the arguments, with the reference's initialisers, and live in ONE
flat fp32 buffer (so that Adam and the gradient all-reduce are one launch /
one collective).
"""
from __future__ import annotations

import functools
import math

import torch
import torch.nn.functional as F_torch

from . import layers, ops, rng
from .base_gattn import BaseGAttN
from .dist import NodePartition
from .graph import CSRGraph, as_graph

D = ops.D


class _ClassOrInstance:
    """Lets ``Cls.method(...)`` run on a lazily created default instance."""

    def __init__(self, fn):
        self.fn = fn
        functools.update_wrapper(self, fn)

    def __get__(self, obj, objtype=None):
        if obj is None:
            obj = objtype.default_instance()
        return functools.partial(self.fn, obj)


GROUP = ops.D        # the K1 / K2 kernels process heads in groups whose concatenated width is 64 columns
FP_SIZES = (4, 8, 16, 32, 64)


def _kernel_head_width(FP: int) -> int:
    """The lane-mapped head width the K1/K2 kernels run a head of width FP at: the next of 4, 8, 16, 32, 64
    (the extra columns carry zero weights: their scores terms and outputs are exactly 0 and are cut off).
    Heads wider than 64 columns run as column slices of 64 (layers.WideHeadAttention)."""
    for w in FP_SIZES:
        if FP <= w:
            return w
    return GROUP


def _head_groups(K: int, FP: int):
    """Heads [k0, k1) per kernel launch group: 64 // (kernel head width) heads each; the last group may be short."""
    kg = GROUP // _kernel_head_width(FP)
    return [(k0, min(k0 + kg, K)) for k0 in range(0, K, kg)]


# name -> shape builder; order == flat layout == gradient all-reduce layout.
# `extra` = [(K_i, FP_i)] for the node-attention layers i >= 1 (models/gat.py:48-57):
# their variables are named W_i, a1_i, ... and sit right after layer 0's.  Every width is the
# reference's own: layer i has K_i * FP_i output columns, whatever that product is.
def _param_shapes(P, F, K, FP, A, C, HC, extra=(), residual=False):
    d_prev = K * FP
    shapes = [("W", (P, F, d_prev)), ("a1", (P, K, FP)), ("b1", (P, K)), ("a2", (P, K, FP)),
              ("b2", (P, K)), ("c", (P, d_prev))]
    for i, (Ki, FPi) in enumerate(extra, start=1):
        d_i = Ki * FPi
        shapes += [(f"W_{i}", (P, d_prev, d_i)), (f"a1_{i}", (P, Ki, FPi)), (f"b1_{i}", (P, Ki)),
                   (f"a2_{i}", (P, Ki, FPi)), (f"b2_{i}", (P, Ki)), (f"c_{i}", (P, d_i))]
        if residual and FPi != d_prev:      # utils/layers.py:38-40: conv1d(seq, F', 1) per head
            shapes += [(f"Wr_{i}", (P, d_prev, d_i)), (f"br_{i}", (P, d_i))]
        d_prev = d_i
    return shapes + [("w_omega", (d_prev, A)), ("b_omega", (A,)), ("u_omega", (A,)),
                     ("Wc", (HC, d_prev, C)), ("bc", (HC, C))]


class HeteGAT_multi(BaseGAttN, torch.nn.Module):
    """models/gat.py:34-77."""

    _default = None

    def __init__(self, device=None):
        torch.nn.Module.__init__(self)
        self._built = False
        self._device = torch.device(device) if device is not None else None
        self.partition: NodePartition | None = None
        self.halo_plans = (None, None)        # (forward, backward) per-meta-path HaloPlan lists
        self.masked_bwd = None                # per-meta-path MaskedBackwardPlan list (HANTrainer(masked_backward=True))
        self.path_streams = None              # one stream per meta-path (HANTrainer(use_graph=True)): layers._on_path
        self.path_order = "index"             # "heavy": the per-meta-path chains of a captured epoch are issued largest graph first (layers._path_order)
        self.overlap_branch = None            # HANTrainer(overlap_eval=True): the eval forward riding in the captured training step (trainer._EvalBranch)
        self.side_stream = None               # second stream of the eager training step on large graphs (layers._on_side):
                                              # dW of meta-path p runs beside the backward gather of meta-path p + 1
        self._graph_cache: dict = {}
        # captured-step mode (HANTrainer(use_graph=True)): a device seed word that the trainer
        # bumps between replays + fixed per-(layer, meta-path) seed constants (han_hip.h "Seeds")
        self.step_seed_dev: torch.Tensor | None = None
        self._fixed_seeds: dict = {}

    @classmethod
    def default_instance(cls):
        if cls._default is None:
            cls._default = cls()
        return cls._default

    @classmethod
    def reset_default(cls):
        """tf.reset_default_graph() analogue."""
        cls._default = None

    # ------------------------------------------------------------------ params
    def build(self, n_metapaths, ft_size, nb_classes, hid_units=(8,), n_heads=(8, 1),
              mp_att_size=128, device=None, generator: torch.Generator | None = None,
              table_dtype=torch.float32, residual=False):
        """Create the variables the reference's first inference() call creates
        (SURVEY.md 8a): glorot-uniform conv1d/dense kernels, zero biases,
        N(0, 0.1^2) semantic-attention variables."""
        if len(n_heads) != len(hid_units) + 1:
            raise ValueError("n_heads needs one entry per hidden layer plus the output entry "
                             "(ex_acm3025.py:27)")
        K, FP, HC = int(n_heads[0]), int(hid_units[0]), int(n_heads[-1])
        self.extra = [(int(n_heads[i]), int(hid_units[i])) for i in range(1, len(hid_units))]
        # residual only ever reaches the layers >= 1 (models/gat.py:43-45 hard-codes False for
        # layer 0) and only acts when the input width differs from the head width
        self.residual = bool(residual)
        for Ki, FPi in [(K, FP)] + self.extra:
            # any number of heads; head widths are the lane-mapped sizes of the K2 kernels; a layer's
            # concatenated width K*F' is served in 64-column head groups (K1/K2) and, for the last
            # layer, zero-padded to 64 or 128 columns for K3 / the classifier
            # -- or, for a head wider than 64 columns, in 64-column slices of one head (layers.WideHeadAttention)
            if FPi < 1 or Ki < 1:
                raise ValueError(f"hid_units entries and n_heads must be positive (got {FPi}, {Ki})")
        if mp_att_size < 1 or nb_classes < 1:
            raise ValueError("mp_att_size and nb_classes must be positive")
        dev = torch.device(device) if device is not None else (self._device or torch.device("cuda:0"))
        self.P, self.F, self.K, self.FP = int(n_metapaths), int(ft_size), K, FP
        self.A, self.C, self.HC = int(mp_att_size), int(nb_classes), HC
        if table_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("table_dtype must be torch.float32 or torch.bfloat16")
        self.table_dtype = table_dtype       # storage of the projected rows / backward tables
        shapes = _param_shapes(self.P, self.F, K, FP, self.A, self.C, HC, self.extra, self.residual)
        total = sum(math.prod(s) for _, s in shapes)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.register_buffer("flat", flat, persistent=False)        # storage owner
        self.register_buffer("flat_grad", torch.zeros_like(flat), persistent=False)
        self._views = {}
        off = 0
        for name, shp in shapes:
            n = math.prod(shp)
            v = torch.nn.Parameter(self.flat[off:off + n].view(shp), requires_grad=True)
            v.grad = self.flat_grad[off:off + n].view(shp)
            self._views[name] = v
            self.register_parameter(name, v)
            off += n
        g = generator

        def uni(t, limit):
            t.data.copy_((torch.rand(t.shape, generator=g, dtype=torch.float32) * 2 - 1) * limit)

        def nrm(t, std):
            t.data.copy_(torch.randn(t.shape, generator=g, dtype=torch.float32) * std)

        uni(self.W, math.sqrt(6.0 / (self.F + FP)))          # conv1d kernel (1,F,F')
        uni(self.a1, math.sqrt(6.0 / (FP + 1)))              # conv1d kernel (1,F',1)
        uni(self.a2, math.sqrt(6.0 / (FP + 1)))
        d_prev = K * FP
        for i, (Ki, FPi) in enumerate(self.extra, start=1):
            uni(getattr(self, f"W_{i}"), math.sqrt(6.0 / (d_prev + FPi)))
            uni(getattr(self, f"a1_{i}"), math.sqrt(6.0 / (FPi + 1)))
            uni(getattr(self, f"a2_{i}"), math.sqrt(6.0 / (FPi + 1)))
            if self.residual and FPi != d_prev:
                uni(getattr(self, f"Wr_{i}"), math.sqrt(6.0 / (d_prev + FPi)))
            d_prev = Ki * FPi
        self.D_out = d_prev                                  # width of final_embed (models/gat.py:61)
        nrm(self.w_omega, 0.1)                               # utils/layers.py:145-147
        nrm(self.b_omega, 0.1)
        nrm(self.u_omega, 0.1)
        uni(self.Wc, math.sqrt(6.0 / (d_prev + self.C)))     # tf.layers.dense kernel
        self._built = True
        return self

    def trainable(self):
        return [self._views[n] for n, _ in self.param_shapes()]

    def _apply(self, fn, recurse=True):
        """nn.Module.to()/.cuda()/.float()...: move the FLAT buffers and re-create every parameter as a
        view of the moved buffer.  (The stock _apply would hand each parameter its own copy, silently
        detaching it from `flat`: Adam would then update a buffer the forward no longer reads.)"""
        if not self._built:
            return super()._apply(fn, recurse)
        new_flat = fn(self.flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("han_amd parameters are float32 (the kernels read fp32 parameters); "
                            "bf16 storage is selected with build(table_dtype=torch.bfloat16)")
        new_grad = fn(self.flat_grad)
        self._buffers["flat"], self._buffers["flat_grad"] = new_flat, new_grad
        off = 0
        for name, shp in self.param_shapes():
            n = math.prod(shp)
            v = self._views[name]
            v.data = new_flat[off:off + n].view(shp)
            v.grad = new_grad[off:off + n].view(shp)
            off += n
        self._graph_cache.clear()
        return self

    def param_shapes(self):
        return _param_shapes(self.P, self.F, self.K, self.FP, self.A, self.C, self.HC, self.extra,
                             self.residual)

    def direct_grads(self, flag: bool = True):
        """Let the backward kernels write each parameter's gradient straight into its slice of
        `flat_grad` (overwrite, no autograd accumulation): valid when every parameter is used
        once per step and `loss.backward()` starts at the loss, which is HANTrainer's step."""
        for v in self._views.values():
            v._han_direct_grad = bool(flag)
        return self

    def zero_grad_flat(self):
        """Zero the flat gradient buffer and (re)bind every .grad to its slice."""
        self.flat_grad.zero_()
        self._rebind_grads()

    def _rebind_grads(self):
        off = 0
        for name, shp in self.param_shapes():
            n = math.prod(shp)
            self._views[name].grad = self.flat_grad[off:off + n].view(shp)
            off += n

    # ---------------------------------------------------------------- forward
    def _graphs(self, bias_mat_list, device):
        out = []
        for b in bias_mat_list:
            if isinstance(b, CSRGraph):
                out.append(b if b.device == device else b.to(device))
                continue
            key = id(b)
            hit = self._graph_cache.get(key)
            if hit is None or hit[0] is not b:
                hit = (b, as_graph(b, device))     # dense mask -> CSR once per tensor identity
                self._graph_cache[key] = hit
            out.append(hit[1])
        return out

    def node_level(self, xs, graphs, attn_drop, ffd_drop, train, act_code, graphs_t=None, coef_sink=None,
                   post=None, xs_full=None):
        """models/gat.py:39-60: every node-attention layer of every meta-path -> M (N,P,K*F').
        coef_sink: a list that receives, per meta-path, the head-mean coefficients of the
        FIRST layer (models/gat.py:143-172), or None.
        post: an arbitrary `activation` callable (models/gat.py:36 takes any): the kernels then emit
        the pre-activation (act_code = identity) and torch applies `post` -- per head, on
        (N,P,K,F') so that an activation acting on the last axis sees what the reference's
        per-head (1,N,F') tensor gives it -- with its own autograd.
        xs_full: under a node partition, the features of ALL rows (P tensors (N,F)): the first layer
        then projects the whole table on every rank instead of exchanging it (replicated projection)."""
        P = len(graphs)

        def act(M, K, FP):
            if post is None:
                return M
            return post(M.reshape(M.shape[0], M.shape[1], K, FP)).reshape(M.shape)

        def seeds(layer):
            if not train:
                return (0,) * P
            if self.step_seed_dev is None:
                return tuple(rng.next_seed() for _ in graphs)
            key = (layer, P)
            if key not in self._fixed_seeds:
                self._fixed_seeds[key] = tuple(rng.next_seed() for _ in graphs)
            return self._fixed_seeds[key]

        def cfg(layer, sd, **kw):
            return {"train": train, "in_drop": float(ffd_drop), "coef_drop": float(attn_drop),
                    "seeds": sd, "seed_dev": self.step_seed_dev if train else None,
                    "act": act_code, "part": self.partition, "graphs_t": graphs_t, "layer": layer,
                    "table_dtype": self.table_dtype, "plans_f": self.halo_plans[0],
                    "plans_b": self.halo_plans[1], "xs_full": xs_full if layer == 0 else None,
                    "masked_bwd": self.masked_bwd if (layer == 0 and train) else None,
                    "streams": self.path_streams, "side_stream": self.side_stream, "path_order": self.path_order,
                    "overlap": self.overlap_branch if (layer == 0 and train) else None, **kw}

        def layer_fwd(layer, Xin, xs_, K, FP, sink):
            """One node-attention layer: its K heads run through the 64-column K1/K2 kernels in groups of
            64 // F' heads.  A group that is not full (K*F' not a multiple of 64) is completed with
            zero-weight heads, whose output columns are exactly 0 and are cut off again; group g draws
            its dropout masks from seed + g, head index = index inside the group."""
            sfx = "" if layer == 0 else f"_{layer}"
            g = lambda n: getattr(self, n + sfx, None)
            W, a1, b1, a2, b2, c, Wr, br = (g(n) for n in ("W", "a1", "b1", "a2", "b2", "c", "Wr", "br"))
            sd = seeds(layer)
            if FP > GROUP:       # heads wider than a K1 / K2 row: one head at a time, S = ceil(F'/64) column slices
                S = -(-FP // GROUP)
                outs, coef_acc = [], None
                for k in range(K):
                    cols = slice(k * FP, (k + 1) * FP)
                    pw = lambda t: F_torch.pad(t[..., cols], (0, S * GROUP - FP)).contiguous()
                    pa = lambda t: F_torch.pad(t[:, k], (0, S * GROUP - FP)).contiguous()
                    gsink = [] if sink is not None else None
                    Mk = layers.WideHeadAttention.apply(
                        Xin, pw(W), pa(a1), b1[:, k].contiguous(), pa(a2), b2[:, k].contiguous(), pw(c),
                        pw(Wr) if Wr is not None else None, pw(br) if br is not None else None, xs_, tuple(graphs),
                        cfg(layer, tuple((s_ + k) & ((1 << 64) - 1) for s_ in sd), coef_sink=gsink, coef_mean=True,
                            group=k))
                    outs.append(Mk[:, :, :FP])
                    if sink is not None:      # (E,) per meta-path and head
                        coef_acc = gsink if coef_acc is None else [x + y for x, y in zip(coef_acc, gsink)]
                if sink is not None:
                    sink.extend(v / K for v in coef_acc)
                return torch.cat(outs, dim=2) if len(outs) > 1 else outs[0].contiguous()
            groups = _head_groups(K, FP)
            if len(groups) == 1 and K * FP == GROUP:      # the reference shapes: no slicing, direct gradients
                return layers.NodeLevelAttention.apply(Xin, W, a1, b1, a2, b2, c, Wr, br, xs_, tuple(graphs),
                                                       cfg(layer, sd, coef_sink=sink, coef_mean=True))
            FPk = _kernel_head_width(FP)          # head width the kernels run at (zero-weight columns beyond FP)
            kg = GROUP // FPk
            outs, coef_acc = [], None
            for gi, (k0, k1) in enumerate(groups):
                nh, cols = k1 - k0, slice(k0 * FP, k1 * FP)

                def pc(t):            # (..., K*FP) columns of this group -> (..., kg heads x FPk) = 64 columns
                    t = t[..., cols].reshape(t.shape[:-1] + (nh, FP))
                    return F_torch.pad(t, (0, FPk - FP, 0, kg - nh)).reshape(t.shape[:-2] + (GROUP,)).contiguous()

                def ph(t, per_head):  # (P,K,FP) / (P,K) -> (P,kg,FPk) / (P,kg)
                    t = t[:, k0:k1]
                    pad = (0, FPk - FP, 0, kg - nh) if per_head else (0, kg - nh)
                    return F_torch.pad(t, pad).contiguous()
                gsink = [] if sink is not None else None
                Mg = layers.NodeLevelAttention.apply(
                    Xin, pc(W), ph(a1, True), ph(b1, False), ph(a2, True), ph(b2, False), pc(c),
                    pc(Wr) if Wr is not None else None, pc(br) if br is not None else None, xs_, tuple(graphs),
                    cfg(layer, tuple((s_ + gi) & ((1 << 64) - 1) for s_ in sd), coef_sink=gsink, coef_mean=False,
                        group=gi))
                outs.append(Mg.reshape(Mg.shape[0], Mg.shape[1], kg, FPk)[:, :, :nh, :FP]
                            .reshape(Mg.shape[0], Mg.shape[1], nh * FP))
                if sink is not None:      # (E, kg) per meta-path -> sum over the real heads
                    part = [v[:, :nh].sum(1) for v in gsink]
                    coef_acc = part if coef_acc is None else [x + y for x, y in zip(coef_acc, part)]
            if sink is not None:
                sink.extend(v / K for v in coef_acc)
            return torch.cat(outs, dim=2) if len(outs) > 1 else outs[0].contiguous()

        M = act(layer_fwd(0, None, tuple(xs), self.K, self.FP, coef_sink), self.K, self.FP)
        for i, (Ki, FPi) in enumerate(self.extra, start=1):                     # gat.py:48-57
            M = act(layer_fwd(i, M, None, Ki, FPi, None), Ki, FPi)
        return M

    def semantic(self, M):
        """models/gat.py:61-63: SimpleAttLayer over the stacked meta-path embeddings M (N,P,D_out) ->
        (final_embed (N,D_out), att_val (N,P)); any width (layers.semantic_attention)."""
        return layers.semantic_attention(M, self.w_omega, self.b_omega, self.u_omega)

    def classify(self, Z):
        """models/gat.py:65-72: logits (N,C) = mean over the output heads of Z Wc[h] + bc[h]."""
        return layers.classifier_any(Z, self.Wc, self.bc)

    def classifier_loss(self, Z, labels, mask, weight):
        """Fused classifier + masked softmax cross-entropy + accuracy (the trainer's step):
        returns (loss, accuracy, logits)."""
        return layers.classifier_loss_any(Z, self.Wc, self.bc, labels, mask, weight)

    @_ClassOrInstance
    def inference(self, inputs_list, nb_classes, nb_nodes, training, attn_drop, ffd_drop,
                  bias_mat_list, hid_units, n_heads, activation=F_torch.elu, residual=False,
                  mp_att_size=128):
        return self._inference(inputs_list, nb_classes, nb_nodes, training, attn_drop, ffd_drop,
                               bias_mat_list, hid_units, n_heads, activation, residual, mp_att_size)

    def _inference(self, inputs_list, nb_classes, nb_nodes, training, attn_drop, ffd_drop,
                   bias_mat_list, hid_units, n_heads, activation=F_torch.elu, residual=False,
                   mp_att_size=128, coef_sink=None):
        """Same positional arguments as models/gat.py:35-37.  `inputs_list[p]`:
        (1,N,F) or (N,F) fp32 GPU tensor; `bias_mat_list[p]`: (1,N,N) additive
        mask, CSRGraph or (rowptr, colidx).  Lists are zipped, shorter wins
        (gat.py:39).  `training` and `nb_nodes` are accepted and ignored, as in
        the reference; dropout is driven by attn_drop / ffd_drop alone.
        Returns (logits (1,N,C), final_embed (N,K*F'), att_val (N,P))."""
        n = min(len(inputs_list), len(bias_mat_list))
        xs = [layers._squeeze_batch(x, f"inputs_list[{i}]") for i, x in enumerate(inputs_list[:n])]
        for x in xs:
            if not x.is_cuda:
                raise ValueError("inputs must be GPU tensors: han_amd has no CPU path")
        dev = xs[0].device
        if not self._built:
            self.build(n, xs[0].shape[1], nb_classes, hid_units, n_heads, mp_att_size, device=dev,
                       residual=residual)
        if bool(residual) != self.residual:
            raise ValueError("residual does not match the variables created by the first call")
        if n != self.P or xs[0].shape[1] != self.F or nb_classes != self.C:
            raise ValueError("arguments do not match the variables created by the first call")
        graphs = self._graphs(bias_mat_list[:n], dev)
        for x, g in zip(xs, graphs):
            if g.n_rows != x.shape[0]:
                raise ValueError(f"graph has {g.n_rows} rows, features have {x.shape[0]}")
        code, post = layers._act_code(activation)     # ELU / identity run in the K2 epilogue, anything else in torch
        train = torch.is_grad_enabled() and self.W.requires_grad
        attn_drop, ffd_drop = float(attn_drop), float(ffd_drop)
        if not train and (attn_drop > 0 or ffd_drop > 0):
            raise ValueError("dropout > 0 needs gradients enabled (training step)")
        M = self.node_level(xs, graphs, attn_drop, ffd_drop, train, code, coef_sink=coef_sink, post=post)  # gat.py:39-60
        if coef_sink is not None:
            coef_sink[:] = [torch.sparse_csr_tensor(g.rowptr, g.colidx.long(), v, (g.n_rows, g.n_cols))
                            for g, v in zip(graphs, coef_sink)]
        final_embed, att_val = self.semantic(M)                                    # gat.py:61-63
        logits = self.classify(final_embed)                                        # gat.py:65-72
        return logits[None], final_embed, att_val                                  # gat.py:76-77

    def forward(self, inputs_list, bias_mat_list, attn_drop=0.0, ffd_drop=0.0):
        if not self._built:
            raise RuntimeError("call build(...) or inference(...) first")
        return self.inference(inputs_list, self.C, None, None, attn_drop, ffd_drop, bias_mat_list,
                              [self.FP] + [e[1] for e in self.extra],
                              [self.K] + [e[0] for e in self.extra] + [self.HC], mp_att_size=self.A)


class HeteGAT(HeteGAT_multi):
    """models/gat.py:132-203: ONE feature tensor shared by all meta-paths (instead of
    inputs_list) and the optional `return_coef`."""

    _default = None

    @_ClassOrInstance
    def inference(self, inputs, nb_classes, nb_nodes, training, attn_drop, ffd_drop,
                  bias_mat_list, hid_units, n_heads, activation=F_torch.elu, residual=False,
                  mp_att_size=128, return_coef=False):
        """Returns (logits, final_embed, att_val[, coef_list]).  coef_list[p] (gat.py:170-172):
        the first layer's coefficients averaged over its heads, as a torch sparse CSR
        tensor (N,N) over the stored neighbours of meta-path p (the reference's dense
        (N,N) tensor is exactly 0 elsewhere); data only, no gradient."""
        sink = [] if return_coef else None
        out = self._inference([inputs] * len(bias_mat_list), nb_classes, nb_nodes, training, attn_drop,
                              ffd_drop, bias_mat_list, hid_units, n_heads, activation, residual,
                              mp_att_size, coef_sink=sink)
        return out + (sink,) if return_coef else out


class HeteGAT_no_coef(HeteGAT):
    """models/gat.py:78-130.  (The reference's body reads an undefined `return_coef`,
    gat.py:93, so it raises NameError when called; this is the evident intent: HeteGAT
    without the coefficient output.)"""

    _default = None

    @_ClassOrInstance
    def inference(self, inputs, nb_classes, nb_nodes, training, attn_drop, ffd_drop,
                  bias_mat_list, hid_units, n_heads, activation=F_torch.elu, residual=False,
                  mp_att_size=128):
        return self._inference([inputs] * len(bias_mat_list), nb_classes, nb_nodes, training, attn_drop,
                               ffd_drop, bias_mat_list, hid_units, n_heads, activation, residual,
                               mp_att_size)
