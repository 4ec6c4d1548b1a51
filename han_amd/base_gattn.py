"""``models/base_gattn.py`` surface: loss, metric and optimiser of the reference.

* :meth:`BaseGAttN.masked_softmax_cross_entropy` -- models/base_gattn.py:41-48
* :meth:`BaseGAttN.masked_accuracy`              -- models/base_gattn.py:61-69
* :meth:`BaseGAttN.training`                     -- models/base_gattn.py:12-24

The standalone loss / metric functions are O(N*C) host-side torch expressions on
whatever device the logits live on ("host-side
PyTorch, no custom kernel").  The training loop (han_amd/trainer.py) does not
use them: it calls the fused classifier+loss kernel and the fused L2+Adam
kernel (han_classifier_loss / han_adam_step).
"""
from __future__ import annotations

import math

import torch


class TFAdam:
    """tf.train.AdamOptimizer(lr) minimising loss + l2_coef * sum_v ||v||^2/2 over
    EVERY trainable (models/base_gattn.py:14-22: the name filter never matches a
    TF variable name, so biases are regularised too), on one flat buffer:
        g' = g + l2*p;  m,v EMAs;  p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)
    (TF's epsilon placement, not torch.optim.Adam's)."""

    def __init__(self, flat_param: torch.Tensor, flat_grad: torch.Tensor, lr=0.005, l2_coef=0.0,
                 beta1=0.9, beta2=0.999, eps=1e-8):
        self.p, self.g = flat_param, flat_grad
        self.m = torch.zeros_like(flat_param)
        self.v = torch.zeros_like(flat_param)
        self.lr, self.l2, self.b1, self.b2, self.eps = lr, l2_coef, beta1, beta2, eps
        self.t = 0
        # captured-step mode: the step count lives on the device (1-element int64 tensor,
        # bumped by the trainer before each step) and the kernel forms lr_t itself
        self.step_dev: torch.Tensor | None = None

    def step(self):
        from . import ops
        self.t += 1
        if self.step_dev is not None:
            ops.adam_step(self.p, self.g, self.m, self.v, self.lr, self.b1, self.b2, self.eps, self.l2,
                          step_dev=self.step_dev)
            return
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        ops.adam_step(self.p, self.g, self.m, self.v, lr_t, self.b1, self.b2, self.eps, self.l2)

    def state_dict(self):
        return {"t": self.t, "m": self.m, "v": self.v}

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])


class BaseGAttN:
    @staticmethod
    def loss(logits, labels, nb_classes, class_weights):
        """models/base_gattn.py:5-10: class-weighted sparse softmax cross-entropy (mean over samples).
        logits (N,C); labels (N,) integer class ids; class_weights (C,)."""
        sample_wts = (torch.nn.functional.one_hot(labels.long(), nb_classes).to(logits.dtype)
                      * torch.as_tensor(class_weights, dtype=logits.dtype, device=logits.device)).sum(-1)   # :6-7
        xent = torch.nn.functional.cross_entropy(logits, labels.long(), reduction="none") * sample_wts     # :8-9
        return xent.mean()                                                                                   # :10

    @staticmethod
    def preshape(logits, labels, nb_classes):
        """models/base_gattn.py:26-31."""
        return logits.reshape(-1, nb_classes), labels.reshape(-1)

    @staticmethod
    def confmat(logits, labels):
        """models/base_gattn.py:33-35: confusion matrix, rows = labels, columns = predictions."""
        preds = logits.argmax(1)
        labels = labels.long()
        n = int(max(int(labels.max()), int(preds.max()))) + 1
        cm = torch.zeros((n, n), dtype=torch.int64, device=logits.device)
        cm.index_put_((labels, preds), torch.ones_like(labels), accumulate=True)
        return cm

    @staticmethod
    def masked_sigmoid_cross_entropy(logits, labels, mask):
        """models/base_gattn.py:50-59 (multi-label)."""
        loss = torch.nn.functional.binary_cross_entropy_with_logits(
            logits, labels.to(logits.dtype), reduction="none").mean(1)                   # :52-55
        mask = mask.to(logits.dtype)
        mask = mask / mask.mean()                                                        # :56-57
        return (loss * mask).mean()                                                      # :58-59

    @staticmethod
    def micro_f1(logits, labels, mask):
        """models/base_gattn.py:71-94."""
        predicted = torch.round(torch.sigmoid(logits)).to(torch.int64)                   # :73-76 (half to even, as tf.round)
        labels = labels.to(torch.int64)
        m = mask.to(torch.int64).unsqueeze(-1)                                           # :78-81
        tp = torch.count_nonzero(predicted * labels * m)                                 # :84
        fp = torch.count_nonzero(predicted * (labels - 1) * m)                           # :86
        fn = torch.count_nonzero((predicted - 1) * labels * m)                           # :87
        precision = tp.double() / (tp + fp).double()                                     # :90
        recall = tp.double() / (tp + fn).double()                                        # :91
        return ((2 * precision * recall) / (precision + recall)).to(torch.float32)       # :92-94

    @staticmethod
    def masked_softmax_cross_entropy(logits, labels, mask):
        """models/base_gattn.py:41-48.  logits, labels (N,C) one-hot; mask (N,)."""
        loss = -(labels.to(logits.dtype) * torch.log_softmax(logits, dim=-1)).sum(-1)   # :43-44
        mask = mask.to(logits.dtype)                                                     # :45
        mask = mask / mask.mean()                                                        # :46
        return (loss * mask).mean()                                                      # :47-48

    @staticmethod
    def masked_accuracy(logits, labels, mask):
        """models/base_gattn.py:61-69."""
        correct = (logits.argmax(1) == labels.argmax(1)).to(logits.dtype)
        mask = mask.to(logits.dtype)
        mask = mask / mask.mean()
        return (correct * mask).mean()

    @staticmethod
    def training(model, lr, l2_coef):
        """models/base_gattn.py:12-24.  TF returns a train_op for a loss tensor;
        eager code has no static loss, so this returns the optimiser bound to
        the model's flat parameter/gradient buffers: call ``loss.backward()``
        then ``opt.step()`` (see han_amd.trainer.HANTrainer for the full
        sess.run([train_op, loss, accuracy]) equivalent)."""
        return TFAdam(model.flat, model.flat_grad, lr=lr, l2_coef=l2_coef)
