"""`GAT` / `SpGAT` with the reference's `inference()` signature, plus the small part of
`BaseGAttN` the Cora drivers call (masked loss, masked accuracy, L2 + Adam training step).

    GAT.inference   — models/gat.py:7-31   (dense [B,N,N] additive mask)
    SpGAT.inference — models/sp_gat.py:7-34 (sparse adjacency, batch 1)

Composition (identical in both): layer 0 has n_heads[0] heads of width hid_units[0] and
residual=False; layers 1..L-1 have n_heads[i] heads and residual=residual; hidden layers
concatenate their heads; the output layer has n_heads[-1] heads of width nb_classes with the
identity activation, residual=False, and averages them.  in_drop=ffd_drop, coef_drop=attn_drop.
Each layer is ONE fused kernel sweep over all its heads instead of n_heads attn_head calls.

The reference calls these as plain functions on the class (no instance, no self) and lets
TensorFlow create variables on the way; here the variables come from the active VariableStore
(gat_amd.layers.variable_scope), created on the first call and reused afterwards.
"""
import torch

from . import layers


class BaseGAttN:
    """Loss / metric / optimiser helpers of models/base_gattn.py used on the Cora path."""

    @staticmethod
    def loss(logits, labels):
        """base_gattn.py:6-17: mean softmax cross-entropy against one-hot (or soft) labels."""
        return -(labels * torch.log_softmax(logits, dim=-1)).sum(-1).mean(0)

    @staticmethod
    def masked_softmax_cross_entropy(logits, labels, mask):
        """base_gattn.py:140-146: per-node CE weighted by mask / mean(mask), mean over ALL nodes."""
        loss = -(labels * torch.log_softmax(logits, dim=-1)).sum(-1)
        mask = mask.to(loss.dtype)
        return (loss * (mask / mask.mean())).mean()

    @staticmethod
    def masked_accuracy(logits, labels, mask):
        """base_gattn.py:158-165."""
        correct = (logits.argmax(1) == labels.argmax(1)).to(torch.float32)
        mask = mask.to(torch.float32)
        return (correct * (mask / mask.mean())).mean()

    @staticmethod
    def accuracy(preds, labels):
        """base_gattn.py:90-93."""
        return (preds.argmax(1) == labels.argmax(1)).to(torch.float32).mean()

    @staticmethod
    def l2_penalty(parameters, l2_coef):
        """base_gattn.py:77-79: l2_coef * sum of tf.nn.l2_loss(v) = 0.5 * ||v||^2 over every
        trainable variable (the name filter there compares against names no TF variable has, so
        biases are included too)."""
        return l2_coef * sum(0.5 * (p * p).sum() for p in parameters)

    @staticmethod
    def training(parameters, lr, l2_coef):
        """base_gattn.py:75-88 as (optimizer, penalty_fn).  Adam at a CONSTANT rate: the reference
        builds an exponential-decay schedule (:82) but never passes global_step to minimize (:86),
        so the step counter stays 0 and the rate never decays."""
        parameters = list(parameters)
        opt = torch.optim.Adam(parameters, lr=lr, betas=(0.9, 0.999), eps=1e-8)
        return opt, (lambda: BaseGAttN.l2_penalty(parameters, l2_coef))


def _compose(head_fn, inputs, nb_classes, training, attn_drop, ffd_drop, graph, hid_units, n_heads,
             activation, residual):
    layers.default_store().begin()
    h = head_fn(inputs, hid_units[0], graph, activation, n_heads[0], "concat", ffd_drop, attn_drop, False, training)
    for i in range(1, len(hid_units)):
        h = head_fn(h, hid_units[i], graph, activation, n_heads[i], "concat", ffd_drop, attn_drop, residual, training)
    return head_fn(h, nb_classes, graph, "identity", n_heads[-1], "mean", ffd_drop, attn_drop, False, training)


def _is_training(training, attn_drop, ffd_drop):
    """The reference feeds dropout 0.0 at evaluation time (execute_cora.py:123) and ignores the
    `training` argument; here either signal switches dropout off."""
    if isinstance(training, bool):
        return training and (attn_drop != 0.0 or ffd_drop != 0.0)
    return attn_drop != 0.0 or ffd_drop != 0.0


class GAT(BaseGAttN):
    @staticmethod
    def inference(inputs, nb_classes, nb_nodes, training, attn_drop, ffd_drop,
                  bias_mat, hid_units, n_heads, activation=layers.elu, residual=False):
        """models/gat.py:8-31.  inputs [B,N,F]; bias_mat [B,N,N] from adj_to_bias; returns [B,N,nb_classes]."""
        def head_fn(seq, out_sz, bm, act, nh, reduce, in_drop, coef_drop, res, train):
            return layers.attn_heads(seq, out_sz, bm, act, nh, reduce, in_drop, coef_drop, res, train)
        return _compose(head_fn, inputs, nb_classes, _is_training(training, attn_drop, ffd_drop), attn_drop, ffd_drop,
                        bias_mat, hid_units, n_heads, activation, residual)


class SpGAT(BaseGAttN):
    @staticmethod
    def inference(inputs, nb_classes, nb_nodes, training, attn_drop, ffd_drop,
                  bias_mat, hid_units, n_heads, activation=layers.elu, residual=False):
        """models/sp_gat.py:8-34.  inputs [1,N,F]; bias_mat is the sparse adjacency: the
        (indices, values, shape) triple of preprocess_adj_bias or a prebuilt CSRGraph."""
        graph = layers._graph_of(bias_mat, inputs.device)

        def head_fn(seq, out_sz, g, act, nh, reduce, in_drop, coef_drop, res, train):
            return layers.sp_attn_heads(seq, out_sz, g, act, nb_nodes, nh, reduce, in_drop, coef_drop, res, train)
        return _compose(head_fn, inputs, nb_classes, _is_training(training, attn_drop, ffd_drop), attn_drop, ffd_drop,
                        graph, hid_units, n_heads, activation, residual)
