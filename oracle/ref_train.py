"""CPU restatement of the reference's training-step arithmetic around the attention layers
(TEST INFRASTRUCTURE: imported by tests/ only, never by gat_amd/).

What is restated, in differentiable torch ops on whatever dtype the inputs carry (the tests use
float64):

  masked_softmax_cross_entropy   models/base_gattn.py:140-146 (commented out in this fork, called at
                                 execute_cora.py:74 and execute_cora_sparse.py:88)
  masked_accuracy                models/base_gattn.py:158-165 (called at execute_cora.py:75)
  l2_loss_sum                    models/base_gattn.py:77-79: tf.nn.l2_loss(v) = sum(v**2) / 2 over the
                                 trainable variables, times l2_coef.  The name filter there compares
                                 v.name with 'bias', 'gamma', ... and TensorFlow's variable names look
                                 like 'conv1d/kernel:0', so it never excludes anything: biases count.
  adam_step                      models/base_gattn.py:83-86: tf.train.AdamOptimizer(learning_rate) with
                                 TensorFlow's defaults beta1 = 0.9, beta2 = 0.999, epsilon = 1e-8.  The
                                 arithmetic lives in TensorFlow 1.x (README.md:33 pins tensorflow-gpu
                                 1.6.0; not installed here), whose documented update is
                                     lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
                                     m_t  = beta1 * m + (1 - beta1) * g
                                     v_t  = beta2 * v + (1 - beta2) * g * g
                                     var -= lr_t * m_t / (sqrt(v_t) + epsilon)
                                 ("epsilon hat" of the Adam paper: epsilon is added to the UNcorrected
                                 sqrt(v_t); torch.optim.Adam adds it to the bias-corrected one, which is
                                 a different number for small gradients).  The learning rate is constant:
                                 :82 builds an exponential decay of global_step but :86 never passes
                                 global_step to minimize(), so the counter stays 0.

PARITY UNPINNED: the reference ships no vectors for any of this; these follow the formulas above.
"""
import math

import torch


def masked_softmax_cross_entropy(logits, labels, mask):
    loss = -(labels * torch.log_softmax(logits, dim=-1)).sum(-1)      # tf.nn.softmax_cross_entropy_with_logits, :142
    mask = mask.to(loss.dtype)                                        # :143
    mask = mask / mask.mean()                                         # :144
    return (loss * mask).mean()                                       # :145-146


def masked_accuracy(logits, labels, mask):
    correct = (logits.argmax(1) == labels.argmax(1)).to(logits.dtype)     # :160-161
    mask = mask.to(logits.dtype)
    mask = mask / mask.mean()                                             # :162-163
    return (correct * mask).mean()                                        # :164-165


def l2_loss_sum(variables, l2_coef):
    return sum((v * v).sum() / 2 for v in variables) * l2_coef           # :78-79


def adam_step(variables, grads, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One tf.train.AdamOptimizer update in place.  state: dict with 't' and per-variable 'm', 'v'
    lists (created on the first call)."""
    if not state:
        state.update(t=0, m=[torch.zeros_like(v) for v in variables], v=[torch.zeros_like(v) for v in variables])
    state["t"] += 1
    t = state["t"]
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    with torch.no_grad():
        for var, g, m, v in zip(variables, grads, state["m"], state["v"]):
            m.mul_(beta1).add_(g, alpha=1.0 - beta1)
            v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            var.sub_(lr_t * m / (v.sqrt() + eps))
