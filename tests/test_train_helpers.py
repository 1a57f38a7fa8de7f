"""Host-side pieces of the training step (models/base_gattn.py): masked loss, masked accuracy, L2
penalty and the Adam update, on hand-computed values and against the oracle's restatement; and the
rule that decides whether dropout acts.  Plain tensor arithmetic: runs without a GPU."""
import math

import numpy as np
import torch

from gat_amd.models import BaseGAttN, TFAdam, _is_training
from oracle import ref_train


def test_masked_softmax_cross_entropy_by_hand():
    """base_gattn.py:140-146 on four nodes, two of them in the mask: the mask is divided by its mean
    (0.5), so masked nodes weigh 2 and the mean runs over all four nodes."""
    logits = torch.tensor([[2.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.5, 0.5, 0.5], [3.0, -1.0, 0.0]], dtype=torch.float64)
    labels = torch.tensor([[1.0, 0, 0], [0, 0, 1.0], [0, 1.0, 0], [0, 1.0, 0]], dtype=torch.float64)
    mask = torch.tensor([1, 0, 1, 0])
    ce0 = -math.log(math.exp(2.0) / (math.exp(2.0) + 2.0))
    ce2 = math.log(3.0)
    want = (2.0 * ce0 + 2.0 * ce2) / 4.0
    got = BaseGAttN.masked_softmax_cross_entropy(logits, labels, mask)
    assert abs(float(got) - want) < 1e-12
    assert abs(float(ref_train.masked_softmax_cross_entropy(logits, labels, mask)) - want) < 1e-12
    # an all-ones mask is the plain mean cross-entropy
    full = BaseGAttN.masked_softmax_cross_entropy(logits, labels, torch.ones(4))
    assert abs(float(full) - float(BaseGAttN.loss(logits, labels))) < 1e-12


def test_masked_accuracy_by_hand():
    """base_gattn.py:158-165: nodes 0 and 3 are predicted correctly; with mask {0, 1, 2} the weights are
    4/3 each on three nodes and the mean over four nodes is (4/3) / 4 = 1/3."""
    logits = torch.tensor([[2.0, 0.0], [0.0, 1.0], [0.7, 0.5], [-1.0, 3.0]])
    labels = torch.tensor([[1.0, 0], [1.0, 0], [0, 1.0], [0, 1.0]])
    assert abs(float(BaseGAttN.masked_accuracy(logits, labels, torch.tensor([1, 1, 1, 0]))) - 1.0 / 3.0) < 1e-6
    assert abs(float(BaseGAttN.masked_accuracy(logits, labels, torch.tensor([1, 0, 0, 1]))) - 1.0) < 1e-6
    assert abs(float(BaseGAttN.masked_accuracy(logits, labels, torch.ones(4))) - float(BaseGAttN.accuracy(logits, labels))) < 1e-7
    assert abs(float(ref_train.masked_accuracy(logits, labels, torch.tensor([1, 1, 1, 0]))) - 1.0 / 3.0) < 1e-6


def test_l2_penalty_counts_every_variable():
    """base_gattn.py:77-79: l2_coef * sum over ALL variables of sum(v^2)/2 (biases included)."""
    ps = [torch.tensor([[1.0, -2.0], [0.5, 0.0]]), torch.tensor([3.0]), torch.tensor([0.1, 0.2, 0.3])]
    want = 0.005 * (0.5 * (1 + 4 + 0.25) + 0.5 * 9 + 0.5 * (0.01 + 0.04 + 0.09))
    assert abs(float(BaseGAttN.l2_penalty(ps, 0.005)) - want) < 1e-7
    assert abs(float(ref_train.l2_loss_sum(ps, 0.005)) - want) < 1e-7


def test_adam_update_is_tensorflows():
    """First step by hand: m = 0.1 g, v = 0.001 g^2, lr_t = lr sqrt(0.001) / 0.1, so
    var -= lr * g / (|g| + 1e-8 / sqrt(0.001)) -- a gradient of 1e-7 moves the variable by about a
    quarter of lr, not by lr (which is what adding epsilon after the bias correction would give)."""
    lr = 0.005
    g = np.array([1.0, -0.5, 1e-7, -3e-6, 0.0])
    w0 = np.array([0.3, -0.2, 0.1, 0.0, 0.7])
    want = w0 - lr * g / (np.abs(g) + 1e-8 / math.sqrt(0.001))
    w = torch.nn.Parameter(torch.tensor(w0, dtype=torch.float64))
    opt, _ = BaseGAttN.training([w], lr=lr, l2_coef=0.0)
    assert isinstance(opt, TFAdam)
    w.grad = torch.tensor(g, dtype=torch.float64)
    opt.step()
    np.testing.assert_allclose(w.detach().numpy(), want, rtol=1e-12, atol=1e-15)
    assert abs((w0[2] - float(w.detach()[2])) / lr - 1e-7 / (1e-7 + 3.1622776601683795e-07)) < 1e-9
    # five more steps against the oracle's restatement, two variables of different shapes
    rng = np.random.default_rng(0)
    a, b = torch.nn.Parameter(torch.tensor(rng.normal(0, 1, (4, 3)))), torch.nn.Parameter(torch.tensor(rng.normal(0, 1, 5)))
    ra, rb = a.detach().clone(), b.detach().clone()
    opt2 = TFAdam([a, b], lr=0.01)
    state = {}
    for _ in range(5):
        ga, gb = torch.tensor(rng.normal(0, 1e-3, (4, 3))), torch.tensor(rng.normal(0, 1.0, 5))
        a.grad, b.grad = ga.clone(), gb.clone()
        opt2.step()
        ref_train.adam_step([ra, rb], [ga, gb], state, 0.01)
    np.testing.assert_allclose(a.detach().numpy(), ra.numpy(), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(b.detach().numpy(), rb.numpy(), rtol=1e-12, atol=1e-15)


def test_penalty_in_the_optimizer_takes_the_same_steps():
    """base_gattn.py:77-86: the L2 penalty reaches Adam as l2_coef * v added to each gradient; through
    autograd or added inside the optimizer, the variables after four steps are the same."""
    rng = np.random.default_rng(1)
    start = [rng.normal(0, 1, (4, 3)), rng.normal(0, 1, 5)]
    x = torch.tensor(rng.normal(0, 1, (6, 4)))
    results = []
    for fused in (False, True):
        vs = [torch.nn.Parameter(torch.tensor(t)) for t in start]
        opt, l2 = BaseGAttN.training(vs, lr=0.01, l2_coef=0.05, penalty_in_optimizer=fused)
        losses = []
        for _ in range(4):
            opt.zero_grad(set_to_none=True)
            loss = ((x @ vs[0]).tanh().sum() + (vs[1] ** 2).sum()) + l2()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        results.append(([v.detach().numpy().copy() for v in vs], losses))
    for a, b in zip(results[0][0], results[1][0]):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(results[0][1], results[1][1], rtol=1e-12)


def test_dropout_acts_on_the_rates_alone():
    """models/gat.py:8 accepts `training` and never reads it; utils/layers.py:8,19,21 test the rates."""
    assert _is_training(False, 0.6, 0.6) and _is_training(True, 0.0, 0.3) and _is_training(None, 0.1, 0.0)
    assert not _is_training(True, 0.0, 0.0) and not _is_training(False, 0.0, 0.0)
