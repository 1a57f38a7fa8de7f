"""End-to-end check of the training step the reference's Cora driver performs (masked softmax
cross-entropy + L2 + Adam over SpGAT.inference, models/base_gattn.py:75-88,140-165): the loop must
learn a planted-partition graph of Cora's shape well above chance."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--sparse-features"], ["--replay"], ["--replay", "--sparse-features"]])
def test_cora_like_training_learns(extra):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_cora_like
    sys.argv = ["train_cora_like.py", "--epochs", "100", "--quiet"] + extra
    acc = train_cora_like.main()
    assert acc > 0.6, "test accuracy %.3f after 100 epochs (chance is 0.14)" % acc


def test_brain_network_autoencoder_learns():
    """GAT_AE_BNF_V2 + AE_classify_loss + Adam (models/gat.py:188-243, base_gattn.py:31-41,75-88) on
    synthetic two-class connectomes: the loss falls and held-out graphs are classified."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_bnf_ae_like
    sys.argv = ["train_bnf_ae_like.py", "--steps", "80", "--quiet"]
    first, last, acc = train_bnf_ae_like.main()
    assert last < 0.8 * first and acc > 0.8, (first, last, acc)


def test_capturable_adam_follows_the_host_counted_one():
    """TFAdam(capturable=True) keeps its step count on the device (so a captured step replays); three
    steps on the same gradients give the host-counted optimiser's parameters to fp32 rounding."""
    import torch
    from gat_amd.models import TFAdam
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    shapes = [(1433, 64), (8, 8), (64,)]
    init = [torch.randn(s, device=dev, generator=gen) for s in shapes]
    grads = [[torch.randn(s, device=dev, generator=gen) * 10.0 ** (-k) for s in shapes] for k in range(3)]
    results = []
    for capturable in (False, True):
        ps = [torch.nn.Parameter(t.clone()) for t in init]
        opt = TFAdam(ps, lr=0.005, capturable=capturable)
        for gs in grads:
            for p, g in zip(ps, gs):
                p.grad = g.clone()
            opt.step()
        results.append([p.detach().clone() for p in ps])
    for a, b, t0 in zip(*results, init):
        assert not torch.equal(a, t0)
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
