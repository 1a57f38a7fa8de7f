"""gatmi_masked_ce (masked softmax cross-entropy + gradient + masked accuracy in one sweep, models/base_gattn.py:140-165)
against the float64 restatement in oracle/ref_train.py and torch autograd on the host."""
import numpy as np
import pytest
import torch

from oracle import ref_train

pytestmark = pytest.mark.gpu


def _case(n, C, seed, soft=False, mask_kind="some"):
    rng = np.random.default_rng(seed)
    z = rng.normal(0, 2.0, (n, C))
    if soft:
        y = rng.random((n, C)) * (rng.random((n, C)) < 0.5)
    else:
        y = np.eye(C)[rng.integers(0, C, n)]
    if mask_kind == "none":
        m = None
    elif mask_kind == "ones":
        m = np.ones(n)
    else:
        m = (rng.random(n) < 0.3).astype(np.float64)
        m[0] = 1.0
    return z, y, m


def _reference(z, y, m, upstream=1.0):
    zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    yt = torch.tensor(y, dtype=torch.float64)
    mt = torch.ones(len(z), dtype=torch.float64) if m is None else torch.tensor(m, dtype=torch.float64)
    loss = ref_train.masked_softmax_cross_entropy(zt, yt, mt)
    (loss * upstream).backward()
    return float(loss), float(ref_train.masked_accuracy(zt.detach(), yt, mt)), zt.grad.numpy()


@pytest.mark.parametrize("n,C,soft,mask_kind", [(1000, 7, False, "some"), (5000, 47, False, "some"), (333, 47, True, "ones"),
                                                 (64, 3, False, "none"), (2000, 130, True, "some"), (1, 5, False, "ones"), (4097, 256, False, "some")])
def test_fused_loss_gradient_and_accuracy_match_the_oracle(n, C, soft, mask_kind):
    from gat_amd import ops
    dev = torch.device("cuda:0")
    z, y, m = _case(n, C, n + C, soft, mask_kind)
    want_loss, want_acc, want_grad = _reference(z, y, m, upstream=2.5)
    zt = torch.tensor(z, dtype=torch.float32, device=dev, requires_grad=True)
    yt = torch.tensor(y, dtype=torch.float32, device=dev)
    mt = None if m is None else torch.tensor(m, dtype=torch.float32, device=dev)
    loss, acc = ops.masked_cross_entropy(zt, yt, mt)
    (loss * 2.5).backward()
    assert abs(float(loss) - want_loss) <= 2e-6 * max(1.0, abs(want_loss))
    assert abs(float(acc) - want_acc) <= 1e-6
    g = zt.grad.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(g - want_grad) <= 1e-5 * np.abs(want_grad) + 2e-6 * np.abs(want_grad).max())
    # the same through the reference's two functions, and bit-reproducible
    from gat_amd.models import BaseGAttN
    again = BaseGAttN.masked_softmax_cross_entropy(zt.detach(), yt, torch.ones(n, device=dev) if mt is None else mt)
    assert float(again) == float(loss)
    assert float(BaseGAttN.masked_accuracy(zt.detach(), yt, torch.ones(n, device=dev) if mt is None else mt)) == float(acc)


def test_fused_loss_reads_padded_rows_in_place_and_returns_their_layout():
    """The averaged output head hands out the first C columns of padded rows: the loss reads them in place (what lies
    in the padding is never touched, not even NaN) and its gradient goes to the padded tensor directly, zeros in the padding."""
    from gat_amd import ops
    dev = torch.device("cuda:0")
    n, C, ld = 777, 47, 64
    z, y, m = _case(n, C, 5)
    want_loss, _, want_grad = _reference(z, y, m)
    base = torch.full((n, ld), float("nan"), dtype=torch.float32, device=dev)
    base[:, :C] = torch.tensor(z, dtype=torch.float32, device=dev)
    base.requires_grad_(True)
    logits = (base * 1.0)[:, :C]                             # a view of a padded autograd node, as sparse_gat_layer returns it
    assert ops.padded_base(logits, C) is logits._base
    loss, _ = ops.masked_cross_entropy(logits, torch.tensor(y, dtype=torch.float32, device=dev), torch.tensor(m, dtype=torch.float32, device=dev))
    loss.backward()
    assert abs(float(loss) - want_loss) <= 2e-6 * abs(want_loss)
    g = base.grad.cpu().numpy().astype(np.float64)
    assert np.all(g[:, C:] == 0.0)
    assert np.all(np.abs(g[:, :C] - want_grad) <= 1e-5 * np.abs(want_grad) + 2e-6 * np.abs(want_grad).max())
    # a view that is not the leading columns of its base is copied instead
    shifted = base.detach().nan_to_num(0.0)[:, 8:8 + 40]
    assert ops.padded_base(shifted, 40) is not shifted._base and ops.padded_base(shifted, 40).is_contiguous()


def test_arg_max_ties_take_the_lowest_class_and_an_empty_mask_gives_zero():
    from gat_amd import ops
    dev = torch.device("cuda:0")
    z = torch.tensor([[1.0, 3.0, 3.0, 0.0, 3.0], [2.0, 2.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 9.0]], device=dev)
    y = torch.tensor([[0.0, 1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.5, 0.5]], device=dev)
    _, acc = ops.masked_cross_entropy(z, y, None)
    assert abs(float(acc) - 1.0 / 3.0) < 1e-7             # row 0: classes 1 = 1; row 1: 0 vs 1; row 2: 4 vs 3 (first of the tie)
    loss, acc = ops.masked_cross_entropy(z, y, torch.zeros(3, device=dev))
    assert float(loss) == 0.0 and float(acc) == 0.0


def test_two_layer_training_step_uses_the_fused_loss_and_matches_torch_ops():
    """SpGAT logits -> fused loss -> backward through both layers, against the same step with the loss written as
    torch operations on a copy of the network."""
    from gat_amd import CSRGraph, layers
    from gat_amd.models import SpGAT
    from tests.util import random_graph
    dev = torch.device("cuda:0")
    n, F, C = 600, 20, 11
    dst, src = random_graph(n, 5.0, 3, hubs=2, hub_deg=200)
    graph = CSRGraph.from_coo(dst, src, n, dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(1, n, F, device=dev, generator=gen)
    y = torch.nn.functional.one_hot(torch.randint(0, C, (n,), device=dev, generator=gen), C).float()
    mask = (torch.rand(n, device=dev, generator=gen) < 0.2).float()
    kw = dict(hid_units=[8], n_heads=[8, 1], activation=layers.elu, residual=False)
    store = layers.VariableStore()

    def logits():
        with layers.variable_scope(store):
            return SpGAT.inference(x, C, n, False, 0.0, 0.0, bias_mat=graph, **kw)[0]
    lg = logits()
    assert lg._base is not None and lg._base.shape == (n, 16)
    fused = SpGAT.masked_softmax_cross_entropy(lg, y, mask)
    fused.backward()
    got = [p.grad.clone() for p in store.parameters()]
    for p in store.parameters():
        p.grad = None
    lg2 = logits()
    ce = -(y * torch.log_softmax(lg2, dim=-1)).sum(-1)
    plain = (ce * (mask / mask.mean())).mean()
    plain.backward()
    assert abs(float(fused) - float(plain)) <= 2e-6 * abs(float(plain))
    for a, p in zip(got, store.parameters()):
        assert torch.allclose(a, p.grad, rtol=2e-5, atol=2e-6 * float(p.grad.abs().max()))
