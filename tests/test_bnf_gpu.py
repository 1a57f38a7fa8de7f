"""GPU parity of the brain-network attention step (utils/layers.py:174-179 with :229-237) against
the float64 oracle: forward, and gradients of x, the weighted adjacency, the projection, the logit
vectors and the two learnable scalars."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy, ref_torch
from tests.util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,F,H,Dz", [(3, 90, 1, 2, 64), (2, 246, 64, 2, 64), (2, 70, 16, 1, 64), (4, 33, 5, 2, 16),
                                         (1, 264, 64, 3, 64), (2, 90, 16, 6, 64), (2, 50, 3, 4, 32), (2, 60, 8, 5, 16)])
def test_bnf_aggregate_forward_and_gradients(B, N, F, H, Dz):
    from gat_amd.layers import BNFAttention
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(N + F)
    x = rng.normal(0, 1, (B, N, F))
    A = rng.random((B, N, N)) * (rng.random((B, N, N)) < 0.2)
    A = (A + A.transpose(0, 2, 1)) / 2
    p = {"W": rng.normal(0, 0.4, (F, H, Dz)), "a1": rng.normal(0, 0.4, (H, Dz)), "a2": rng.normal(0, 0.4, (H, Dz)),
         "b1": rng.normal(0, 0.1, H), "b2": rng.normal(0, 0.1, H), "m1": 1 + rng.normal(0, 0.2, H), "m2": 1 + rng.normal(0, 0.2, H)}
    mod = BNFAttention(F, H, Dz, device=dev)
    with torch.no_grad():
        for k, v in p.items():
            getattr(mod, k).copy_(torch.tensor(v, dtype=torch.float32))
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    At = torch.tensor(A, dtype=torch.float32, device=dev, requires_grad=True)
    out = mod(xt, At)
    ref = ref_numpy.bnf_v2e_aggregate(x, A, p)
    assert_close(out.detach().cpu().numpy(), ref, what="bnf forward")

    dout = rng.normal(0, 1, ref.shape)
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    rp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rA = torch.tensor(A, dtype=torch.float64, requires_grad=True)
    ref_torch.bnf_v2e_aggregate(rx, rA, rp).backward(torch.tensor(dout))
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what="bnf dX")
    assert_close(At.grad.cpu().numpy(), rA.grad.numpy(), atol_scale=5e-6, what="bnf dA")
    for k in ("W", "a1", "a2", "m1", "m2"):
        assert_close(getattr(mod, k).grad.cpu().numpy(), rp[k].grad.numpy(), rtol=2e-5, atol_scale=1e-5, what="bnf d" + k)


@pytest.mark.parametrize("F,out_sz,heads,residual", [(1, 16, 2, True), (8, 8, 3, False), (1, 64, 6, True)])
def test_attn_e2v_bnf_v2_composition(F, out_sz, heads, residual):
    """attn_e2v_BNF_V2 (utils/layers.py:158-227): five rounds of attention heads on the kernels with
    the reference's dense layers around them, against the float64 restatement with the same weights.
    The last case has BNF_AE.py's widths (six heads of 64) and enough node rows for the dense layers'
    parameter gradients to come from the library's row-split reductions."""
    from gat_amd.layers import AttnE2VBNFV2
    from gat_amd import ops
    dev = torch.device("cuda:0")
    B, N = (3, 50) if heads < 6 else (12, 90)
    assert (B * N >= ops.TALL_LINEAR_ROWS) == (heads == 6)
    rng = np.random.default_rng(F)
    x = rng.normal(0, 1, (B, N, F))
    A = rng.random((B, N, N)) * (rng.random((B, N, N)) < 0.2)
    A = (A + A.transpose(0, 2, 1)) / 2
    torch.manual_seed(0)
    mod = AttnE2VBNFV2(F, out_sz, heads, "relu", residual, reps=5, feat_dim=16, device=dev)
    with torch.no_grad():
        for att in mod.att:                      # non-trivial logit biases and scalars
            att.b1.normal_(0, 0.1); att.b2.normal_(0, 0.1); att.m1.add_(torch.randn_like(att.m1) * 0.1); att.m2.add_(torch.randn_like(att.m2) * 0.1)
        if heads == 6:                           # six heads of 64 at the default initialisation grow to 6e4 over the five rounds
            for w in mod.head_w:
                w.mul_(0.2)
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    At = torch.tensor(A, dtype=torch.float32, device=dev)
    out, aggall = mod(xt, At)
    assert tuple(out.shape) == (B, N, out_sz) and tuple(aggall.shape) == (B, N, 5 * out_sz)
    d = lambda t: t.detach().cpu().double().requires_grad_(True)
    rounds = []
    for rep in range(5):
        att = mod.att[rep]
        r = {k: d(getattr(att, k)) for k in ("W", "a1", "a2", "b1", "b2", "m1", "m2")}
        r.update(head_w=d(mod.head_w[rep]), head_b=d(mod.head_b[rep]), merge_w=d(mod.merge[rep].weight.t()), merge_b=d(mod.merge[rep].bias))
        if isinstance(mod.res[rep], torch.nn.Linear):
            r.update(res_w=d(mod.res[rep].weight.t()), res_b=d(mod.res[rep].bias))
        rounds.append(r)
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rout, ragg = ref_torch.attn_e2v_bnf_v2(rx, torch.tensor(A), rounds, (d(mod.out1.weight.t()), d(mod.out1.bias)),
                                           (d(mod.out2.weight.t()), d(mod.out2.bias)), "relu", residual)
    assert_close(aggall.detach().cpu().numpy(), ragg.detach().numpy(), rtol=2e-5, atol_scale=1e-5, what="e2v aggall")
    # out is normalised over the nodes per column; with 64 columns some keep a handful of non-zero nodes after the
    # ReLUs and a small norm, which magnifies the fp32 error of the dense layers before it
    wide = heads == 6
    assert_close(out.detach().cpu().numpy(), rout.detach().numpy(), rtol=2e-4 if wide else 2e-5, atol_scale=1e-4 if wide else 1e-5, what="e2v out")
    g = rng.normal(0, 1, tuple(aggall.shape))
    aggall.backward(torch.tensor(g, dtype=torch.float32, device=dev))
    ragg.backward(torch.tensor(g))
    gr, ga = (3e-4, 1.5e-4) if wide else (1e-4, 5e-5)       # five rounds deep: the wide case accumulates three times the error
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), rtol=gr, atol_scale=ga, what="e2v dX")
    assert_close(mod.att[0].W.grad.cpu().numpy(), rounds[0]["W"].grad.numpy(), rtol=gr, atol_scale=ga, what="e2v dW(repetition 0)")
    assert_close(mod.att[4].m1.grad.cpu().numpy(), rounds[4]["m1"].grad.numpy(), rtol=gr, atol_scale=ga, what="e2v dm1(repetition 4)")
    assert_close(mod.merge[1].weight.grad.t().cpu().numpy(), rounds[1]["merge_w"].grad.numpy(), rtol=gr, atol_scale=ga, what="e2v d merge_w(repetition 1)")
    assert_close(mod.merge[1].bias.grad.cpu().numpy(), rounds[1]["merge_b"].grad.numpy(), rtol=gr, atol_scale=ga, what="e2v d merge_b(repetition 1)")
    assert_close(mod.head_b[2].grad.cpu().numpy(), rounds[2]["head_b"].grad.numpy(), rtol=gr, atol_scale=ga, what="e2v d head_b(repetition 2)")
    assert_close(mod.head_w[2].grad.cpu().numpy(), rounds[2]["head_w"].grad.numpy(), rtol=gr, atol_scale=ga, what="e2v d head_w(repetition 2)")


def test_gat_ae_bnf_v2_model_runs_on_the_kernels():
    """models/gat.py:188-243 at a reduced size: output shapes, reconstructions in (0,1) and
    symmetric (sigmoid of a Gram matrix), class-activation map consistent with the logits
    (mean over nodes of cam + bias = digits), gradients reach every parameter."""
    from gat_amd.models import GAT_AE_BNF_V2
    dev = torch.device("cuda:0")
    B, N, F, C = 4, 90, 1, 2
    torch.manual_seed(1)
    net = torch.rand(B, N, N, device=dev) * (torch.rand(B, N, N, device=dev) < 0.3)
    net = (net + net.transpose(1, 2)) / 2
    x = torch.randn(B, N, F, device=dev)
    model = GAT_AE_BNF_V2(F, N, C, nb_CAM=32, hid_units=[16, 8, 8], n_heads=[2, 2, 2], device=dev)
    digits, rec_a, rec_b, prob, cam = model(x, net, training=True)
    assert tuple(digits.shape) == (B, C) and tuple(rec_a.shape) == (B, N, N) == tuple(rec_b.shape) and tuple(cam.shape) == (B, N, C)
    assert (rec_a > 0).all() and (rec_a < 1).all() and torch.allclose(rec_a, rec_a.transpose(1, 2), atol=1e-6)
    assert torch.allclose(prob.sum(-1), torch.ones(B, device=dev), atol=1e-5)
    assert torch.allclose(cam.mean(1) + model.dense_out.bias, digits, atol=1e-4)
    (digits.sum() + rec_a.mean() + rec_b.mean()).backward()
    missing = [n for n, p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not missing, missing


@pytest.mark.parametrize("n,f_in,w,strided,activation", [(5000, 384, 64, False, "relu"), (4500, 16, 16, False, "identity"), (12300, 64, 64, True, "relu"),
                                                          (3000, 1, 64, False, "identity"), (2500, 320, 64, False, "elu"), (2048, 128, 256, False, "relu"),
                                                          (1500, 10, 6, False, "relu"), (1100, 64, 7, True, "elu")])
def test_tall_linear_matches_the_plain_dense_layer(n, f_in, w, strided, activation):
    """The width-1 convolutions of utils/layers.py:180,183,187,223-224 as a dense layer over all node rows: output
    and input gradient are torch's GEMMs, the activation's backward, the bias gradient and the weight gradient come from
    the library's passes over the rows; all against a float64 evaluation."""
    from gat_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n + f_in + w)
    act = {"relu": torch.relu, "elu": torch.nn.functional.elu, "identity": lambda t: t}[activation]
    x64 = torch.tensor(rng.normal(0, 1, (n, 3 * f_in if strided else f_in)), requires_grad=True)
    W64 = torch.tensor(rng.normal(0, 0.3, (w, f_in)), requires_grad=True)
    b64 = torch.tensor(rng.normal(0, 0.3, w), requires_grad=True)
    dy = torch.tensor(rng.normal(0, 1, (n, w)))
    xs64 = x64[:, f_in:2 * f_in] if strided else x64
    ref = act(torch.nn.functional.linear(xs64, W64, b64))
    (ref * dy).sum().backward()
    x = x64.detach().to(dev, torch.float32).requires_grad_(True)
    W = W64.detach().to(dev, torch.float32).requires_grad_(True)
    b = b64.detach().to(dev, torch.float32).requires_grad_(True)
    xs = x[:, f_in:2 * f_in] if strided else x
    y = ops.tall_linear(xs, W, b, activation)
    assert y.grad_fn is not None and "TallLinearFn" in type(y.grad_fn).__name__
    assert_close(y.detach().cpu().numpy(), ref.detach().numpy(), what="tall linear forward")
    # an output within fp32 rounding of the ReLU's kink may fall on the other side of it than in float64
    clear = (ref.detach().abs() > 1e-5) if activation == "relu" else torch.ones_like(ref, dtype=torch.bool)
    assert bool(((y.detach().cpu() > 0) == (ref.detach() > 0))[clear].all())
    y.backward(dy.to(dev, torch.float32))
    assert_close(W.grad.cpu().numpy(), W64.grad.numpy(), rtol=2e-5, atol_scale=1e-5, what="tall linear dW")
    assert_close(b.grad.cpu().numpy(), b64.grad.numpy(), rtol=2e-5, atol_scale=1e-5, what="tall linear db")
    assert_close(x.grad.cpu().numpy(), x64.grad.numpy(), rtol=2e-5, atol_scale=1e-5, what="tall linear dX")
    # without a bias, and below the row limit the layer is torch's own
    y2 = ops.tall_linear(xs, W, None, activation)
    assert_close(y2.detach().cpu().numpy(), act(torch.nn.functional.linear(xs64, W64)).detach().numpy(), what="tall linear forward, no bias")
    y2.backward(dy.to(dev, torch.float32))
    small = ops.tall_linear(xs[:100], W, b, activation)
    assert small.grad_fn is not None and "TallLinearFn" not in type(small.grad_fn).__name__
    assert_close(small.detach().cpu().numpy(), ref[:100].detach().numpy(), what="small linear forward")


@pytest.mark.parametrize("shape,activation", [((50, 246, 384), "relu"), ((3, 1000, 7), "elu"), ((2048, 64), "identity")])
def test_row_bias_and_activation(shape, activation):
    from gat_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(shape[-1])
    act = {"relu": torch.relu, "elu": torch.nn.functional.elu, "identity": lambda t: t}[activation]
    x64, b64, dy = torch.tensor(rng.normal(0, 1, shape), requires_grad=True), torch.tensor(rng.normal(0, 1, shape[-1]), requires_grad=True), torch.tensor(rng.normal(0, 1, shape))
    ref = act(x64 + b64)
    (ref * dy).sum().backward()
    x = x64.detach().to(dev, torch.float32).requires_grad_(True)
    b = b64.detach().to(dev, torch.float32).requires_grad_(True)
    y = ops.add_row_bias(x, b, activation)
    assert_close(y.detach().cpu().numpy(), ref.detach().numpy(), what="row bias forward")
    y.backward(dy.to(dev, torch.float32))
    assert_close(b.grad.cpu().numpy(), b64.grad.numpy(), rtol=2e-5, atol_scale=1e-5, what="row bias db")
    assert_close(x.grad.cpu().numpy(), x64.grad.numpy(), rtol=2e-5, atol_scale=1e-5, what="row bias dX")
