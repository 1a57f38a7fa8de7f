"""Coefficient and feature dropout inside the CSR kernels (utils/layers.py:20,22).

TensorFlow's random stream is not a parity target; what is checked is (1) that the kernels'
counter-based masks, replayed on the host with the same hash, give the oracle's result with
explicit masks — forward and every gradient, so forward and backward saw the same mask;
(2) rate 0 is an exact identity; (3) the keep frequency matches the rate."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy, ref_torch
from tests.util import assert_close, feat_keep_host, input_keep_host, keep_mask_host, layer_params, random_graph

pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("H,D,coef_drop,in_drop,storage", [(8, 8, 0.6, 0.6, "f32"), (1, 7, 0.3, 0.0, "f32"), (4, 8, 0.0, 0.5, "f32"),
                                                           (8, 8, 0.6, 0.6, "bf16"), (2, 16, 0.5, 0.0, "bf16")])
def test_masks_replay_in_forward_and_backward(H, D, coef_drop, in_drop, storage):
    from gat_amd import CSRGraph, ops
    dev = torch.device("cuda:0")
    n, F = 300, 16
    dp = ops.padded_width(D)
    dst, src = random_graph(n, 6.0, seed=H, hubs=2, hub_deg=200)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    E = len(col)
    p = layer_params(F, H, D, seed=3)
    x = np.random.default_rng(4).normal(0, 1, (n, F))
    seed = 123456789
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    cfg = ops.make_cfg(H, D, "elu", "concat", coef_drop=coef_drop, in_drop=in_drop, training=True, seed=seed, storage=storage)
    graph = CSRGraph.from_coo(dst, src, n, dev, chunk=64)
    out = ops.sparse_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph, cfg)
    dout = np.random.default_rng(5).normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))

    coef_mask = ft_mask = None
    if coef_drop > 0:
        idx = (np.arange(E, dtype=np.uint64)[:, None] * np.uint64(H) + np.arange(H, dtype=np.uint64)[None, :])
        coef_mask = keep_mask_host(seed, idx, 1.0 - coef_drop).astype(np.float64)
        assert abs(coef_mask.mean() - (1.0 - coef_drop)) < 0.03
    if in_drop > 0:
        cols = (np.arange(H)[:, None] * dp + np.arange(D)[None, :]).astype(np.uint64)          # padded column ids
        idx = np.arange(n, dtype=np.uint64)[:, None, None] * np.uint64(H * dp) + cols[None]
        ft_mask = feat_keep_host(seed, idx, 1.0 - in_drop).astype(np.float64)
        assert abs(ft_mask.mean() - (1.0 - in_drop)) < 0.03
    rp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = ref_torch.layer_csr(rx, rp, torch.tensor(rowptr), torch.tensor(col), "elu", "concat",
                              coef_mask=None if coef_mask is None else torch.tensor(coef_mask),
                              ft_mask=None if ft_mask is None else torch.tensor(ft_mask),
                              coef_keep=1.0 - coef_drop, ft_keep=1.0 - in_drop)
    ref.backward(torch.tensor(dout))
    # bf16 gather tables: same masks, the 2e-2 class of that storage mode
    tol = dict(rtol=2e-2, atol_scale=2e-2) if storage == "bf16" else dict(atol_scale=5e-6)
    fwd_tol = dict(rtol=2e-2, atol_scale=2e-2) if storage == "bf16" else {}
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what="forward with dropout", **fwd_tol)
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), what="dX", **tol)
    for k in ("W", "a1", "a2", "bias"):
        assert_close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), what="d" + k, **tol)


def test_rate_zero_and_eval_are_exact_identities():
    from gat_amd import CSRGraph, ops
    dev = torch.device("cuda:0")
    n, F, H, D = 150, 10, 8, 8
    dst, src = random_graph(n, 4.0, seed=9)
    p = layer_params(F, H, D, seed=1)
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in p.items()}
    x = torch.randn(n, F, device=dev)
    g = CSRGraph.from_coo(dst, src, n, dev)
    run = lambda cfg: ops.sparse_gat_layer(x, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], g, cfg)
    base = run(ops.make_cfg(H, D, "elu"))
    assert torch.equal(base, run(ops.make_cfg(H, D, "elu", coef_drop=0.0, in_drop=0.0, training=True)))
    assert torch.equal(base, run(ops.make_cfg(H, D, "elu", coef_drop=0.6, in_drop=0.6, training=False)))
    dropped = run(ops.make_cfg(H, D, "elu", coef_drop=0.6, in_drop=0.6, training=True, seed=5))
    assert not torch.equal(base, dropped)
    assert torch.equal(dropped, run(ops.make_cfg(H, D, "elu", coef_drop=0.6, in_drop=0.6, training=True, seed=5)))
    assert not torch.equal(dropped, run(ops.make_cfg(H, D, "elu", coef_drop=0.6, in_drop=0.6, training=True, seed=6)))


@pytest.mark.parametrize("n,F,H,D,res,x_drop", [(300, 16, 8, 8, False, 0.6), (5000, 100, 8, 8, False, 0.6), (4500, 64, 1, 7, False, 0.3),
                                                 (700, 512, 4, 16, True, 0.5), (4200, 36, 2, 8, True, 0.6)])
def test_input_dropout_inside_the_projection_replays(n, F, H, D, res, x_drop):
    """Dropout of the layer's input (utils/layers.py:9) applied by the projection kernels to the tiles of X they
    load — every projection kernel family: the whole-row kernel with resident weights (n >= 4096, F <= 128), the
    slab kernel, its split-K form (F >= 256), the residual projection, the whole-line weight gradient and the
    masked store of the input gradient.  The host twin of the hash gives the mask; the oracle is fed X * mask / keep."""
    from gat_amd import CSRGraph, ops
    dev = torch.device("cuda:0")
    dst, src = random_graph(n, 4.0, seed=F, hubs=1, hub_deg=100)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    p = layer_params(F, H, D, seed=13, res=res)
    x = np.random.default_rng(14).normal(0, 1, (n, F))
    seed = 24681357
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    assert ops.input_dropout_fusable(xt)
    cfg = ops.make_cfg(H, D, "elu", "concat", training=True, seed=seed, x_drop=x_drop)
    graph = CSRGraph.from_coo(dst, src, n, dev)
    out = ops.sparse_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph, cfg,
                               tp.get("W_res"), tp.get("b_res"))
    dout = np.random.default_rng(15).normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    keep = 1.0 - x_drop
    xm = input_keep_host(seed, n, F, keep)
    assert abs(xm.mean() - keep) < 0.02
    rp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = ref_torch.layer_csr(rx * torch.tensor(xm.astype(np.float64)) / keep, rp, torch.tensor(rowptr), torch.tensor(col), "elu", "concat", residual=res)
    ref.backward(torch.tensor(dout))
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what="forward with input dropout")
    got_dx = xt.grad.cpu().numpy()
    assert np.all(got_dx[~xm] == 0.0)
    assert_close(got_dx, rx.grad.numpy(), atol_scale=5e-6, what="dX")
    for k in p:
        if k in ("b1", "b2"):
            continue
        assert_close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), atol_scale=5e-6, what="d" + k)
    # rate 0 draws no mask: bit-identical to the layer without the option
    base = ops.sparse_gat_layer(xt.detach(), tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph, ops.make_cfg(H, D, "elu", "concat"),
                                tp.get("W_res"), tp.get("b_res"))
    zero = ops.sparse_gat_layer(xt.detach(), tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph,
                                ops.make_cfg(H, D, "elu", "concat", training=True, seed=seed, x_drop=0.0), tp.get("W_res"), tp.get("b_res"))
    assert torch.equal(base, zero)


def test_heads_module_drops_its_input_inside_the_projection_when_it_can():
    """GATHeads with in_drop > 0: an aligned input takes the fused mask (no dropped copy of X), an input whose width
    is not a multiple of four falls back to a separate dropout pass; both train (finite gradients, masks change with the seed)."""
    from gat_amd import CSRGraph, layers
    dev = torch.device("cuda:0")
    n = 500
    dst, src = random_graph(n, 4.0, seed=2)
    graph = CSRGraph.from_coo(dst, src, n, dev)
    for F in (24, 23):
        m = layers.GATHeads(F, 8, 8, device=dev)
        x = torch.randn(n, F, device=dev, generator=torch.Generator(device=dev).manual_seed(F), requires_grad=True)
        a = m(x, graph, "elu", "concat", in_drop=0.5, coef_drop=0.0, training=True, seed=7)
        b = m(x, graph, "elu", "concat", in_drop=0.5, coef_drop=0.0, training=True, seed=7)
        c = m(x, graph, "elu", "concat", in_drop=0.5, coef_drop=0.0, training=True, seed=8)
        if F % 4 == 0:
            assert torch.equal(a, b)                       # in-kernel masks are a function of the seed
        assert not torch.equal(a, c)
        a.sum().backward()
        assert torch.isfinite(x.grad).all() and float((x.grad == 0).float().mean()) > 0.3


@pytest.mark.parametrize("path", ["dense", "lists"])
@pytest.mark.parametrize("B,N,H,D,coef_drop,in_drop", [(2, 60, 8, 8, 0.6, 0.6), (1, 700, 8, 8, 0.5, 0.3), (3, 33, 1, 7, 0.4, 0.0),
                                                       (256, 28, 8, 8, 0.6, 0.6), (257, 20, 2, 32, 0.3, 0.5), (256, 24, 4, 16, 0.0, 0.6)])
def test_dense_layer_masks_replay(B, N, H, D, coef_drop, in_drop, path):
    """Same check for the dense kernels (the LDS-resident and the chunked matrix-core variants, and
    the compacted-list kernels, which draw the very same masks): masks are
    keyed on (graph, i, j, head) and (graph, j, padded column).  Batches of 256 graphs and more take the
    graph-resident list kernels, dropout included (the feature mask is applied once per row while a graph is staged)."""
    from gat_amd import ops
    if path == "lists" and H * ops.padded_width(D) < 32:
        pytest.skip("the list kernels take rows of at least 32 floats")
    dev = torch.device("cuda:0")
    F = 12
    dp = ops.padded_width(D)
    rng = np.random.default_rng(N)
    adj = (rng.random((B, N, N)) < 0.1) | np.eye(N, dtype=bool)[None]
    bm = np.where(adj, 0.0, -1e9)
    p = layer_params(F, H, D, seed=2)
    x = rng.normal(0, 1, (B, N, F))
    seed = 987654321
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    cfg = ops.make_cfg(H, D, "elu", "concat", coef_drop=coef_drop, in_drop=in_drop, training=True, seed=seed, mask_path=path)
    out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"],
                              torch.tensor(bm, dtype=torch.float32, device=dev), cfg)
    dout = rng.normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    coef_mask = ft_mask = None
    if coef_drop > 0:
        b, i, j, h = np.meshgrid(np.arange(B), np.arange(N), np.arange(N), np.arange(H), indexing="ij")
        idx = (((b * N + i) * N + j) * H + h).astype(np.uint64)
        coef_mask = keep_mask_host(seed, idx, 1.0 - coef_drop).astype(np.float64).transpose(0, 3, 1, 2)   # [B,H,N,N]
    if in_drop > 0:
        cols = (np.arange(H)[:, None] * dp + np.arange(D)[None, :]).astype(np.uint64)
        idx = np.arange(B * N, dtype=np.uint64)[:, None, None] * np.uint64(H * dp) + cols[None]
        ft_mask = feat_keep_host(seed, idx, 1.0 - in_drop).astype(np.float64).reshape(B, N, H, D)
    rp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = ref_torch.layer_dense(rx, rp, torch.tensor(bm), "elu", "concat",
                                coef_mask=None if coef_mask is None else torch.tensor(coef_mask),
                                ft_mask=None if ft_mask is None else torch.tensor(ft_mask),
                                coef_keep=1.0 - coef_drop, ft_keep=1.0 - in_drop)
    ref.backward(torch.tensor(dout))
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what="dense forward with dropout")
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what="dense dX")
    for k in ("W", "a1", "a2", "bias"):
        assert_close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), atol_scale=5e-6, what="dense d" + k)
