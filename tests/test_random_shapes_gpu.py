"""Randomised shape sweep: layer shapes, graph sizes, degree skews, chunk sizes and mask densities
drawn from a seed, every draw checked against the float64 oracle (forward, dX, dW, da1).  The
number of draws is small in the suite; GATMI_SWEEP=<n> runs more."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch
from tests.util import assert_close, dense_dropout_masks, input_keep_host, layer_params, random_graph, sparse_dropout_masks

pytestmark = pytest.mark.gpu
DRAWS = int(os.environ.get("GATMI_SWEEP", "10"))
KINK = 2e-6       # logits closer to LeakyReLU's kink than fp32 rounding can resolve (|f1|, |f2| reach ~10: eps * 10)
RELU_KINK = 1e-6  # the same for the layer's pre-activation under a ReLU output (O(1) values computed in fp32)
HEAD_SHAPES = [(8, 8), (1, 7), (4, 8), (2, 16), (1, 64), (8, 4), (1, 4), (16, 8), (4, 64), (2, 3), (8, 32), (1, 100)]


def _oracle_params(p):
    return {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}


def _close(got, ref, what, **kw):
    """assert_close, except that a reference that is pure rounding noise (a gradient that is exactly
    zero in exact arithmetic, e.g. da1 when every row attends to one entry) only asks for ~0."""
    if ref.size and np.abs(ref).max() < 1e-9:
        assert np.abs(got).max() < 1e-5, what       # fp32 noise of the logit-gradient identity (dropout scales it by 1/keep)
    else:
        assert_close(got, ref, what=what, **kw)


def _min_abs_logit(x, p, pairs):
    """Smallest |f1_i + f2_j| over the entries `pairs` = (i, j): LeakyReLU has a kink at 0, where an
    fp32 and an fp64 evaluation may take different slopes and the gradients legitimately differ by
    0.8 * alpha * (dalpha - c); draws that land within rounding distance of it are redrawn."""
    F = p["W"].shape[0]
    H, D = p["a1"].shape
    wh = (x.reshape(-1, F) @ p["W"].reshape(F, H * D)).reshape(-1, H, D)
    f1 = (wh * p["a1"]).sum(-1) + p["b1"]
    f2 = (wh * p["a2"]).sum(-1) + p["b2"]
    i, j = pairs
    return float(np.abs(f1[i] + f2[j]).min()) if len(i) else 1.0


@pytest.mark.parametrize("seed", range(DRAWS))
def test_sparse_layer_random_draw(seed):
    from gat_amd import CSRGraph, ops
    from oracle import ref_numpy
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1000 + seed)
    H, D = HEAD_SHAPES[rng.integers(len(HEAD_SHAPES))]
    n = int(rng.integers(1, 900))
    F = int(rng.integers(1, 70))
    deg = float(rng.uniform(0.5, 12))
    hubs = int(rng.integers(0, 3))
    chunk = int(rng.choice([4, 32, 256, 2048]))
    activation = str(rng.choice(["elu", "relu", "identity"]))
    reduce = "mean" if activation == "identity" and rng.random() < 0.5 else "concat"
    res = bool(rng.random() < 0.3) and F != D      # equal widths: the reference's residual is a no-op (layers.py:31-32)
    dst, src = random_graph(n, deg, seed=seed, hubs=hubs, hub_deg=min(n, 400))
    p = layer_params(F, H, D, seed=seed + 1, res=res)
    coef_drop, in_drop = (float(rng.choice([0.0, 0.3, 0.6])), float(rng.choice([0.0, 0.5]))) if rng.random() < 0.35 else (0.0, 0.0)
    # dropout of the layer's input inside the projection kernels, for the widths that allow it
    x_drop = float(rng.choice([0.3, 0.6])) if (F % 4 == 0 and rng.random() < 0.5) else 0.0
    xkeep = input_keep_host(4242 + seed, n, F, 1.0 - x_drop).astype(np.float64) / (1.0 - x_drop) if x_drop else 1.0
    xmask = torch.tensor(xkeep) if x_drop else None
    x = rng.normal(0, 1, (n, F))
    for _ in range(50):
        if _min_abs_logit(x * xkeep, p, (dst, src)) > KINK:      # the logits are those of the input as the layer sees it
            break
        x = rng.normal(0, 1, (n, F))
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
    what = "seed %d: n=%d F=%d H=%d D=%d deg=%.1f hubs=%d chunk=%d %s/%s res=%s drop=%.1f/%.1f x_drop=%.1f" % (
        seed, n, F, H, D, deg, hubs, chunk, activation, reduce, res, coef_drop, in_drop, x_drop)
    cfg = ops.make_cfg(H, D, activation, reduce, coef_drop=coef_drop, in_drop=in_drop, training=True, seed=4242 + seed, x_drop=x_drop)
    graph = CSRGraph.from_coo(dst, src, n, dev, chunk=chunk)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    coef_mask, ft_mask = sparse_dropout_masks(4242 + seed, len(col), n, H, D, cfg.dp, coef_drop, in_drop)
    rp = _oracle_params(p)

    def oracle(rx, act):
        return ref_torch.layer_csr(rx if xmask is None else rx * xmask, rp, torch.tensor(rowptr), torch.tensor(col), act, reduce, residual=res,
                                   coef_mask=None if coef_mask is None else torch.tensor(coef_mask),
                                   ft_mask=None if ft_mask is None else torch.tensor(ft_mask),
                                   coef_keep=1.0 - coef_drop, ft_keep=1.0 - in_drop)
    if activation == "relu":
        # a pre-activation within fp32 rounding of 0 may fall on the other side of ReLU's kink than in float64:
        # that element's gradient is then all or nothing (found by draw 758 of a 1500-draw sweep, |pre| = 1.6e-7);
        # such inputs are redrawn like those at LeakyReLU's kink
        for _ in range(50):
            with torch.no_grad():
                pre = oracle(torch.tensor(x, dtype=torch.float64), "identity")
            if float(pre.abs().min()) > RELU_KINK and _min_abs_logit(x * xkeep, p, (dst, src)) > KINK:
                break
            x = rng.normal(0, 1, (n, F))
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    out = ops.sparse_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph, cfg, tp.get("W_res"), tp.get("b_res"))
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = oracle(rx, activation)
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what=what + " forward")
    dout = rng.normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    ref.backward(torch.tensor(dout))
    _close(xt.grad.cpu().numpy(), rx.grad.numpy(), what + " dX", atol_scale=5e-6)
    for k in ("W", "a1", "bias") + (("W_res",) if res else ()):
        _close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), what + " d" + k, atol_scale=5e-6)


@pytest.mark.parametrize("seed", range(DRAWS))
def test_dense_layer_random_draw(seed):
    from gat_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2000 + seed)
    H, D = [(8, 8), (1, 7), (4, 8), (2, 16), (1, 64), (8, 4), (1, 4), (4, 16), (2, 32), (1, 30)][rng.integers(10)]
    B = int(rng.integers(1, 6))
    N = int(rng.integers(1, 400))
    F = int(rng.integers(1, 40))
    density = float(rng.choice([0.02, 0.1, 0.5, 1.0]))
    soft = bool(rng.random() < 0.3)
    activation = str(rng.choice(["elu", "relu", "identity"]))
    adj = (rng.random((B, N, N)) < density) | np.eye(N, dtype=bool)[None]
    bm = np.where(adj, rng.normal(0, 1, (B, N, N)) if soft else 0.0, -1e9)
    p = layer_params(F, H, D, seed=seed + 3)
    x = rng.normal(0, 1, (B, N, F))
    bb, ii, jj = np.nonzero(adj)
    for _ in range(50):
        if _min_abs_logit(x, p, (bb * N + ii, bb * N + jj)) > KINK:
            break
        x = rng.normal(0, 1, (B, N, F))
    hd = H * ops.padded_width(D)
    paths = ["dense", "lists"] if 32 <= hd <= 64 else (["lists"] if hd > 64 else ["dense"])
    coef_drop, in_drop = (float(rng.choice([0.0, 0.3, 0.6])), float(rng.choice([0.0, 0.5]))) if (rng.random() < 0.35 and N <= 150) else (0.0, 0.0)
    coef_mask, ft_mask = dense_dropout_masks(777 + seed, B, N, H, D, ops.padded_width(D), coef_drop, in_drop)
    for path in paths:
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
        what = "seed %d: B=%d N=%d F=%d H=%d D=%d density=%.2f soft=%s %s %s drop=%.1f/%.1f" % (
            seed, B, N, F, H, D, density, soft, activation, path, coef_drop, in_drop)
        cfg = ops.make_cfg(H, D, activation, mask_path=path, coef_drop=coef_drop, in_drop=in_drop, training=True, seed=777 + seed)
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"],
                                  torch.tensor(bm, dtype=torch.float32, device=dev), cfg)
        rp = _oracle_params(p)
        rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        ref = ref_torch.layer_dense(rx, rp, torch.tensor(bm), activation, "concat",
                                    coef_mask=None if coef_mask is None else torch.tensor(coef_mask),
                                    ft_mask=None if ft_mask is None else torch.tensor(ft_mask),
                                    coef_keep=1.0 - coef_drop, ft_keep=1.0 - in_drop)
        assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what=what + " forward")
        if activation == "relu":
            with torch.no_grad():
                pre = ref_torch.layer_dense(rx, rp, torch.tensor(bm), "identity", "concat",
                                            coef_mask=None if coef_mask is None else torch.tensor(coef_mask),
                                            ft_mask=None if ft_mask is None else torch.tensor(ft_mask),
                                            coef_keep=1.0 - coef_drop, ft_keep=1.0 - in_drop)
            if float(pre.abs().min()) < RELU_KINK:
                continue                    # ReLU's kink, as in the sparse draw above: forward checked, gradients not comparable
        dout = rng.normal(0, 1, tuple(out.shape))
        out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
        ref.backward(torch.tensor(dout))
        _close(xt.grad.cpu().numpy(), rx.grad.numpy(), what + " dX", atol_scale=5e-6)
        for k in ("W", "a1", "bias"):
            _close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), what + " d" + k, atol_scale=5e-6)


@pytest.mark.parametrize("seed", range(max(1, DRAWS // 2)))
def test_graph_resident_random_draw(seed):
    """Batches of many small graphs (the graph-resident kernels): rows of 64 floats in every head split, graphs of
    4..120 nodes, hard and soft masks, dropout of coefficients / projected features / the input, through the mask and —
    for hard masks — through its prebuilt lists (bit-identical), against the float64 oracle with host-twin masks."""
    from gat_amd import ops
    from gat_amd.graph import MaskLists
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5000 + seed)
    H, D = [(8, 8), (4, 16), (2, 32), (1, 64), (8, 7), (4, 13)][rng.integers(6)]
    dp = ops.padded_width(D)
    B = int(rng.integers(256, 300))
    N = 4 * int(rng.integers(1, 31)) if rng.random() < 0.6 else int(rng.integers(4, 121))   # (any N: prebuilt lists only; the mask then runs other kernels)
    F = int(rng.choice([3, 8, 17, 24]))
    density = float(rng.choice([0.05, 0.15, 0.4] if N <= 60 else [0.05, 0.15]))     # (a million logits seldom all clear the kink)
    soft = bool(rng.random() < 0.25)
    activation = str(rng.choice(["elu", "identity"]))
    adj = (rng.random((B, N, N)) < density) | np.eye(N, dtype=bool)[None]
    bm = np.where(adj, rng.normal(0, 1, (B, N, N)) if soft else 0.0, -1e9)
    p = layer_params(F, H, D, seed=seed + 5)
    coef_drop, in_drop = (float(rng.choice([0.0, 0.3, 0.6])), float(rng.choice([0.0, 0.5]))) if rng.random() < 0.5 else (0.0, 0.0)
    x_drop = float(rng.choice([0.3, 0.6])) if (F % 4 == 0 and rng.random() < 0.5) else 0.0
    sd = 999 + seed
    coef_mask, ft_mask = dense_dropout_masks(sd, B, N, H, D, dp, coef_drop, in_drop)
    xkeep = (input_keep_host(sd, B * N, F, 1.0 - x_drop).astype(np.float64) / (1.0 - x_drop)).reshape(B, N, F) if x_drop else 1.0
    xmask = torch.tensor(xkeep) if x_drop else None
    x = rng.normal(0, 1, (B, N, F))
    bb, ii, jj = np.nonzero(adj)
    for _ in range(200):
        if _min_abs_logit(x * xkeep, p, (bb * N + ii, bb * N + jj)) > KINK:
            break
        x = rng.normal(0, 1, (B, N, F))
    else:
        pytest.skip("no draw clear of LeakyReLU's kink")
    what = "seed %d: B=%d N=%d F=%d H=%d D=%d density=%.2f soft=%s %s drop=%.1f/%.1f x_drop=%.1f" % (
        seed, B, N, F, H, D, density, soft, activation, coef_drop, in_drop, x_drop)
    bt = torch.tensor(bm, dtype=torch.float32, device=dev)
    rp = _oracle_params(p)
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = ref_torch.layer_dense(rx if xmask is None else rx * xmask, rp, torch.tensor(bm), activation, "concat",
                                coef_mask=None if coef_mask is None else torch.tensor(coef_mask),
                                ft_mask=None if ft_mask is None else torch.tensor(ft_mask),
                                coef_keep=1.0 - coef_drop, ft_keep=1.0 - in_drop)
    dout = rng.normal(0, 1, tuple(ref.shape))
    ref.backward(torch.tensor(dout))
    results = []
    # (one head of 64 floats has no list-walking backward kernel for sizes that are not a multiple of 4)
    for mask in ([bt] if soft or (H == 1 and N % 4) else [bt, MaskLists(bt)]):
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
        cfg = ops.make_cfg(H, D, activation, mask_path="lists", coef_drop=coef_drop, in_drop=in_drop, training=True, seed=sd, x_drop=x_drop)
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], mask, cfg)
        out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
        assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what=what + " forward")
        _close(xt.grad.cpu().numpy(), rx.grad.numpy(), what + " dX", atol_scale=5e-6)
        for k in ("W", "a1", "a2", "bias"):
            _close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), what + " d" + k, atol_scale=5e-6)
        results.append([out.detach(), xt.grad] + [tp[k].grad for k in ("W", "a1", "a2", "b1", "b2", "bias")])
    if len(results) == 2 and D == dp and N % 4 == 0:          # (a padded head width changes nothing either, but keep the claim to the plain shapes)
        for a, b in zip(results[0][:3], results[1][:3]):            # out, dX, dW (the sums over all nodes: equal to rounding)
            assert torch.equal(a, b), what + ": prebuilt lists differ from the mask route"


@pytest.mark.parametrize("seed", range(max(1, DRAWS // 2)))
def test_decoder_and_bnf_random_draw(seed):
    from gat_amd import ops
    from gat_amd.layers import BNFAttention
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3000 + seed)
    # inner-product decoder
    B, N, F = int(rng.integers(1, 7)), int(rng.integers(1, 320)), int(rng.integers(1, 129))
    activation = str(rng.choice(["sigmoid", "identity"]))
    z = rng.normal(0, 0.5, (B, N, F))
    zt = torch.tensor(z, dtype=torch.float32, device=dev, requires_grad=True)
    what = "seed %d: decoder B=%d N=%d F=%d %s" % (seed, B, N, F, activation)
    out = ops.inner_product_decoder(zt, activation).reshape(-1)
    rz = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    ref = ref_torch.inner_product_decoder(rz, activation)
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what=what + " forward")
    dout = rng.normal(0, 1, (B * N * N,))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    ref.backward(torch.tensor(dout))
    _close(zt.grad.cpu().numpy(), rz.grad.numpy(), what + " dZ", atol_scale=5e-6)
    # brain-network attention step
    B, N, F = int(rng.integers(1, 4)), int(rng.integers(2, 280)), int(rng.integers(1, 65))
    H, Dz = int(rng.integers(1, 5)), int(rng.choice([8, 16, 64]))
    x = rng.normal(0, 1, (B, N, F))
    A = rng.random((B, N, N)) * (rng.random((B, N, N)) < rng.choice([0.05, 0.3, 1.0]))
    p = {"W": rng.normal(0, 0.4, (F, H, Dz)), "a1": rng.normal(0, 0.4, (H, Dz)), "a2": rng.normal(0, 0.4, (H, Dz)),
         "b1": rng.normal(0, 0.1, H), "b2": rng.normal(0, 0.1, H), "m1": 1 + rng.normal(0, 0.2, H), "m2": 1 + rng.normal(0, 0.2, H)}

    def min_abs_logit(x):
        # every pair (i, j) of a graph has a logit here (no mask): one of the ~N^2 H of them within fp32 rounding of
        # LeakyReLU's kink is common, and its gradient then legitimately takes the other slope (see _min_abs_logit)
        z = (x.reshape(B, N, F) @ p["W"].reshape(F, H * Dz)).reshape(B, N, H, Dz)
        zf1 = (z * p["a1"]).sum(-1) + p["b1"]
        zf2 = (z * p["a2"]).sum(-1) + p["b2"]
        zz = np.abs(zf1[:, :, None, :] + zf2[:, None, :, :])
        return float(zz.min()), float(zz.max())
    for _ in range(50):
        if min_abs_logit(x)[0] > KINK:
            break
        x = rng.normal(0, 1, (B, N, F))
    mod = BNFAttention(F, H, Dz, device=dev)
    with torch.no_grad():
        for k, v in p.items():
            getattr(mod, k).copy_(torch.tensor(v, dtype=torch.float32))
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    At = torch.tensor(A, dtype=torch.float32, device=dev, requires_grad=True)
    what = "seed %d: bnf B=%d N=%d F=%d H=%d Dz=%d" % (seed, B, N, F, H, Dz)
    out = mod(xt, At)
    rp = _oracle_params(p)
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rA = torch.tensor(A, dtype=torch.float64, requires_grad=True)
    ref = ref_torch.bnf_v2e_aggregate(rx, rA, rp)
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what=what + " forward")
    if min_abs_logit(x)[0] <= KINK:
        pytest.skip("fifty redraws all left a logit within fp32 rounding of LeakyReLU's kink")
    dout = rng.normal(0, 1, tuple(ref.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    ref.backward(torch.tensor(dout))
    _close(xt.grad.cpu().numpy(), rx.grad.numpy(), what + " dX", atol_scale=5e-6)
    _close(At.grad.cpu().numpy(), rA.grad.numpy(), what + " dA", atol_scale=5e-6)
    # The gradient of a logit half, df1_i = sum_j dlogit_ij, is a sum of terms of the size of |G||out| (tens, with 64
    # features) that cancel to something often a hundred times smaller (zero where a row's logits all lie on one side
    # of LeakyReLU's kink, which the kernel's form of the sum reproduces term by term): its fp32 error is absolute,
    # ~1e-7 of those terms per node, and dW / da1 inherit it.  Band: 1e-4 relative + 3e-5 absolute for unit-scale inputs.
    # With wide features the drawn logits reach |z| of 40 and more: the softmax is then one-hot in fp32 terms, the
    # gradient of a logit is a difference of two numbers that agree to six digits (n D_ij - c_i for the one source j
    # that holds the row's mass), and the exponent's own rounding error grows with |z|: the absolute part of the
    # band grows with the largest logit beyond 8.
    spread = max(1.0, min_abs_logit(x)[1] / 8.0)
    for k in ("W", "a1", "a2", "m1", "m2"):
        got, want = getattr(mod, k).grad.cpu().numpy().astype(np.float64), rp[k].grad.numpy()
        tol = 1e-4 * np.abs(want) + 1e-4 * np.abs(want).max() + 3e-5 * spread
        assert (np.abs(got - want) <= tol).all(), "%s d%s: worst %.3g of the band" % (what, k, (np.abs(got - want) / tol).max())
