"""Randomised shape sweep: layer shapes, graph sizes, degree skews, chunk sizes and mask densities
drawn from a seed, every draw checked against the float64 oracle (forward, dX, dW, da1).  The
number of draws is small in the suite; GATMI_SWEEP=<n> runs more."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch
from tests.util import assert_close, layer_params, random_graph

pytestmark = pytest.mark.gpu
DRAWS = int(os.environ.get("GATMI_SWEEP", "10"))
HEAD_SHAPES = [(8, 8), (1, 7), (4, 8), (2, 16), (1, 64), (8, 4), (1, 4), (16, 8), (4, 64), (2, 3), (8, 32), (1, 100)]


def _oracle_params(p):
    return {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}


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
    res = bool(rng.random() < 0.3) and reduce == "concat"
    dst, src = random_graph(n, deg, seed=seed, hubs=hubs, hub_deg=min(n, 400))
    p = layer_params(F, H, D, seed=seed + 1, res=res)
    x = rng.normal(0, 1, (n, F))
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    what = "seed %d: n=%d F=%d H=%d D=%d deg=%.1f hubs=%d chunk=%d %s/%s res=%s" % (seed, n, F, H, D, deg, hubs, chunk, activation, reduce, res)
    out = ops.sparse_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"],
                               CSRGraph.from_coo(dst, src, n, dev, chunk=chunk), ops.make_cfg(H, D, activation, reduce),
                               tp.get("W_res"), tp.get("b_res"))
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    rp = _oracle_params(p)
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = ref_torch.layer_csr(rx, rp, torch.tensor(rowptr), torch.tensor(col), activation, reduce, residual=res)
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what=what + " forward")
    dout = rng.normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    ref.backward(torch.tensor(dout))
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what=what + " dX")
    for k in ("W", "a1", "bias") + (("W_res",) if res else ()):
        assert_close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), atol_scale=5e-6, what=what + " d" + k)


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
    hd = H * ops.padded_width(D)
    paths = ["dense", "lists"] if 32 <= hd <= 64 else (["lists"] if hd > 64 else ["dense"])
    for path in paths:
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
        what = "seed %d: B=%d N=%d F=%d H=%d D=%d density=%.2f soft=%s %s %s" % (seed, B, N, F, H, D, density, soft, activation, path)
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"],
                                  torch.tensor(bm, dtype=torch.float32, device=dev), ops.make_cfg(H, D, activation, mask_path=path))
        rp = _oracle_params(p)
        rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        ref = ref_torch.layer_dense(rx, rp, torch.tensor(bm), activation, "concat")
        assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what=what + " forward")
        dout = rng.normal(0, 1, tuple(out.shape))
        out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
        ref.backward(torch.tensor(dout))
        assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what=what + " dX")
        for k in ("W", "a1", "bias"):
            assert_close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), atol_scale=5e-6, what=what + " d" + k)
