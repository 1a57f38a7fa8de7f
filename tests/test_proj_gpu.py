"""GPU parity of the projection kernels alone (utils/layers.py:11,14,15 and their gradients):
Wh = X.W, f1 = Wh.a1 + b1, f2 = Wh.a2 + b2, dW = X^T.dWh, dX = dWh.W^T, da/db — at row counts
and widths that reach every kernel variant (resident-weight tiles, chunked tiles, aligned and
unaligned rows), against float64 matmuls."""
import numpy as np
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,F,H,D", [(5000, 100, 8, 8), (4100, 36, 1, 7), (6000, 128, 4, 16), (4500, 8, 2, 4),
                                     (3000, 100, 8, 8), (2708, 1433, 8, 8), (700, 130, 1, 64), (5000, 52, 8, 8),
                                     (1500, 33, 16, 8), (900, 1433, 4, 64), (1200, 64, 8, 32), (800, 7, 2, 64)])
def test_projection_forward_and_backward(n, F, H, D):
    from gat_amd import _lib, ops
    from gat_amd._lib import ptr
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n + F)
    cfg = ops.make_cfg(H, D)
    dp, hd = cfg.dp, H * cfg.dp
    x = rng.normal(0, 1, (n, F))
    W = rng.normal(0, 0.3, (F, H, D))
    a1, a2 = rng.normal(0, 1, (H, D)), rng.normal(0, 1, (H, D))
    b1, b2 = rng.normal(0, 1, H), rng.normal(0, 1, H)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    xt = t(x)
    Wp, a1p, a2p, _, Wh, f1, f2, _ = ops._project(xt, t(W), t(a1), t(a2), t(b1), t(b2), cfg)
    wh_ref = (x @ W.reshape(F, H * D)).reshape(n, H, D)
    got = Wh.reshape(n, H, dp).cpu().numpy()
    assert_close(got[..., :D], wh_ref, what="Wh")
    assert (got[..., D:] == 0).all()
    assert_close(f1.cpu().numpy(), (wh_ref * a1).sum(-1) + b1, what="f1")
    assert_close(f2.cpu().numpy(), (wh_ref * a2).sum(-1) + b2, what="f2")

    dWh = rng.normal(0, 1, (n, H, dp))
    dWh[..., D:] = 0
    df1, df2 = rng.normal(0, 1, (n, H)), rng.normal(0, 1, (n, H))
    dx, dW, da1, da2, db1, db2, _ = ops._project_backward(xt, Wp, None, Wh, t(dWh.reshape(n, hd)), t(df1), t(df2), None, cfg, True)
    d = dWh[..., :D].reshape(n, H * D)
    assert_close(dW.cpu().numpy().reshape(F, H * D), x.T @ d, atol_scale=5e-6, what="dW")
    assert_close(dx.cpu().numpy(), d @ W.reshape(F, H * D).T, atol_scale=5e-6, what="dX")
    assert_close(da1.cpu().numpy(), np.einsum("nh,nhd->hd", df1, wh_ref), atol_scale=5e-6, what="da1")
    assert_close(da2.cpu().numpy(), np.einsum("nh,nhd->hd", df2, wh_ref), atol_scale=5e-6, what="da2")
    tol = lambda terms: 1e-5 * np.abs(terms.sum(0)) + 1e-6 * np.abs(terms).sum(0)
    assert (np.abs(db1.cpu().numpy() - df1.sum(0)) <= tol(df1)).all()
    assert (np.abs(db2.cpu().numpy() - df2.sum(0)) <= tol(df2)).all()


def test_parameter_gradients_are_bitwise_reproducible():
    from gat_amd import ops
    dev = torch.device("cuda:0")
    n, F, H, D = 20000, 100, 8, 8
    cfg = ops.make_cfg(H, D)
    g = torch.Generator(device=dev).manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    x, W, a1, a2, b1, b2 = r(n, F), r(F, H, D), r(H, D), r(H, D), r(H), r(H)
    Wp, _, _, _, Wh, _, _, _ = ops._project(x, W, a1, a2, b1, b2, cfg)
    dWh, df1, df2 = r(n, H * D), r(n, H), r(n, H)
    a = ops._project_backward(x, Wp, None, Wh, dWh, df1, df2, None, cfg, False)
    b = ops._project_backward(x, Wp, None, Wh, dWh, df1, df2, None, cfg, False)
    for u, v in zip(a[1:6], b[1:6]):
        assert torch.equal(u, v)
