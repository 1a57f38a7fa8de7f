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
                                         (1, 264, 64, 3, 64)])
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
