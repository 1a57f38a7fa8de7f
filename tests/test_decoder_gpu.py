"""GPU parity of the inner-product decoder (utils/layers.py:307-313) against the float64 oracle:
forward and the gradient of the embeddings, with and without input dropout (mask replayed on the
host).  Tolerance: 1e-5 relative plus 2e-6 (5e-6 for gradients) of the tensor's scale."""
import numpy as np
import pytest
import torch

from oracle import ref_torch
from tests.util import assert_close, feat_keep_host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,F,activation,dropout", [
    (3, 246, 64, "sigmoid", 0.0), (2, 90, 32, "sigmoid", 0.0), (2, 264, 64, "sigmoid", 0.3), (5, 17, 7, "identity", 0.0),
    (1, 300, 128, "sigmoid", 0.0), (2, 130, 100, "identity", 0.5), (4, 1, 3, "sigmoid", 0.0), (2, 64, 16, "sigmoid", 0.0)])
def test_decoder_forward_and_gradient(B, N, F, activation, dropout):
    from gat_amd import ops
    from gat_amd.layers import InnerProductDecoder
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(N + F)
    z = rng.normal(0, 0.5, (B, N, F))
    zt = torch.tensor(z, dtype=torch.float32, device=dev, requires_grad=True)
    seed = 1234
    if dropout:
        out = ops.inner_product_decoder(zt, activation, dropout, seed=seed).reshape(-1)
        mask = feat_keep_host(seed, np.arange(B * N * F, dtype=np.uint64), 1.0 - dropout).reshape(B, N, F)
        assert abs((1.0 - mask.mean()) - dropout) < 0.05
        mask = torch.tensor(mask.astype(np.float64))
    else:
        out = InnerProductDecoder(zt, activation)
        mask = None
    assert out.shape == (B * N * N,)
    rz = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    ref = ref_torch.inner_product_decoder(rz, activation, 1.0 - dropout, mask)
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what="decoder forward")
    dout = rng.normal(0, 1, (B * N * N,))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    ref.backward(torch.tensor(dout))
    assert_close(zt.grad.cpu().numpy(), rz.grad.numpy(), atol_scale=5e-6, what="decoder dZ")


def test_decoder_rejects_wide_embeddings():
    from gat_amd import ops
    with pytest.raises(ValueError):
        ops.inner_product_decoder(torch.zeros(1, 4, 129, device="cuda:0"))


def test_decoder_replays_from_a_hip_graph():
    """The whole-graph kernels raise their dynamic-LDS limit before launching (more than 48 KB for
    264 nodes x 64 features); that call, and the launches, must be capturable."""
    from gat_amd.layers import InnerProductDecoder
    dev = torch.device("cuda:0")
    B, N, F = 3, 264, 64
    z = torch.randn(B, N, F, device=dev, requires_grad=True)
    d = torch.randn(B * N * N, device=dev)

    def step():
        z.grad = None
        out = InnerProductDecoder(z)
        out.backward(d)
        return out
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            eager = step().detach().clone()
    torch.cuda.current_stream().wait_stream(side)
    eager_grad = z.grad.detach().clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        captured = step()
    z.grad.zero_()
    captured.detach().zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured.detach(), eager) and torch.equal(z.grad, eager_grad)
