"""Edge cases and size-independent properties of the CSR layer on the GPU: rows and columns without
entries, graphs without edges, zero-size calls through the C ABI, node relabelling, the logit
biases not cancelling (utils/layers.py:14-17: LeakyReLU sits between the bias and the softmax),
and the error codes of the C ABI for arguments the kernels are not built for."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy, ref_torch
from tests.util import assert_close, layer_params, random_graph

pytestmark = pytest.mark.gpu


def _layer(graph, x, p, dev, activation="elu", reduce="concat", grad=True):
    from gat_amd import ops
    H, D = p["a1"].shape
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=grad) for k, v in p.items()}
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=grad)
    out = ops.sparse_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph,
                               ops.make_cfg(H, D, activation, reduce))
    return out, xt, tp


def test_rows_and_columns_without_entries():
    """A destination without any stored entry (possible only when the caller bypasses
    preprocess_adj_bias, which adds self-loops) aggregates nothing: out = act(bias); a source that
    nobody reads gets exactly zero gradient from the attention part."""
    from gat_amd import CSRGraph
    dev = torch.device("cuda:0")
    n, F, H, D = 60, 9, 8, 8
    dst, src = random_graph(n, 4.0, seed=8)
    keep = (dst != 7) & (dst != 59) & (src != 13)          # rows 7, 59 empty; column 13 empty
    dst, src = dst[keep], src[keep]
    p = layer_params(F, H, D, seed=3)
    x = np.random.default_rng(1).normal(0, 1, (n, F))
    graph = CSRGraph.from_coo(dst, src, n, dev)
    out, xt, tp = _layer(graph, x, p, dev)
    o = out.detach().cpu().numpy()
    assert_close(o[[7, 59]], np.broadcast_to(ref_numpy.elu(p["bias"].reshape(-1)), (2, H * D)), what="empty rows")
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    rows = np.setdiff1d(np.arange(n), [7, 59])
    ref = ref_numpy.gat_layer_csr(x, p, rowptr, col, activation=ref_numpy.elu, reduce="concat")
    assert_close(o[rows], ref[rows], what="other rows")
    dout = np.random.default_rng(2).normal(0, 1, o.shape)
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    assert torch.isfinite(xt.grad).all() and all(torch.isfinite(t.grad).all() for t in tp.values())
    # node 13 is read by nobody; as a destination it still has its own row, so compare with the oracle
    rp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    # rows 7 and 59 have no entries: their oracle softmax is over an empty set, leave them out of the loss
    mask = np.ones((n, 1)); mask[[7, 59]] = 0
    ref_t = ref_torch.layer_csr(rx, rp, torch.tensor(rowptr), torch.tensor(col), "elu", "concat")
    (torch.nan_to_num(ref_t) * torch.tensor(dout * mask)).sum().backward()
    got = xt.grad.cpu().numpy()
    # gradient through rows 7 / 59 is zero on the device as well (constant output)
    assert_close(got, rx.grad.numpy(), atol_scale=5e-6, what="dX with empty rows")


def test_graph_without_edges_and_zero_sizes():
    from gat_amd import CSRGraph, _lib
    from gat_amd._lib import ptr
    dev = torch.device("cuda:0")
    n, F, H, D = 10, 5, 2, 4
    graph = CSRGraph.from_coo(np.zeros(0, np.int32), np.zeros(0, np.int32), n, dev)
    assert graph.nnz == 0
    p = layer_params(F, H, D, seed=0)
    out, xt, tp = _layer(graph, np.ones((n, F)), p, dev, "identity")
    assert_close(out.detach().cpu().numpy(), np.broadcast_to(p["bias"].reshape(-1), (n, H * D)), what="no edges")
    out.sum().backward()
    assert torch.equal(xt.grad, torch.zeros_like(xt.grad))
    # zero rows through the C ABI: accepted, nothing launched
    st = _lib.current_stream()
    _lib.call("gatmi_proj_fwd", None, 0, 5, 5, None, None, None, None, None, 2, 4, None, None, 0, None, None, None, None, None, 1.0, 0, st)
    _lib.call("gatmi_gat_fwd", 0, 2, 4, None, 0, None, 0, None, None, 0, 8, None, None, None, None, None, None, 1, 0,
              0.2, 1.0, 1.0, 0, None, None, None, None, None, None, st)
    _lib.call("gatmi_dense_attn_fwd", 0, 16, 8, 8, None, None, None, None, None, None, 1, 0.2, 1.0, 1.0, 0,
              None, None, None, None, None, st)
    _lib.call("gatmi_ipd_fwd", 0, 16, 8, None, 2, 1.0, 0, None, st)
    _lib.call("gatmi_halo_pack", None, None, 0, 64, None, st)


def test_relabelling_nodes_permutes_the_output():
    from gat_amd import CSRGraph
    dev = torch.device("cuda:0")
    n, F, H, D = 150, 12, 8, 8
    dst, src = random_graph(n, 5.0, seed=4, hubs=1, hub_deg=100)
    p = layer_params(F, H, D, seed=5)
    x = np.random.default_rng(6).normal(0, 1, (n, F))
    perm = np.random.default_rng(7).permutation(n)          # new id of node v is perm[v]
    a, _, _ = _layer(CSRGraph.from_coo(dst, src, n, dev, chunk=32), x, p, dev, grad=False)
    xp = np.empty_like(x); xp[perm] = x
    b, _, _ = _layer(CSRGraph.from_coo(perm[dst].astype(np.int32), perm[src].astype(np.int32), n, dev, chunk=32), xp, p, dev, grad=False)
    assert_close(b.cpu().numpy()[perm], a.cpu().numpy(), rtol=2e-6, atol_scale=1e-6, what="relabelled")


def test_logit_biases_do_not_cancel():
    """b1 is added before LeakyReLU, so shifting it changes the coefficients (a shift after the
    LeakyReLU would cancel in the softmax); the kernel follows the oracle in both cases."""
    from gat_amd import CSRGraph
    dev = torch.device("cuda:0")
    n, F, H, D = 80, 7, 4, 8
    dst, src = random_graph(n, 6.0, seed=9)
    p = layer_params(F, H, D, seed=1)
    x = np.random.default_rng(2).normal(0, 1, (n, F))
    g = CSRGraph.from_coo(dst, src, n, dev)
    base, _, _ = _layer(g, x, p, dev, grad=False)
    p2 = dict(p); p2["b1"] = p["b1"] + 1.5
    shifted, _, _ = _layer(g, x, p2, dev, grad=False)
    assert (shifted - base).abs().max() > 1e-3
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    assert_close(shifted.cpu().numpy(), ref_numpy.gat_layer_csr(x, p2, rowptr, col, activation=ref_numpy.elu, reduce="concat"), what="shifted b1")


def test_c_abi_rejects_what_it_is_not_built_for():
    from gat_amd import _lib
    st = _lib.current_stream()
    t = torch.zeros(64, device="cuda:0")
    cases = [
        ("gatmi_gat_fwd", (4, 3, 8, None, 0, None, 0, None, None, 0, 24, None, None, None, None, None, None, 1, 0, 0.2, 1.0, 1.0, 0,
                           None, None, None, None, None, None, st), "power of two"),                       # 3 heads
        ("gatmi_ipd_fwd", (1, 4, 129, t.data_ptr(), 2, 1.0, 0, t.data_ptr(), st), "128"),                  # too wide
        ("gatmi_ipd_fwd", (1, 4, 8, t.data_ptr(), 7, 1.0, 0, t.data_ptr(), st), "activation"),
        ("gatmi_dense_attn_fwd_masked", (1, 4, 1, 8, t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None, None, 1, 0.2,
                                         1.0, 1.0, 0, t.data_ptr(), None, None, None, None, None, st), "narrow"),
        ("gatmi_proj_fwd", (t.data_ptr() + 4, 4, 4, 4, t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 2, 4,
                            t.data_ptr() + 4, None, 0, t.data_ptr(), t.data_ptr(), None, None, None, 1.0, 0, st), "aligned"),
    ]
    for name, args, needle in cases:
        with pytest.raises(_lib.GatmiError) as e:
            _lib.call(name, *args)
        assert needle in str(e.value), (name, str(e.value))
