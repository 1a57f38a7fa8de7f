"""The two-layer Cora network at its real size (2708 nodes, 1433 features, 8 heads x 8 hidden, one
7-class output head; execute_cora.py:13-22, execute_cora_sparse.py:16-26) against the float64 oracle:

  * one whole training step -- logits, masked cross-entropy, masked accuracy, L2 penalty, every
    gradient, and the variables after ONE Adam update (models/base_gattn.py:75-88, 140-165);
  * every inference surface: SpGAT.inference on dense features and on the sparse feature tuple,
    GAT.inference on the dense [1, 2708, 2708] mask called three times with the same tensor (first
    the list kernels with the 7-wide output head widened to 32 floats, then the conversion of the
    mask into its graph and the CSR kernels), and the matrix-core kernels on the same mask.

Only this size reaches the split-K projection (22 row blocks would leave the chip idle), the widened
output head and the mask memo.  The graph and the features are synthetic (the Planetoid files are not
in the reference tree); the oracle is evaluated once per module."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import ref_numpy, ref_torch, ref_train
from tests.util import assert_close

pytestmark = pytest.mark.gpu

N, F, C = 2708, 1433, 7
KW = dict(hid_units=[8], n_heads=[8, 1], residual=False)
LR, L2 = 0.005, 0.0005          # execute_cora.py:15-16
NAMES = ("W", "a1", "a2", "b1", "b2", "bias")


@pytest.fixture(scope="module")
def cora():
    from gat_amd import CSRGraph, layers, preprocess
    from gat_amd.features import SparseFeatures
    from gat_amd.models import SpGAT
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    adj = sp.random(N, N, density=0.001, random_state=rng, format="csr")
    adj = ((adj + adj.T) > 0).astype(np.float64)
    feats, feats_tuple = preprocess.preprocess_features(sp.random(N, F, density=0.0127, random_state=rng, format="csr"))
    feats = np.asarray(feats, dtype=np.float64)
    triple = preprocess.preprocess_adj_bias(adj)
    labels = np.zeros((N, C))
    labels[np.arange(N), rng.randint(C, size=N)] = 1.0
    mask = np.zeros(N, dtype=bool)
    mask[rng.permutation(N)[:140]] = True                       # execute_cora.py: 140 training nodes
    graph = CSRGraph.from_reference_triple(triple, dev)
    x = torch.tensor(feats[None], dtype=torch.float32, device=dev)
    store = layers.VariableStore()
    with layers.variable_scope(store):
        SpGAT.inference(x, C, N, False, 0.0, 0.0, bias_mat=graph, activation=layers.elu, **KW)      # creates the variables
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for m in store.layers:
            for k in ("b1", "b2", "bias"):
                getattr(m, k).copy_(torch.randn(getattr(m, k).shape, generator=g) * 0.1)
    init = [{k: getattr(m, k).detach().cpu().numpy().astype(np.float64) for k in NAMES} for m in store.layers]

    # ---- the oracle, once: forward, loss, every gradient, one Adam update (float64)
    rowptr, col = ref_numpy.coo_to_csr(triple[0], N)
    leaves = [{k: torch.tensor(v, requires_grad=True) for k, v in lp.items()} for lp in init]
    node_terms = [{k: torch.zeros((N, lp[k].shape[0]), dtype=torch.float64, requires_grad=True) for k in ("b1", "b2")} for lp in init]
    run = [dict(lp, b1=lp["b1"][None] + nt["b1"], b2=lp["b2"][None] + nt["b2"]) for lp, nt in zip(leaves, node_terms)]
    rx = torch.tensor(feats)
    logits = ref_torch.inference_csr(rx, run, torch.tensor(rowptr), torch.tensor(col))
    y, mk = torch.tensor(labels), torch.tensor(mask)
    flat = [lp[k] for lp in leaves for k in NAMES]
    ce = ref_train.masked_softmax_cross_entropy(logits, y, mk)
    l2 = ref_train.l2_loss_sum(flat, L2)
    (ce + l2).backward()
    grads = [{k: lp[k].grad.numpy().copy() for k in NAMES} for lp in leaves]
    terms = [{k: nt[k].grad.abs().sum(0).numpy() for k in ("b1", "b2")} for nt in node_terms]      # sum of |per-node terms|
    after = [{k: lp[k].detach().clone() for k in NAMES} for lp in leaves]
    ref_train.adam_step([lp[k] for lp in after for k in NAMES], [torch.tensor(gr[k]) for gr in grads for k in NAMES], {}, LR)
    return dict(dev=dev, adj=adj, feats=feats, feats_tuple=feats_tuple, triple=triple, graph=graph, x=x, labels=labels, mask=mask,
                init=init, logits=logits.detach().numpy(), ce=float(ce.detach()), l2=float(l2.detach()),
                acc=float(ref_train.masked_accuracy(logits.detach(), y, mk)), grads=grads, terms=terms,
                after=[{k: v.numpy() for k, v in lp.items()} for lp in after], rowptr=rowptr, col=col,
                xfeat=SparseFeatures.from_tuple(feats_tuple, dev))


def _fresh_store(cora):
    from gat_amd import layers
    store = layers.VariableStore()
    for lp in cora["init"]:
        f_in, heads, width = lp["W"].shape
        store.next_heads(f_in, heads, width, False, cora["dev"]).load(lp)
    store.begin()
    return store


def test_one_training_step_matches_the_oracle(cora):
    from gat_amd import layers
    from gat_amd.models import SpGAT
    dev = cora["dev"]
    store = _fresh_store(cora)
    y = torch.tensor(cora["labels"], dtype=torch.float32, device=dev)
    mk = torch.tensor(cora["mask"], device=dev)
    opt, l2 = SpGAT.training(store.parameters(), lr=LR, l2_coef=L2)
    with layers.variable_scope(store):
        logits = SpGAT.inference(cora["x"], C, N, True, 0.0, 0.0, bias_mat=cora["graph"], activation=layers.elu, **KW)[0]
    ce = SpGAT.masked_softmax_cross_entropy(logits, y, mk)
    pen = l2()
    acc = SpGAT.masked_accuracy(logits, y, mk)
    opt.zero_grad(set_to_none=True)
    (ce + pen).backward()
    assert_close(logits.detach().cpu().numpy(), cora["logits"], rtol=2e-5, atol_scale=4e-6, what="logits")
    assert abs(float(ce.detach()) - cora["ce"]) <= 2e-6 * abs(cora["ce"]), (float(ce.detach()), cora["ce"])
    assert abs(float(pen.detach()) - cora["l2"]) <= 2e-6 * abs(cora["l2"]), (float(pen.detach()), cora["l2"])
    assert abs(float(acc) - cora["acc"]) <= 1e-6, (float(acc), cora["acc"])
    gtol = []
    for li, m in enumerate(store.layers):
        for k in NAMES:
            got, ref = getattr(m, k).grad.cpu().numpy().astype(np.float64), cora["grads"][li][k]
            if k in ("b1", "b2"):       # sums of per-node terms that cancel almost completely: relative to the terms
                tol = 1e-5 * np.abs(ref) + 1e-6 * cora["terms"][li][k]
            else:
                tol = 1e-5 * np.abs(ref) + 5e-6 * np.abs(ref).max()
            bad = np.abs(got - ref) > tol
            assert not bad.any(), "layer %d d%s: %d of %d out of tolerance, worst %.3g x tol" % (
                li, k, bad.sum(), bad.size, (np.abs(got - ref) / tol).max())
            gtol.append(tol)
    before = [getattr(m, k).detach().cpu().numpy().astype(np.float64) for m in store.layers for k in NAMES]
    opt.step()
    # after one update: var - lr g / (|g| + e'), e' = 1e-8 / sqrt(1 - beta2).  A gradient error dg moves the
    # update by at most lr e' dg / (|g| + e')^2 (and never by more than 2 lr), which bounds the tolerance
    e1 = 1e-8 / np.sqrt(1.0 - 0.999)
    i = 0
    for li, m in enumerate(store.layers):
        for k in NAMES:
            got, ref, g = getattr(m, k).detach().cpu().numpy().astype(np.float64), cora["after"][li][k], cora["grads"][li][k]
            assert np.abs(got - before[i]).max() > 0.1 * LR                              # the step was taken
            tol = 4e-7 * np.abs(ref) + 4e-6 * LR + np.minimum(2.0, e1 * gtol[i] / (np.abs(g) + e1) ** 2) * LR
            bad = np.abs(got - ref) > tol
            assert not bad.any(), "layer %d %s after Adam: %d of %d out of tolerance, worst %.3g x tol" % (
                li, k, bad.sum(), bad.size, (np.abs(got - ref) / tol).max())
            i += 1


def _check_dw(store, cora, what, rtol=1e-5, atol_scale=5e-6):
    for li, m in enumerate(store.layers):
        assert_close(m.W.grad.cpu().numpy(), cora["dW_ref"][li], rtol=rtol, atol_scale=atol_scale, what="%s dW layer %d" % (what, li))
        m.W.grad = None


def _reference_dw(cora):
    """dW of both layers for the loss sum(logits * weights), weights fixed: float64 oracle, once."""
    if "dW_ref" in cora:
        return
    rng = np.random.default_rng(9)
    cora["dout"] = rng.normal(0, 1, (N, C))
    leaves = [{k: torch.tensor(v, requires_grad=(k == "W")) for k, v in lp.items()} for lp in cora["init"]]
    out = ref_torch.inference_csr(torch.tensor(cora["feats"]), leaves, torch.tensor(cora["rowptr"]), torch.tensor(cora["col"]))
    (out * torch.tensor(cora["dout"])).sum().backward()
    cora["dW_ref"] = [lp["W"].grad.numpy() for lp in leaves]


def test_spgat_inference_on_dense_and_sparse_features(cora):
    from gat_amd import layers
    from gat_amd.models import SpGAT
    _reference_dw(cora)
    dout = torch.tensor(cora["dout"], dtype=torch.float32, device=cora["dev"])
    for name, inp in (("dense features", cora["x"]), ("sparse features", cora["xfeat"]), ("reference triple", cora["x"])):
        store = _fresh_store(cora)
        adj_arg = cora["triple"] if name == "reference triple" else cora["graph"]
        with layers.variable_scope(store):
            out = SpGAT.inference(inp, C, N, False, 0.0, 0.0, bias_mat=adj_arg, activation=layers.elu, **KW)
        assert tuple(out.shape) == (1, N, C)
        assert_close(out[0].detach().cpu().numpy(), cora["logits"], rtol=2e-5, atol_scale=4e-6, what="SpGAT.inference, " + name)
        (out[0] * dout).sum().backward()
        _check_dw(store, cora, name)


def test_gat_inference_on_the_dense_mask_through_every_route(cora):
    from gat_amd import layers, ops, preprocess
    from gat_amd.models import GAT
    _reference_dw(cora)
    dev = cora["dev"]
    dout = torch.tensor(cora["dout"], dtype=torch.float32, device=dev)
    bias = preprocess.adj_to_bias(np.asarray(cora["adj"].todense())[None], [N], nhood=1)
    bm = torch.tensor(bias, dtype=torch.float32, device=dev)
    assert int((bm > -1.0).sum()) == cora["graph"].nnz                      # the same graph as the sparse triple
    store = _fresh_store(cora)
    layers.mask_graphs._entries.clear()
    assert ops._widen_for_lists(ops.make_cfg(1, C, "identity"), bm).dp == 32    # the 7-wide head goes to the list kernels widened
    for call in range(3):
        # call 0 with the memo of repeated masks off: both layers take the list kernels; calls 1 and 2 with it
        # on: the mask is met, converted into its graph on the second sighting, and served by the CSR kernels
        layers.mask_graphs.enabled = call > 0
        try:
            with layers.variable_scope(store):
                out = GAT.inference(cora["x"], C, N, False, 0.0, 0.0, bias_mat=bm, activation=layers.elu, **KW)
        finally:
            layers.mask_graphs.enabled = True
        assert_close(out[0].detach().cpu().numpy(), cora["logits"], rtol=2e-5, atol_scale=4e-6, what="GAT.inference call %d" % call)
        (out[0] * dout).sum().backward()
        _check_dw(store, cora, "GAT.inference call %d" % call)
        if call == 0:
            assert id(bm) not in layers.mask_graphs._entries
    assert layers.mask_graphs.lookup(bm) is not None                        # converted: later calls run the CSR kernels
    # the matrix-core kernels on the same mask (hidden layer: 8 heads x 8; output layer: one 7-wide head)
    hidden = None
    for li, (lp, act) in enumerate(zip(cora["init"], ("elu", "identity"))):
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=(k == "W")) for k, v in lp.items()}
        H, D = lp["a1"].shape
        xin = cora["x"] if li == 0 else hidden.detach()
        o = ops.dense_gat_layer(xin, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], bm,
                                ops.make_cfg(H, D, act, mask_path="dense"))
        if li == 0:
            hidden = o
            ref_h = ref_numpy.gat_layer_csr(cora["feats"], lp, cora["rowptr"], cora["col"], ref_numpy.elu, "concat")
            assert_close(o[0].detach().cpu().numpy(), ref_h, what="matrix-core kernels, hidden layer")
        else:
            assert_close(o[0].detach().cpu().numpy(), cora["logits"], rtol=2e-5, atol_scale=4e-6, what="matrix-core kernels, output layer")
