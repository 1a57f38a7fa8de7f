"""PyTorch-CPU restatement of the reference attention head with autograd (TEST INFRASTRUCTURE).

Checker only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Same math as oracle/ref_numpy.py (utils/layers.py:6-34 of the reference), written with
differentiable torch ops so that gradients of every input and parameter are available in
float64 (the gradient oracle) and float32 (the timed CPU baseline).  PARITY UNPINNED for the
floating-point path: TensorFlow is not installed and the reference has no golden vectors.
"""
import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.2  # tf.nn.leaky_relu default, utils/layers.py:17


def _act(name):
    if name == "elu":
        return F.elu
    if name == "relu":
        return F.relu
    if name in ("identity", "none", None):
        return lambda x: x
    raise ValueError(name)


def layer_csr(x, p, rowptr, col, activation="elu", reduce="concat", residual=False,
              coef_mask=None, ft_mask=None, coef_keep=1.0, ft_keep=1.0):
    """All H heads of one layer over a CSR graph (dst-major rows, col = source ids).

    x [N,F]; p: W [F,H,D], a1,a2 [H,D], b1,b2 [H], bias [H,D] (+W_res [F,H,D], b_res [H,D]).
    coef_mask [E,H] and ft_mask [N,H,D] are explicit 0/1 dropout masks (layers.py:20,22).
    Returns [N,H*D] (concat, sp_gat.py:17) or [N,D] (mean over heads, sp_gat.py:32)."""
    n = x.shape[0]
    H, D = p["a1"].shape
    E = col.shape[0]
    dst = torch.repeat_interleave(torch.arange(n, dtype=torch.long), (rowptr[1:] - rowptr[:-1]).long())
    src = col.long()
    wh = (x @ p["W"].reshape(x.shape[1], H * D)).reshape(n, H, D)          # layers.py:11
    f1 = (wh * p["a1"]).sum(-1) + p["b1"]                                   # layers.py:14
    f2 = (wh * p["a2"]).sum(-1) + p["b2"]                                   # layers.py:15
    e = F.leaky_relu(f1[dst] + f2[src], LEAKY_SLOPE)                        # layers.py:16-17
    emax = torch.full((n, H), float("-inf"), dtype=x.dtype).scatter_reduce(
        0, dst[:, None].expand(E, H), e.detach(), reduce="amax", include_self=True)
    emax = torch.where(torch.isinf(emax), torch.zeros_like(emax), emax)
    pe = torch.exp(e - emax[dst])
    den = torch.zeros((n, H), dtype=x.dtype).index_add(0, dst, pe)
    coefs = pe / den[dst]                                                   # softmax, layers.py:17
    if coef_mask is not None:
        coefs = coefs * coef_mask / coef_keep                               # layers.py:20
    src_fts = wh
    if ft_mask is not None:
        src_fts = wh * ft_mask / ft_keep                                    # layers.py:22
    vals = torch.zeros((n, H, D), dtype=x.dtype).index_add(0, dst, coefs[:, :, None] * src_fts[src])
    ret = vals + p["bias"]                                                  # layers.py:24-25
    if residual and x.shape[1] != D:                                        # layers.py:28-30
        ret = ret + (x @ p["W_res"].reshape(x.shape[1], H * D)).reshape(n, H, D) + p["b_res"]
    out = _act(activation)(ret)                                             # layers.py:34
    if reduce == "concat":
        return out.reshape(n, H * D)
    return out.mean(1)


def layer_dense(x, p, bias_mat, activation="elu", reduce="concat", residual=False,
                coef_mask=None, ft_mask=None, coef_keep=1.0, ft_keep=1.0):
    """Dense twin on [B,N,F] with additive [B,N,N] mask (layers.py:6-34, gat.py:11-15,25-29).
    coef_mask [B,H,N,N] and ft_mask [B,N,H,D] are explicit 0/1 dropout masks (layers.py:20,22)."""
    B, n, Fin = x.shape
    H, D = p["a1"].shape
    wh = (x @ p["W"].reshape(Fin, H * D)).reshape(B, n, H, D)
    f1 = (wh * p["a1"]).sum(-1) + p["b1"]                    # [B,N,H]
    f2 = (wh * p["a2"]).sum(-1) + p["b2"]
    logits = f1.permute(0, 2, 1)[:, :, :, None] + f2.permute(0, 2, 1)[:, :, None, :]   # [B,H,N(i),N(j)]
    coefs = torch.softmax(F.leaky_relu(logits, LEAKY_SLOPE) + bias_mat[:, None], dim=-1)
    if coef_mask is not None:
        coefs = coefs * coef_mask / coef_keep
    src_fts = wh if ft_mask is None else wh * ft_mask / ft_keep
    vals = torch.einsum("bhij,bjhd->bihd", coefs, src_fts)
    ret = vals + p["bias"]
    if residual and Fin != D:
        ret = ret + (x @ p["W_res"].reshape(Fin, H * D)).reshape(B, n, H, D) + p["b_res"]
    out = _act(activation)(ret)
    if reduce == "concat":
        return out.reshape(B, n, H * D)
    return out.mean(2)


def inference_csr(x, layers, rowptr, col, activation="elu", residual=False):
    """models/sp_gat.py:7-34 composition."""
    h = layer_csr(x, layers[0], rowptr, col, activation, "concat", False)
    for lp in layers[1:-1]:
        h = layer_csr(h, lp, rowptr, col, activation, "concat", residual)
    return layer_csr(h, layers[-1], rowptr, col, "identity", "mean", False)


def inference_dense(x, layers, bias_mat, activation="elu", residual=False):
    """models/gat.py:7-31 composition."""
    h = layer_dense(x, layers[0], bias_mat, activation, "concat", False)
    for lp in layers[1:-1]:
        h = layer_dense(h, lp, bias_mat, activation, "concat", residual)
    return layer_dense(h, layers[-1], bias_mat, "identity", "mean", False)


def bnf_v2e_aggregate(x, net_mat, p):
    """Inner step of attn_e2v_BNF_V2 (utils/layers.py:171-179) with the coefficients of
    attn_v2e_BNF_V2 (:229-237), for H "heads" that each own a projection, written with
    differentiable torch ops.

    x [B,N,F]; net_mat [B,N,N] weighted adjacency; p: W [F,H,Dz], a1,a2 [H,Dz], b1,b2 [H],
    m1,m2 [H] (stimulate_mag_1 / _2, initialised to one in the reference).
      seq  = conv1d(x, Dz, 1, use_bias=False)                      :231
      f_1, f_2 = conv1d(seq, 1, 1) twice                           :232-233
      coefs = softmax(leaky_relu(f_1 + f_2^T))      (UNmasked)     :234-237
      att  = ones * m1 + coefs                                     :176
      net1 = (net_mat > 0) * m2 + net_mat                          :168,178
      agg  = (net1 * att) @ x                                      :179
    Returns agg [B,N,H,F]."""
    B, n, Fin = x.shape
    _, H, Dz = p["W"].shape                 # a1 / a2 may also come as one leaf per node, [B,N,H,Dz]
    z = (x @ p["W"].reshape(Fin, H * Dz)).reshape(B, n, H, Dz)
    f1 = (z * p["a1"]).sum(-1) + p["b1"]                       # [B,N,H]
    f2 = (z * p["a2"]).sum(-1) + p["b2"]
    logits = f1.permute(0, 2, 1)[:, :, :, None] + f2.permute(0, 2, 1)[:, :, None, :]   # [B,H,i,j]
    coefs = torch.softmax(F.leaky_relu(logits, LEAKY_SLOPE), dim=-1)
    att = p["m1"][None, :, None, None] + coefs
    thr = (net_mat > 0).to(x.dtype)
    net1 = thr[:, None] * p["m2"][None, :, None, None] + net_mat[:, None]
    return torch.einsum("bhij,bjf->bihf", net1 * att, x)


def inner_product_decoder(z, activation="sigmoid", keep=1.0, mask=None):
    """InnerProductDecoder (utils/layers.py:307-313): inputs = dropout(seq, 1 - dropout) (:308),
    x = inputs @ inputs^T per graph (:309-310), flattened (:311), activation (:312).
    mask: the 0/1 keep mask of the dropout, same shape as z (tf.nn.dropout scales kept entries by
    1/keep); None = no dropout.  Returns the flat [B*N*N] vector."""
    inputs = z if mask is None else z * mask / keep
    x = inputs @ inputs.transpose(1, 2)
    x = x.reshape(-1)
    return torch.sigmoid(x) if activation == "sigmoid" else x


def attn_e2v_bnf_v2(x, net_mat, rounds, out1, out2, activation="relu", residual=False):
    """attn_e2v_BNF_V2 (utils/layers.py:158-227) with explicit weights.  rounds: one dict per
    repetition (:171) with the attention parameters of bnf_v2e_aggregate plus head_w [H,F,out],
    head_b [H,out] (:180), merge_w [H*out,out], merge_b (:183), and res_w / res_b when the residual
    needs a projection (:187).  out1 / out2: (weight [in,out], bias) of the closing conv1d layers
    (:223-224).  Returns (out, aggall)."""
    act = {"relu": torch.relu, "elu": F.elu, "identity": lambda t: t}[activation]
    seq_out, outs = x, []
    for r in rounds:
        agg = bnf_v2e_aggregate(seq_out, net_mat, r)                                   # [B,N,H,F]
        heads = act(torch.einsum("bnhf,hfo->bnho", agg, r["head_w"]) + r["head_b"])
        merged = act(heads.flatten(2) @ r["merge_w"] + r["merge_b"])
        if residual:
            seq_out = merged + (seq_out @ r["res_w"] + r["res_b"] if "res_w" in r else seq_out)
        else:
            seq_out = merged
        outs.append(seq_out)
    aggall = torch.cat(outs, dim=-1)
    out = act(act(aggall @ out1[0] + out1[1]) @ out2[0] + out2[1])
    out = out / torch.sqrt(torch.clamp((out * out).sum(dim=-2, keepdim=True), min=1e-12))   # l2_normalize(axis=-2), :225
    return out, aggall
