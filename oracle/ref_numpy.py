"""CPU restatement of the reference graph-attention math and preprocessing (TEST INFRASTRUCTURE).

This module is the *checker*, not the product: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  Nothing under gat_amd/ imports it.

Parity status
-------------
* Integer / mask preprocessing (`adj_to_bias`, `preprocess_adj_bias`, `preprocess_features`) is
  PINNED: tests/golden/preproc_*.npz were produced by importing the reference's own
  utils/process.py (see oracle/make_fixtures.py) and this restatement reproduces them bit-exactly.
* Floating-point attention math is PARITY UNPINNED: the reference computes it with TensorFlow 1.x
  (README.md:28-35), which is not installed here, and the reference has no tests or golden vectors.
  What follows restates utils/layers.py:6-34 op for op in numpy (any float dtype), and the
  composition in models/gat.py:7-31 / models/sp_gat.py:7-34.

Every function cites the reference lines it follows (paths relative to the reference root).
"""
import numpy as np

LEAKY_SLOPE = 0.2  # tf.nn.leaky_relu default alpha, used at utils/layers.py:17


# --------------------------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------------------------
def elu(x):
    """tf.nn.elu: x if x > 0 else exp(x) - 1 (default activation, models/sp_gat.py:9)."""
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def identity(x):
    """`lambda x: x` used for the output heads (models/sp_gat.py:30)."""
    return x


def relu(x):
    """tf.nn.relu, the activation the brain-network models pass (models/gat.py:191)."""
    return np.maximum(x, 0)


def leaky_relu(x, slope=LEAKY_SLOPE):
    return np.where(x > 0, x, slope * x)


def softmax_lastaxis(x):
    """tf.nn.softmax over the last axis (utils/layers.py:17)."""
    m = np.max(x, axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=-1, keepdims=True)


# --------------------------------------------------------------------------------------------
# dense head: utils/layers.py:6-34
# --------------------------------------------------------------------------------------------
def attn_head_dense(seq, params, bias_mat, activation=elu, residual=False,
                    in_mask=None, coef_mask=None, ft_mask=None, in_keep=1.0, coef_keep=1.0,
                    return_coefs=False):
    """One attention head on dense [B,N,F] input with an additive [B,N,N] mask.

    params: dict with
        W      [F, D]   kernel of conv1d(seq, out_sz, 1, use_bias=False)      (layers.py:11)
        a1     [D]      kernel of conv1d(seq_fts, 1, 1) #1, b1 scalar bias    (layers.py:14)
        a2     [D]      kernel of conv1d(seq_fts, 1, 1) #2, b2 scalar bias    (layers.py:15)
        bias   [D]      tf.contrib.layers.bias_add                            (layers.py:25)
        W_res  [F, D], b_res [D]   residual conv1d with bias                  (layers.py:30)
    Dropout is expressed with explicit 0/1 masks (TF's RNG stream is not a parity target):
        in_mask  [B,N,F]  on seq            (layers.py:9),   scaled by 1/in_keep
        coef_mask[B,N,N]  on coefs          (layers.py:20),  scaled by 1/coef_keep
        ft_mask  [B,N,D]  on seq_fts        (layers.py:22),  scaled by 1/in_keep
    """
    dt = seq.dtype
    W = params["W"].astype(dt)
    a1 = params["a1"].astype(dt)
    a2 = params["a2"].astype(dt)
    b1 = np.asarray(params.get("b1", 0.0), dtype=dt)
    b2 = np.asarray(params.get("b2", 0.0), dtype=dt)
    bias = np.asarray(params.get("bias", 0.0), dtype=dt)

    if in_mask is not None:                                   # layers.py:8-9
        seq = seq * in_mask.astype(dt) / dt.type(in_keep)
    seq_fts = seq @ W                                         # layers.py:11
    f_1 = seq_fts @ a1[:, None] + b1                          # layers.py:14  [B,N,1]
    f_2 = seq_fts @ a2[:, None] + b2                          # layers.py:15  [B,N,1]
    logits = f_1 + np.transpose(f_2, (0, 2, 1))               # layers.py:16  [B,N,N], row i = dst
    coefs = softmax_lastaxis(leaky_relu(logits).astype(dt) + bias_mat.astype(dt))  # layers.py:17
    coefs_nodrop = coefs
    if coef_mask is not None:                                 # layers.py:19-20
        coefs = coefs * coef_mask.astype(dt) / dt.type(coef_keep)
    if ft_mask is not None:                                   # layers.py:21-22
        seq_fts = seq_fts * ft_mask.astype(dt) / dt.type(in_keep)
    vals = coefs @ seq_fts                                    # layers.py:24
    ret = vals + bias                                         # layers.py:25
    if residual:                                              # layers.py:28-32
        if seq.shape[-1] != ret.shape[-1]:
            ret = ret + (seq @ params["W_res"].astype(dt) + params["b_res"].astype(dt))
        else:
            pass  # layers.py:32 assigns to the dead name seq_fts: a no-op on the result
    out = activation(ret)                                     # layers.py:34
    if return_coefs:
        return out, coefs_nodrop, logits
    return out


# --------------------------------------------------------------------------------------------
# sparse head: signature from models/sp_gat.py:13-16; body = layers.py:6-34 on stored entries
# --------------------------------------------------------------------------------------------
def coo_to_csr(indices, n):
    """indices [E,2] as emitted by preprocess_adj_bias (process.py:186): column 0 is the
    SparseTensor row (= destination i), column 1 the SparseTensor column (= source j).
    Returns (rowptr int64 [n+1], col int32 [E]) sorted by (dst, src), duplicates kept."""
    indices = np.asarray(indices)
    dst = indices[:, 0].astype(np.int64)
    src = indices[:, 1].astype(np.int64)
    order = np.lexsort((src, dst))
    dst, src = dst[order], src[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, src.astype(np.int32)


def sp_attn_head_csr(seq, params, rowptr, col, activation=elu, residual=False,
                     coef_mask=None, ft_mask=None, in_mask=None, in_keep=1.0, coef_keep=1.0,
                     return_coefs=False):
    """Same head as attn_head_dense restricted to stored (dst,src) entries; seq is [N,F]
    (the sparse path is batch-1 only, README.md:20-22).  coef_mask is per stored edge [E]."""
    dt = seq.dtype
    n = seq.shape[0]
    W = params["W"].astype(dt)
    a1 = params["a1"].astype(dt)
    a2 = params["a2"].astype(dt)
    b1 = np.asarray(params.get("b1", 0.0), dtype=dt)
    b2 = np.asarray(params.get("b2", 0.0), dtype=dt)
    bias = np.asarray(params.get("bias", 0.0), dtype=dt)
    if in_mask is not None:
        seq = seq * in_mask.astype(dt) / dt.type(in_keep)
    seq_fts = seq @ W
    f_1 = seq_fts @ a1 + b1
    f_2 = seq_fts @ a2 + b2
    agg_src = seq_fts
    if ft_mask is not None:
        agg_src = seq_fts * ft_mask.astype(dt) / dt.type(in_keep)
    vals = np.zeros((n, W.shape[1]), dtype=dt)
    coefs = np.zeros(len(col), dtype=dt)
    for i in range(n):
        lo, hi = int(rowptr[i]), int(rowptr[i + 1])
        if hi == lo:
            continue
        js = col[lo:hi]
        e = leaky_relu(f_1[i] + f_2[js]).astype(dt)
        e = e - e.max()
        p = np.exp(e)
        p = p / p.sum()
        coefs[lo:hi] = p
        if coef_mask is not None:
            p = p * coef_mask[lo:hi].astype(dt) / dt.type(coef_keep)
        vals[i] = p @ agg_src[js]
    ret = vals + bias
    if residual and seq.shape[-1] != ret.shape[-1]:
        ret = ret + (seq @ params["W_res"].astype(dt) + params["b_res"].astype(dt))
    out = activation(ret)
    if return_coefs:
        return out, coefs
    return out


def gat_layer_csr(seq, layer, rowptr, col, activation=elu, reduce="concat", residual=False):
    """All H heads of one layer: concat on the feature axis (models/sp_gat.py:17,26) or mean
    over heads (models/sp_gat.py:32).  layer: W [F,H,D], a1,a2 [H,D], b1,b2 [H], bias [H,D]
    (+ W_res [F,H,D], b_res [H,D])."""
    H = layer["a1"].shape[0]
    outs = []
    for h in range(H):
        p = {"W": layer["W"][:, h, :], "a1": layer["a1"][h], "a2": layer["a2"][h],
             "b1": layer["b1"][h], "b2": layer["b2"][h], "bias": layer["bias"][h]}
        if "W_res" in layer:
            p["W_res"] = layer["W_res"][:, h, :]
            p["b_res"] = layer["b_res"][h]
        outs.append(sp_attn_head_csr(seq, p, rowptr, col, activation=activation, residual=residual))
    if reduce == "concat":
        return np.concatenate(outs, axis=-1)
    return sum(outs) / seq.dtype.type(H)


def gat_layer_dense(seq, layer, bias_mat, activation=elu, reduce="concat", residual=False):
    """Dense twin of gat_layer_csr on [B,N,F] (models/gat.py:11-15, 25-29)."""
    H = layer["a1"].shape[0]
    outs = []
    for h in range(H):
        p = {"W": layer["W"][:, h, :], "a1": layer["a1"][h], "a2": layer["a2"][h],
             "b1": layer["b1"][h], "b2": layer["b2"][h], "bias": layer["bias"][h]}
        if "W_res" in layer:
            p["W_res"] = layer["W_res"][:, h, :]
            p["b_res"] = layer["b_res"][h]
        outs.append(attn_head_dense(seq, p, bias_mat, activation=activation, residual=residual))
    if reduce == "concat":
        return np.concatenate(outs, axis=-1)
    return sum(outs) / seq.dtype.type(H)


def gat_inference(inputs, layers, graph, activation=elu, residual=False):
    """models/gat.py:7-31 / models/sp_gat.py:7-34: hidden layers concat their heads, the last
    layer averages its heads with identity activation and residual=False.  The first layer
    always has residual=False (gat.py:14).  graph is ("dense", bias_mat) or ("csr", rowptr, col)."""
    def run(x, layer, act, red, res):
        if graph[0] == "dense":
            return gat_layer_dense(x, layer, graph[1], act, red, res)
        return gat_layer_csr(x, layer, graph[1], graph[2], act, red, res)
    h = run(inputs, layers[0], activation, "concat", False)
    for layer in layers[1:-1]:
        h = run(h, layer, activation, "concat", residual)
    return run(h, layers[-1], identity, "mean", False)


# --------------------------------------------------------------------------------------------
# preprocessing: utils/process.py
# --------------------------------------------------------------------------------------------
def adj_to_bias(adj, sizes, nhood=1):
    """utils/process.py:14-25.  adj [G,N,N]; returns float64 [G,N,N] = -1e9*(1-mt) where
    mt = (adj+I)^nhood with entries inside the sizes[g] x sizes[g] block clamped to 1 when > 0;
    entries outside that block keep their computed value (process.py:21-24 only loops the block)."""
    adj = np.asarray(adj)
    G, N = adj.shape[0], adj.shape[1]
    mt = np.empty(adj.shape)
    for g in range(G):
        m = np.eye(N)
        for _ in range(nhood):
            m = np.matmul(m, (np.asarray(adj[g]) + np.eye(N)))
        s = int(sizes[g])
        blk = m[:s, :s]
        blk[blk > 0.0] = 1.0
        mt[g] = m
    return -1e9 * (1.0 - mt)


def preprocess_adj_bias(adj):
    """utils/process.py:179-188 for a scipy sparse matrix: add self-loops, binarise entries > 0,
    COO, float32 values, indices emitted as (col,row)."""
    import scipy.sparse as sp
    n = adj.shape[0]
    adj = adj + sp.eye(n)
    adj[adj > 0.0] = 1.0
    if not sp.isspmatrix_coo(adj):
        adj = adj.tocoo()
    adj = adj.astype(np.float32)
    indices = np.vstack((adj.col, adj.row)).transpose()
    return indices, adj.data, adj.shape


def preprocess_features(features):
    """utils/process.py:155-162: D^-1 X with inf -> 0; returns the dense matrix only."""
    import scipy.sparse as sp
    rowsum = np.array(features.sum(1))
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    return np.asarray(sp.diags(r_inv).dot(features).todense())


# --------------------------------------------------------------------------------------------
# brain-network variant: utils/layers.py:171-179 with :229-237
# --------------------------------------------------------------------------------------------
def bnf_v2e_aggregate(x, net_mat, p):
    """Loop-level numpy restatement of the inner step of attn_e2v_BNF_V2 for one graph batch.
    x [B,N,F]; net_mat [B,N,N]; p: W [F,H,Dz], a1,a2 [H,Dz], b1,b2,m1,m2 [H].  Returns [B,N,H,F]."""
    B, n, Fin = x.shape
    H, Dz = p["a1"].shape
    out = np.zeros((B, n, H, Fin), dtype=x.dtype)
    for b in range(B):
        thr = (net_mat[b] > 0).astype(x.dtype)                              # layers.py:168
        for h in range(H):
            seq = x[b] @ p["W"][:, h, :]                                    # :231
            f_1 = seq @ p["a1"][h] + p["b1"][h]                             # :232
            f_2 = seq @ p["a2"][h] + p["b2"][h]                             # :233
            coefs = softmax_lastaxis(leaky_relu(f_1[:, None] + f_2[None, :]))   # :234-237
            att = np.ones((n, n), dtype=x.dtype) * p["m1"][h] + coefs       # :176
            net1 = thr * p["m2"][h] + net_mat[b]                            # :178
            out[b, :, h, :] = (net1 * att) @ x[b]                           # :179
    return out
