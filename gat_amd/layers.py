"""Drop-in surface of the reference's utils/layers.py for the graph-attention head.

    attn_head(seq, out_sz, bias_mat, activation, in_drop=0.0, coef_drop=0.0, residual=False)
        — utils/layers.py:6, dense [B,N,F] input with an additive [B,N,N] mask
    sp_attn_head(seq, out_sz, adj_mat, activation, nb_nodes, in_drop=0.0, coef_drop=0.0, residual=False)
        — called at models/sp_gat.py:13-16 (its body is absent from the reference tree); batch 1

TensorFlow 1 creates a layer's variables implicitly, in call order, every time the graph is built
(tf.layers.conv1d at layers.py:11,14,15,30; bias_add at :25).  Here the variables live in a
VariableStore that hands them out in the same call order: the first pass creates them (Glorot
uniform kernels, zero biases — the TF defaults), later passes reuse them.  `attn_heads` /
`sp_attn_heads` are the fused form of n_heads consecutive calls followed by tf.concat or
tf.add_n / n (models/sp_gat.py:12-17, 27-32): one kernel sweep for all heads.
"""
import contextlib
import math
import weakref

import torch
import torch.nn as nn

from . import ops
from .features import SparseFeatures
from .graph import CSRGraph, MaskLists

elu = "elu"
identity = "identity"


def _activation_name(activation):
    """Accept the reference's callables (tf.nn.elu, lambda x: x) by name, or our string names."""
    if activation in ("elu", "relu", "identity", None):
        return activation or "identity"
    name = getattr(activation, "__name__", "")
    if name in ("elu", "relu"):
        return name
    if name == "<lambda>" or name == "identity":
        return "identity"
    raise ValueError("unsupported activation %r: the kernels fuse ELU, ReLU or the identity" % (activation,))


class _MaskGraphs:
    """Memo of block-diagonal graphs for dense masks that are passed again and again.

    The reference feeds the same bias_mat every step.  When a mask tensor is seen a second time
    unchanged (same tensor object, same version counter), is hard and sparse, it is converted once
    with CSRGraph.from_bias_mat and the layer runs the CSR kernels from then on (about twice as fast
    as re-reading the N x N floats at a density of 10 %).  The memo keys on the tensor OBJECT through
    a weak reference, so a different tensor that happens to reuse the address is never mistaken for
    it, and any in-place change (version counter) drops the entry.  `enabled = False` switches it off."""

    def __init__(self):
        self.enabled = True
        self._entries = {}          # id(tensor) -> [weakref, version, graph | None (not decided) | False (stay dense)]

    def _drop(self, key, ref):
        """The mask tensor died: release its graph (unless the id was taken by a newer entry)."""
        e = self._entries.get(key)
        if e is not None and e[0] is ref:
            del self._entries[key]

    def lookup(self, t, heads=None):
        """heads = (H, padded width) of the asking layer: when its rows are the 64 floats of the graph-resident
        kernels and the batch is one of many small graphs, the memo holds prebuilt lists (MaskLists: built in a few
        launches, results bit-identical to the mask-reading kernels) instead of a block-diagonal CSR graph."""
        if not self.enabled or not torch.is_tensor(t) or t.dim() != 3 or not t.is_cuda:
            return None
        e = self._entries.get(id(t))
        if e is None or e[0]() is not t or e[1] != t._version:
            if len(self._entries) > 16:
                self._entries = {k: v for k, v in self._entries.items() if v[0]() is not None}
            key = id(t)
            e = self._entries[key] = [weakref.ref(t, lambda _r, k=key: self._drop(k, _r)), t._version, None]
            # A batch of small graphs whose size is not a multiple of 4 (the reference's 246- and 90-node atlases) has no
            # mask-reading graph-resident kernels to fall back on: building the lists costs less than ONE step on the
            # row-block kernels saves (4096 x 246: 1.6 ms once, then 1.2 instead of 3.2 ms per step), so decide at once.
            B, N = t.shape[:2]
            if not (ops.prebuilt_lists_batch(B, N) and not ops.graph_resident_batch(B, N) and t.dtype == torch.float32):
                return None
        if e[2] is None:             # second sighting (or the first, see above): decide
            if torch.cuda.is_current_stream_capturing():
                return None
            f = ops._mask_facts(t)
            e[2] = False
            B, N = t.shape[:2]
            small = ops.prebuilt_lists_batch(B, N) and t.dtype == torch.float32
            # many small graphs: prebuilt lists beat the matrix-core kernels at (nearly) every density of a hard mask
            # (4096 x 264, a full mask: 4.7 ms on prebuilt lists, 5.3 ms on the matrix cores)
            if f is not None and not (small and f[1]) and (small or f[0] <= ops.MASK_LIST_DENSITY):
                try:
                    e[2] = MaskLists(t) if small else CSRGraph.from_bias_mat(t)
                except ValueError:   # biases differ inside a row: the dense kernels add them
                    pass
        g = e[2] or None
        if isinstance(g, MaskLists) and not (heads and heads[0] * heads[1] == 64 and heads[1] >= 8 and g.fits(*heads)):
            # this layer's rows are not those of the graph-resident kernels: it needs the CSR form of the same mask
            if len(e) < 4:
                try:
                    e.append(CSRGraph.from_bias_mat(t) if not torch.cuda.is_current_stream_capturing() else None)
                except ValueError:
                    e.append(None)
            return e[3]
        return g


mask_graphs = _MaskGraphs()


class GATHeads(nn.Module):
    """Parameters of H attention heads of width D over F input features (one `attn_head` call
    per head in the reference).  W [F,H,D] <- conv1d(seq, out_sz, 1, use_bias=False) kernels;
    a1/a2 [H,D], b1/b2 [H] <- the two conv1d(seq_fts, 1, 1); bias [H,D] <- bias_add;
    W_res/b_res <- the residual conv1d (layers.py:30), present only when residual and F != D."""

    def __init__(self, f_in, heads, width, residual=False, device=None, storage="f32"):
        super().__init__()
        self.f_in, self.heads, self.width = int(f_in), int(heads), int(width)
        self.storage = storage      # "bf16": gathered tables of the CSR path in bf16 (2e-2 class)
        kw = {"device": device, "dtype": torch.float32}
        self.W = nn.Parameter(torch.empty(f_in, heads, width, **kw))
        self.a1 = nn.Parameter(torch.empty(heads, width, **kw))
        self.a2 = nn.Parameter(torch.empty(heads, width, **kw))
        self.b1 = nn.Parameter(torch.zeros(heads, **kw))
        self.b2 = nn.Parameter(torch.zeros(heads, **kw))
        self.bias = nn.Parameter(torch.zeros(heads, width, **kw))
        self.has_res = bool(residual) and f_in != width      # layers.py:29: no-op when widths agree
        if self.has_res:
            self.W_res = nn.Parameter(torch.empty(f_in, heads, width, **kw))
            self.b_res = nn.Parameter(torch.zeros(heads, width, **kw))
        self.reset_parameters()

    def reset_parameters(self):
        lim_w = math.sqrt(6.0 / (self.f_in + self.width))
        lim_a = math.sqrt(6.0 / (self.width + 1))
        with torch.no_grad():
            self.W.uniform_(-lim_w, lim_w)
            self.a1.uniform_(-lim_a, lim_a)
            self.a2.uniform_(-lim_a, lim_a)
            if self.has_res:
                self.W_res.uniform_(-lim_w, lim_w)

    def load(self, params):
        """Copy explicit weights (dict of array-likes with the parameter names)."""
        with torch.no_grad():
            for k, v in params.items():
                getattr(self, k).copy_(torch.as_tensor(v, dtype=torch.float32))

    def forward(self, seq, graph_or_bias, activation="elu", reduce="concat", in_drop=0.0, coef_drop=0.0,
                training=None, seed=None):
        training = self.training if training is None else training
        H, D = self.heads, self.width
        act = _activation_name(activation)
        if isinstance(graph_or_bias, CSRGraph) and seq.dim() == 3:
            # the sparse path of the reference has batch 1; a batch of small graphs may come with ONE
            # block-diagonal graph over all B*N nodes (CSRGraph.from_bias_mat)
            B, N = seq.shape[:2]
            if B * N != graph_or_bias.n_dst:
                raise ValueError("seq is [%d,%d,.] but the graph has %d nodes" % (B, N, graph_or_bias.n_dst))
            out = self.forward(seq.reshape(B * N, -1), graph_or_bias, act, reduce, in_drop, coef_drop, training, seed)
            return out.reshape(B, N, -1)
        if isinstance(seq, SparseFeatures):                                  # sparse node features stand for [1,N,F]
            if not isinstance(graph_or_bias, CSRGraph):
                raise ValueError("sparse node features go with the sparse attention head (sp_attn_head)")
            if training and in_drop > 0.0:
                seq = seq.dropout(in_drop)
            groups = _head_groups(H, D)
            if len(groups) != 1:
                raise ValueError("sparse node features: %d heads of width %d need more than one sweep" % (H, D))
            if seed is None and training and (in_drop > 0.0 or coef_drop > 0.0):
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            return self._run(seq, graph_or_bias, slice(0, H), act, reduce, in_drop, coef_drop, training, seed)[None]
        groups = _head_groups(H, D)
        g = mask_graphs.lookup(graph_or_bias, (groups[0][1], ops.padded_width(D)) if len(groups) == 1 else None)   # a dense mask met before, unchanged
        if isinstance(g, MaskLists) and seq.dim() == 3:
            graph_or_bias = g                                                # same surface, the mask's lists instead of its floats
        elif g is not None and seq.dim() == 3:
            return self.forward(seq, g, act, reduce, in_drop, coef_drop, training, seed)
        x_drop = 0.0
        if training and in_drop > 0.0:                                       # layers.py:8-9
            if ops.input_dropout_fusable(seq.reshape(-1, seq.shape[-1])):
                x_drop = in_drop                                             # the projection kernels mask the tiles of X they load
            else:
                seq = torch.nn.functional.dropout(seq, in_drop, True)
        if seed is None and training and (in_drop > 0.0 or coef_drop > 0.0):
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        groups = _head_groups(H, D)
        if len(groups) == 1:
            return self._run(seq, graph_or_bias, slice(0, H), act, reduce, in_drop, coef_drop, training, seed, x_drop)
        # head counts the fused kernels do not take in one sweep (not a power of two, or too wide):
        # power-of-two groups of heads, concatenated (and averaged afterwards if asked to)
        if act != "identity" and reduce == "mean":
            raise ValueError("mean over heads is defined for the identity activation (models/sp_gat.py:30-32)")
        outs = [self._run(seq, graph_or_bias, slice(h0, h0 + hn), act, "concat", in_drop, coef_drop, training,
                          None if seed is None else seed + h0, x_drop) for h0, hn in groups]
        out = torch.cat(outs, dim=-1)
        if reduce == "mean":
            out = out.reshape(*out.shape[:-1], H, D).mean(-2)
        return out

    def _run(self, seq, graph_or_bias, hs, act, reduce, in_drop, coef_drop, training, seed, x_drop=0.0):
        """One fused sweep over the heads in slice hs."""
        hn, D = hs.stop - hs.start, self.width
        sl = lambda t: t[hs] if hn != self.heads else t
        W = self.W[:, hs] if hn != self.heads else self.W
        wr = (self.W_res[:, hs] if hn != self.heads else self.W_res) if self.has_res else None
        br = sl(self.b_res) if self.has_res else None
        if isinstance(graph_or_bias, CSRGraph):
            cfg = ops.make_cfg(hn, D, act, reduce, coef_drop, in_drop, training, seed, storage=self.storage, x_drop=x_drop)
            return ops.sparse_gat_layer(seq, W, sl(self.a1), sl(self.a2), sl(self.b1), sl(self.b2), sl(self.bias),
                                        graph_or_bias, cfg, wr, br)
        cfg = ops.make_cfg(hn, D, act, "concat", coef_drop, in_drop, training, seed, x_drop=x_drop)
        out = ops.dense_gat_layer(seq, W, sl(self.a1), sl(self.a2), sl(self.b1), sl(self.b2), sl(self.bias),
                                  graph_or_bias, cfg, wr, br)
        if reduce == "mean":
            B, N = out.shape[:2]
            out = out.reshape(B, N, hn, D).mean(2)
        return out


def _head_groups(H, D):
    """Split H heads into runs (start, count) whose count is a power of two with
    count * padded_width(D) <= 256, largest first: 6 heads of width 8 -> (0,4), (4,2)."""
    dp = ops.padded_width(D)
    if dp > 256:
        raise ValueError("head width %d is beyond the kernels' 256-column rows" % D)
    cap = 1
    while cap * 2 * dp <= 256:
        cap *= 2
    groups, h0 = [], 0
    while h0 < H:
        g = cap
        while g > H - h0:
            g //= 2
        groups.append((h0, g))
        h0 += g
    return groups


# ------------------------------------------------------------------------------------------------
# TF1-style implicit variables
# ------------------------------------------------------------------------------------------------
class VariableStore(nn.Module):
    """Hands out GATHeads modules in call order, creating them on the first pass."""

    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList()
        self._cursor = 0

    def begin(self):
        self._cursor = 0

    def next_heads(self, f_in, heads, width, residual, device):
        if self._cursor == len(self.layers):
            self.layers.append(GATHeads(f_in, heads, width, residual, device))
        m = self.layers[self._cursor]
        if (m.f_in, m.heads, m.width) != (f_in, heads, width):
            raise ValueError("variable %d was created for (f_in,heads,width)=%s, now asked for %s"
                             % (self._cursor, (m.f_in, m.heads, m.width), (f_in, heads, width)))
        self._cursor += 1
        return m


_default_store = VariableStore()
_active = [_default_store]


def default_store():
    return _active[-1]


@contextlib.contextmanager
def variable_scope(store):
    """Route the implicit variables of attn_head / sp_attn_head calls to `store`."""
    _active.append(store)
    try:
        yield store
    finally:
        _active.pop()


def _graph_of(adj_mat, device):
    if isinstance(adj_mat, CSRGraph):
        return adj_mat
    return CSRGraph.from_reference_triple(adj_mat, device)


def attn_heads(seq, out_sz, bias_mat, activation, n_heads, reduce="concat", in_drop=0.0, coef_drop=0.0,
               residual=False, training=True):
    m = default_store().next_heads(seq.shape[-1], n_heads, out_sz, residual, seq.device)
    return m(seq, bias_mat, activation, reduce, in_drop, coef_drop, training)


def sp_attn_heads(seq, out_sz, adj_mat, activation, nb_nodes, n_heads, reduce="concat", in_drop=0.0,
                  coef_drop=0.0, residual=False, training=True):
    graph = _graph_of(adj_mat, seq.device)
    if graph.n_dst != nb_nodes:
        raise ValueError("nb_nodes=%d but adj_mat has %d rows" % (nb_nodes, graph.n_dst))
    m = default_store().next_heads(seq.shape[-1], n_heads, out_sz, residual, seq.device)
    return m(seq, graph, activation, reduce, in_drop, coef_drop, training)


def attn_head(seq, out_sz, bias_mat, activation, in_drop=0.0, coef_drop=0.0, residual=False):
    """utils/layers.py:6.  seq [B,N,F], bias_mat [B,N,N] additive mask; returns [B,N,out_sz].
    As in the reference, a non-zero dropout rate means "training" (layers.py:8,19,21 test the rate
    itself; evaluation feeds 0.0, execute_cora.py:123)."""
    training = in_drop != 0.0 or coef_drop != 0.0
    return attn_heads(seq, out_sz, bias_mat, activation, 1, "concat", in_drop, coef_drop, residual, training)


def sp_attn_head(seq, out_sz, adj_mat, activation, nb_nodes, in_drop=0.0, coef_drop=0.0, residual=False):
    """Signature of the call at models/sp_gat.py:13-16.  seq [1,N,F]; adj_mat is the
    (indices, values, shape) triple of preprocess_adj_bias (utils/process.py:188) or a CSRGraph."""
    training = in_drop != 0.0 or coef_drop != 0.0
    return sp_attn_heads(seq, out_sz, adj_mat, activation, nb_nodes, 1, "concat", in_drop, coef_drop, residual, training)


class BNFAttention(nn.Module):
    """The attention step of the brain-network models: one `attn_v2e_BNF_V2` coefficient matrix per
    head (utils/layers.py:229-237: conv1d to feat_dim without bias, two conv1d(.,1,1) logit halves,
    LeakyReLU, unmasked softmax) combined with the weighted adjacency as in the loop body of
    `attn_e2v_BNF_V2` (:174-179):  agg = ((net_mat>0)*m2 + net_mat) * (m1 + coefs) @ seq.
    Returns [B, N, heads, F]; the surrounding conv1d / concat layers of the reference are ordinary
    dense layers and stay in the host framework."""

    def __init__(self, f_in, heads, feat_dim=64, device=None):
        super().__init__()
        self.f_in, self.heads, self.feat_dim = int(f_in), int(heads), int(feat_dim)
        kw = {"device": device, "dtype": torch.float32}
        self.W = nn.Parameter(torch.empty(f_in, heads, feat_dim, **kw))
        self.a1 = nn.Parameter(torch.empty(heads, feat_dim, **kw))
        self.a2 = nn.Parameter(torch.empty(heads, feat_dim, **kw))
        self.b1 = nn.Parameter(torch.zeros(heads, **kw))
        self.b2 = nn.Parameter(torch.zeros(heads, **kw))
        self.m1 = nn.Parameter(torch.ones(heads, **kw))      # stimulate_mag_1, layers.py:175
        self.m2 = nn.Parameter(torch.ones(heads, **kw))      # stimulate_mag_2, layers.py:177
        lim_w = math.sqrt(6.0 / (f_in + feat_dim))
        lim_a = math.sqrt(6.0 / (feat_dim + 1))
        with torch.no_grad():
            self.W.uniform_(-lim_w, lim_w)
            self.a1.uniform_(-lim_a, lim_a)
            self.a2.uniform_(-lim_a, lim_a)

    def forward(self, seq, net_mat):
        return ops.bnf_aggregate(seq, net_mat, self.W, self.a1, self.a2, self.b1, self.b2, self.m1, self.m2)


def InnerProductDecoder(seq, activation="sigmoid", dropout=0.0):
    """utils/layers.py:307-313.  seq [B,N,F] -> flat [B*N*N] vector of activation(<z_i, z_j>) per
    graph, as the reference returns it (callers reshape to [-1,N,N], models/gat.py:206).
    activation: "sigmoid" (the reference's default, tf.nn.sigmoid) or "identity"."""
    return ops.inner_product_decoder(seq, activation, dropout).reshape(-1)


def _linear(layer, x, activation="identity"):
    """A width-1 convolution of the reference (tf.layers.conv1d(x, out, 1, activation)) held as torch.nn.Linear, or the
    identity."""
    if isinstance(layer, nn.Identity):
        return x
    return ops.tall_linear(x, layer.weight, layer.bias, activation)


class AttnE2VBNFV2(nn.Module):
    """`attn_e2v_BNF_V2` (utils/layers.py:158-227): `reps` (5 in the reference, :171) rounds of
    n_heads brain-network attention steps (BNFAttention: the libgatmi kernels), each head followed
    by conv1d(agg, out_sz, 1, activation) (:180), the heads concatenated and merged by another
    conv1d (:182-183), with the reference's residual rule (:185-191: a projection when widths
    differ, a plain sum otherwise); the five round outputs are concatenated (:221), passed through two
    conv1d layers (:223-224) and l2-normalised over the node axis (:225).  Returns (out [B,N,out_sz],
    aggall [B,N,reps*out_sz]).  The width-1 convolutions are ordinary dense layers (torch)."""

    def __init__(self, f_in, out_sz, n_heads, activation="relu", residual=False, reps=5, feat_dim=64, device=None):
        super().__init__()
        if activation not in ("relu", "elu", "identity"):
            raise ValueError("activation must be 'relu', 'elu' or 'identity'")
        self.activation, self.residual, self.reps, self.n_heads, self.out_sz = activation, bool(residual), reps, n_heads, out_sz
        kw = {"device": device, "dtype": torch.float32}
        self.att, self.head_w, self.head_b, self.merge, self.res = nn.ModuleList(), nn.ParameterList(), nn.ParameterList(), nn.ModuleList(), nn.ModuleList()
        for rep in range(reps):
            f = f_in if rep == 0 else out_sz
            self.att.append(BNFAttention(f, n_heads, feat_dim, device=device))
            w = torch.empty(n_heads, f, out_sz, **kw)
            lim = math.sqrt(6.0 / (f + out_sz))
            self.head_w.append(nn.Parameter(w.uniform_(-lim, lim)))
            self.head_b.append(nn.Parameter(torch.zeros(n_heads, out_sz, **kw)))
            self.merge.append(nn.Linear(n_heads * out_sz, out_sz, **kw))
            self.res.append(nn.Linear(f, out_sz, **kw) if (self.residual and f != out_sz) else nn.Identity())
        self.out1 = nn.Linear(reps * out_sz, out_sz, **kw)
        self.out2 = nn.Linear(out_sz, out_sz, **kw)

    def forward(self, seq, net_mat):
        seq_out, outs = seq, []
        for rep in range(self.reps):
            agg = self.att[rep](seq_out, net_mat)                                        # [B,N,H,F]   layers.py:174-179
            heads = torch.einsum("bnhf,hfo->bnho", agg, self.head_w[rep])
            heads = ops.add_row_bias(heads.flatten(2), self.head_b[rep].reshape(-1), self.activation)         # :180
            merged = _linear(self.merge[rep], heads, self.activation)                    # :182-183
            if self.residual:
                seq_out = merged + _linear(self.res[rep], seq_out)                      # :185-189
            else:
                seq_out = merged
            outs.append(seq_out)
        aggall = torch.cat(outs, dim=-1)
        out = _linear(self.out2, _linear(self.out1, aggall, self.activation), self.activation)
        out = out / torch.sqrt(torch.clamp((out * out).sum(dim=-2, keepdim=True), min=1e-12))   # tf.nn.l2_normalize(axis=-2)
        return out, aggall
