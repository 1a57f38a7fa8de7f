"""Autograd operators over the C ABI of libgatmi: one fused graph-attention layer (all heads)
for CSR graphs and one for batched dense graphs.  PyTorch supplies device memory, streams and
the autograd tape; every floating-point operation of the layer runs in the HIP kernels.

Math and reference lines: utils/layers.py:6-34 (one head), models/sp_gat.py:12-17,27-32 (heads
concatenated or averaged).
"""
import torch

from . import _lib
from ._lib import ptr
from .features import SparseFeatures
from .graph import MaskLists

LEAKY_SLOPE = 0.2  # tf.nn.leaky_relu default, utils/layers.py:17

_ACT = {"elu": _lib.ACT_ELU, "relu": _lib.ACT_RELU, "identity": _lib.ACT_IDENTITY, None: _lib.ACT_IDENTITY}
_RED = {"concat": _lib.REDUCE_CONCAT, "mean": _lib.REDUCE_MEAN}


def padded_width(d):
    """Per-head width used by the kernels: next power of two >= max(d, 4)."""
    p = 4
    while p < d:
        p *= 2
    return p


def supported_shape(h, d):
    dp = padded_width(d)
    return h > 0 and (h & (h - 1)) == 0 and h * dp <= 256


def _pad_last(t, d, dp):
    if d == dp:
        return t.contiguous()
    return torch.nn.functional.pad(t, (0, dp - d)).contiguous()


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gat_amd: the graph-attention kernels run on the GPU only; got a %s tensor "
                               "(there is no CPU fallback)" % t.device)


def _f32(t):
    if t.dtype != torch.float32:
        raise TypeError("gat_amd: fp32 tensors expected, got %s" % t.dtype)
    return t


class _Cfg:
    __slots__ = ("H", "D", "dp", "activation", "reduce", "slope", "coef_keep", "feat_keep", "x_keep", "seed", "storage", "mask_path")


def _bf16_record_elems(H, dp):
    """bf16 elements per row of the forward gather records [hd bf16 | f2: H fp32 | pad]: whole 64-byte pieces."""
    return (2 * H * dp + 4 * H + 63) // 64 * 32


def _bf16_grad_record_elems(H, dp):
    """bf16 elements per row of the backward records [g: hd bf16 | f1 lse c: 3H fp32 | pad]: whole 128-byte lines."""
    return (2 * H * dp + 12 * H + 127) // 128 * 64


def _record_width(H, dp):
    """Floats per row of the [g | f1 lse c | pad] records the backward kernels exchange: whole 128-byte lines."""
    return (H * dp + 3 * H + 31) // 32 * 32


def _new(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def input_dropout_fusable(x2d):
    """Whether the projection kernels can apply the dropout of the layer's input (utils/layers.py:9) to the tiles
    of X they load (cfg.x_keep), instead of a separate pass that writes a dropped copy: dense rows that are 16-byte
    aligned and a multiple of four wide (and not wider than the whole-line weight-gradient kernel takes)."""
    return (torch.is_tensor(x2d) and x2d.dim() == 2 and x2d.shape[1] % 4 == 0 and x2d.shape[1] <= 1024 and x2d.stride(1) == 1
            and x2d.stride(0) % 4 == 0 and x2d.data_ptr() % 16 == 0)


def _project(x2d, W, a1, a2, b1, b2, cfg, W_res=None, b_res=None, Wh16_out=None):
    """K1: Wh = X.W (+ f1, f2) on [n, F] rows (dense tensor or SparseFeatures).  Returns padded
    parameter views and outputs.  Wh16_out: a list that receives the bf16 gather table of the forward kernel,
    one record [hd bf16 of Wh | the row's fp32 f2 | pad] per row, written by the projection's epilogue.
    cfg.x_keep < 1: the input is dropped inside the kernels (see input_dropout_fusable)."""
    n, F = x2d.shape
    sparse_x = isinstance(x2d, SparseFeatures)
    if sparse_x and W_res is not None:
        raise ValueError("sparse node features: the residual projection is not built (the reference's first "
                         "layer has residual=False, models/sp_gat.py:16)")
    H, D, dp = cfg.H, cfg.D, cfg.dp
    hd = H * dp
    Wp = _pad_last(W.reshape(F, H, D), D, dp).reshape(F, hd)
    a1p = _pad_last(a1.reshape(H, D), D, dp).reshape(hd)
    a2p = _pad_last(a2.reshape(H, D), D, dp).reshape(hd)
    b1c, b2c = b1.reshape(H).contiguous(), b2.reshape(H).contiguous()
    Wh = _new((n, hd), x2d)
    f1 = _new((n, H), x2d)
    f2 = _new((n, H), x2d)
    Wrp = brp = res = None
    if W_res is not None:
        Wrp = _pad_last(W_res.reshape(F, H, D), D, dp).reshape(F, hd)
        brp = _pad_last(b_res.reshape(H, D), D, dp).reshape(hd)
        res = _new((n, hd), x2d)
    Wh16, ld16 = None, 0
    if Wh16_out is not None:
        ld16 = _bf16_record_elems(H, dp)
        Wh16 = torch.empty((n, ld16), dtype=torch.bfloat16, device=Wh.device)
        Wh16_out.append(Wh16)
    if sparse_x:
        if cfg.x_keep < 1.0:
            raise ValueError("sparse node features: drop the stored values outside (SparseFeatures.dropout)")
        _lib.call("gatmi_proj_fwd_csr", ptr(x2d.rowptr), ptr(x2d.col), ptr(x2d.val), n, F, ptr(Wp), ptr(a1p), ptr(a2p),
                  ptr(b1c), ptr(b2c), H, dp, ptr(Wh), ptr(f1), ptr(f2), _lib.current_stream())
        if Wh16 is not None:
            _lib.call("gatmi_to_bf16_records", ptr(Wh), ptr(f2), n, H, dp, ld16, ptr(Wh16), _lib.current_stream())
        return Wp, a1p, a2p, None, Wh, f1, f2, None
    _lib.call("gatmi_proj_fwd", ptr(x2d), n, F, x2d.stride(0), ptr(Wp), ptr(a1p), ptr(a2p), ptr(b1c), ptr(b2c),
              H, dp, ptr(Wh), ptr(Wh16), ld16, ptr(f1), ptr(f2), ptr(Wrp), ptr(brp), ptr(res), cfg.x_keep, cfg.seed, _lib.current_stream())
    return Wp, a1p, a2p, Wrp, Wh, f1, f2, res


def _project_backward(x2d, Wp, Wrp, Wh, dWh, df1, df2, g, cfg, need_dx, attn_grads=None):
    """Backward of _project: dW, da1, da2, db1, db2 (+ dW_res, db_res, dX).  attn_grads: (da1, da2, db1, db2) when
    the attention kernel has already summed them (padded [hd], [hd], [H], [H]); Wh is then not read again."""
    n, F = x2d.shape
    H, D, dp = cfg.H, cfg.D, cfg.dp
    hd = H * dp
    st = _lib.current_stream()
    ws = _new((_lib.lib.gatmi_proj_bwd_ws_floats(F, H, dp),), x2d)
    dW = _new((F, hd), x2d)
    if attn_grads is not None:
        da1, da2, db1, db2 = attn_grads
        Wh = df1 = df2 = None
    else:
        da1, da2 = _new((hd,), x2d), _new((hd,), x2d)
        db1, db2 = _new((H,), x2d), _new((H,), x2d)

    def unpad(t, lead):
        return t.reshape(*lead, H, dp)[..., :D]
    if isinstance(x2d, SparseFeatures):
        _lib.call("gatmi_proj_bwd_params_csr", ptr(x2d.t_rowptr), ptr(x2d.t_row), ptr(x2d.t_val), n, F, ptr(dWh), ptr(Wh),
                  ptr(df1), ptr(df2), H, dp, ptr(dW), ptr(da1), ptr(da2), ptr(db1), ptr(db2), ptr(ws), st)
        return None, unpad(dW, (F,)), unpad(da1, ()), unpad(da2, ()), db1, db2, None
    xk, sd = cfg.x_keep, cfg.seed
    if attn_grads is not None:
        _lib.call("gatmi_proj_bwd_params", ptr(x2d), n, F, x2d.stride(0), ptr(dWh), None, None, None, H, dp,
                  ptr(dW), None, None, None, None, ptr(ws), xk, sd, st)
    else:
        _lib.call("gatmi_proj_bwd_params", ptr(x2d), n, F, x2d.stride(0), ptr(dWh), ptr(Wh), ptr(df1), ptr(df2), H, dp,
                  ptr(dW), ptr(da1), ptr(da2), ptr(db1), ptr(db2), ptr(ws), xk, sd, st)
    dWr = None
    if Wrp is not None:
        dWr = _new((F, hd), x2d)
        _lib.call("gatmi_proj_bwd_params", ptr(x2d), n, F, x2d.stride(0), ptr(g), None, None, None, H, dp,
                  ptr(dWr), None, None, None, None, ptr(ws), xk, sd, st)
    dx = None
    if need_dx:
        dx = _new((n, F), x2d)
        _lib.call("gatmi_proj_bwd_input", ptr(dWh), n, F, ptr(Wp), H, dp, ptr(dx), F, 0, xk, sd, st)
        if Wrp is not None:
            _lib.call("gatmi_proj_bwd_input", ptr(g), n, F, ptr(Wrp), H, dp, ptr(dx), F, 1, xk, sd, st)
    return (dx, unpad(dW, (F,)), unpad(da1, ()), unpad(da2, ()), db1, db2,
            None if dWr is None else unpad(dWr, (F,)))


class SparseGATLayerFn(torch.autograd.Function):
    """out = act( softmax-attention aggregate over the CSR graph + bias (+ residual) ), all heads."""

    @staticmethod
    def forward(ctx, x, W, a1, a2, b1, b2, bias, W_res, b_res, graph, cfg):
        _require_cuda(x, W)
        sparse_x = isinstance(x, SparseFeatures)
        if not sparse_x:
            x = _f32(x).contiguous()
        n, F = x.shape
        if n != graph.n_dst or graph.n_src != graph.n_dst:
            raise ValueError("x has %d rows but the graph is %d x %d" % (n, graph.n_dst, graph.n_src))
        H, D, dp = cfg.H, cfg.D, cfg.dp
        hd = H * dp
        st = _lib.current_stream()
        made16 = [] if cfg.storage == "bf16" else None
        Wp, a1p, a2p, Wrp, Wh, f1, f2, res = _project(x, W, a1, a2, b1, b2, cfg, W_res, b_res, made16)
        # fp32 rows: the kernels form the source half of the logits, <Wh_j, a2> + b2, from the rows they gather
        # (b2c given, no f2 gather); bf16 rows would add their rounding to the logits, so their records carry
        # the exact f2 behind the row (no f2 gather either; the per-source backward kernel reads the f2 table)
        bf16 = cfg.storage == "bf16"
        b2c = None if bf16 else _f32(b2).reshape(H).contiguous()
        if not bf16:
            f2 = None
        biasp = None if bias is None else _pad_last(bias.reshape(H, D), D, dp).reshape(hd)
        need_grad = any(ctx.needs_input_grad)
        out_w = dp if cfg.reduce == "mean" else hd
        out = _new((n, out_w), x)
        agg = pos = lse = q = None
        if need_grad:
            agg, pos = _new((n, hd), x), _new((n, hd), x)
            lse, q = _new((n, H), x), _new((n, H), x)
        ws = None
        if len(graph.split_csr):
            ws = _new((_lib.lib.gatmi_gat_fwd_ws_floats(graph.n_slots_csr, H, dp),), x)
        feat, fdt, ldf = Wh, _lib.DTYPE_F32, hd
        if bf16:
            feat, fdt, ldf = made16[0], _lib.DTYPE_BF16, made16[0].shape[1]
        _lib.call("gatmi_gat_fwd", n, H, dp, ptr(graph.items_csr), len(graph.items_csr),
                  ptr(graph.split_csr) if len(graph.split_csr) else None, len(graph.split_csr),
                  ptr(graph.col), ptr(feat), fdt, ldf, ptr(f1), None, None if bf16 else ptr(a2p), ptr(b2c), ptr(biasp), ptr(res),
                  _ACT[cfg.activation], _RED[cfg.reduce], cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed,
                  ptr(out), ptr(agg), ptr(pos), ptr(lse), ptr(q), ptr(ws), st)
        if need_grad:
            ctx.graph, ctx.cfg = graph, cfg
            ctx.has_bias, ctx.has_res = bias is not None, W_res is not None
            ctx.sparse_x = x if sparse_x else None
            ctx.save_for_backward(x.val if sparse_x else x, Wp, a1p, a2p, Wrp, Wh, f1, f2, b2c, out, agg, pos, lse, q, biasp)
        if cfg.reduce == "mean":
            # the averaged head in its padded rows [n, dp]: sparse_gat_layer slices the padding off OUTSIDE this
            # function, so that the result is a view whose base is this tensor and a fused loss can read the padded
            # rows and hand back a gradient of the same layout (no padding pass in either direction)
            return out
        return out.reshape(n, H, dp)[:, :, :D].reshape(n, H * D) if D != dp else out

    @staticmethod
    def backward(ctx, dout):
        x, Wp, a1p, a2p, Wrp, Wh, f1, f2, b2c, out, agg, pos, lse, q, biasp = ctx.saved_tensors
        graph, cfg = ctx.graph, ctx.cfg
        H, D, dp = cfg.H, cfg.D, cfg.dp
        hd = H * dp
        if ctx.sparse_x is not None:
            x = ctx.sparse_x
        n = x.shape[0]
        st = _lib.current_stream()
        if cfg.reduce == "mean":
            dout = dout.contiguous()                     # [n, dp]: the forward pass returned the padded rows
        else:
            dout = _pad_last(dout.reshape(n, H, D), D, dp).reshape(n, hd)
        bf16 = cfg.storage == "bf16"
        # one record per destination row, [g | f1 lse c | pad]: the per-source kernel's two gathers per entry
        # read one contiguous piece of memory (g in bf16 with storage="bf16")
        g = g16 = None
        ldg = ldg16 = hd
        if bf16:
            ldg16 = _bf16_grad_record_elems(H, dp)
            g16 = torch.empty((n, ldg16), dtype=torch.bfloat16, device=x.device)
            stats = g16.view(torch.float32)[:, hd // 2:hd // 2 + 3 * H]
            if ctx.has_res:
                g = _new((n, hd), x)                                      # the residual branch reads contiguous fp32 g
        elif ctx.has_res:
            g, stats = _new((n, hd), x), _new((n, 3 * H), x)
        else:
            rec = _new((n, _record_width(H, dp)), x)
            g, stats = rec[:, :hd], rec[:, hd:hd + 3 * H]
            ldg = rec.stride(0)
        lds = stats.stride(0)
        df1 = _new((n, H), x)
        dbias = _new((hd,), x)
        ws = _new((_lib.lib.gatmi_gat_bwd_node_ws_floats(H, dp),), x)
        # without a residual the pre-activation is agg + bias: the kernel then does not read `out`
        _lib.call("gatmi_gat_bwd_node", n, H, dp, ptr(dout), ptr(out) if ctx.has_res else None, ptr(agg), ptr(pos), ptr(q), ptr(f1),
                  ptr(lse), ptr(biasp), _ACT[cfg.activation], _RED[cfg.reduce], cfg.slope, ptr(g), ldg, ptr(g16), ldg16, ptr(stats), lds, ptr(df1),
                  ptr(dbias), ptr(ws), st)
        gtab, gdt = (g16, _lib.DTYPE_BF16) if bf16 else (g, _lib.DTYPE_F32)
        dWh = _new((n, hd), x)
        df2 = _new((n, H), x)
        ws2 = None
        if len(graph.split_csc):
            ws2 = _new((_lib.lib.gatmi_gat_bwd_ws_floats(graph.n_slots_csc, H, dp),), x)
        _lib.call("gatmi_gat_bwd_csc", n, H, dp, ptr(graph.items_csc), len(graph.items_csc),
                  ptr(graph.split_csc) if len(graph.split_csc) else None, len(graph.split_csc),
                  ptr(graph.rowidx), ptr(graph.perm), ptr(gtab), gdt, ldg16 if bf16 else ldg, ptr(stats), lds, ptr(Wh), ptr(f2), None if bf16 else ptr(a2p), ptr(b2c),
                  ptr(df1), ptr(a1p), ptr(a2p), cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed, ptr(dWh), ptr(df2), ptr(ws2), st)
        dx, dW, da1, da2, db1, db2, dWr = _project_backward(x, Wp, Wrp, Wh, dWh, df1, df2, g, cfg, ctx.needs_input_grad[0])
        db = dbias.reshape(H, dp)[:, :D] if ctx.has_bias else None
        dbr = dbias.reshape(H, dp)[:, :D] if ctx.has_res else None
        return dx, dW, da1, da2, db1, db2, db, dWr, dbr, None, None


MASK_LIST_DENSITY = 0.35      # the list kernels win below about a third of the entries contributing ...
# ... except for batches of many small graphs with rows of 64 floats (the graph-resident list kernels): at 4096 graphs
# of 264 nodes, 8 heads of 8, they take 2.4 / 3.0 / 3.7 / 5.3 ms at densities 0.2 / 0.35 / 0.5 / 1.0 against 5.3 ms of
# the matrix-core kernels at any density (hard masks; fp32 matrix instructions share the vector ALUs' multipliers, and at
# heads of 8 floats nine of a tile's sixteen columns do work).  Soft masks, whose walk fetches a bias per entry, keep
# the lower bound.
GRAPH_LIST_DENSITY = 0.9
_density_cache = {}


def graph_resident_batch(B, N):
    """Batch shapes the mask-reading graph-resident kernels take: at least 256 graphs of 4..320 nodes, a multiple of 4."""
    return B >= 256 and N % 4 == 0 and 4 <= N <= 320


def prebuilt_lists_batch(B, N):
    """Batch shapes that can run on prebuilt lists (MaskLists): the same without the multiple of 4, which only the 16-byte
    mask loads ask for — the reference's atlases have 246 and 90 nodes (Data_processing.py:46,518)."""
    return B >= 256 and 4 <= N <= 320


def _mask_facts(bias_mat):
    """(density, soft): the fraction of mask entries that can contribute (within 1e6 of their row's maximum) and
    whether some of them differ from that maximum (the bias then matters to the softmax), estimated on
    the first graphs of the batch and remembered per tensor version.  None while a HIP graph is being
    captured and nothing is remembered (the estimate needs a device-to-host copy).  The memo is keyed
    on address, shape and version counter; a different mask that reuses a freed tensor's address can
    inherit its estimate, which can only cost speed: both kernel families give the same results."""
    key = (bias_mat.data_ptr(), tuple(bias_mat.shape), bias_mat._version)
    d = _density_cache.get(key)
    if d is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        sample = bias_mat[:max(1, min(bias_mat.shape[0], (1 << 22) // max(1, bias_mat.shape[1] * bias_mat.shape[2])))]
        top = sample.amax(-1, keepdim=True)
        seen = sample > top - 1.0e6
        facts = torch.stack((seen.float().mean(), (seen & (sample != top)).any().float())).tolist()
        d = (float(facts[0]), facts[1] > 0)
        if len(_density_cache) > 64:
            _density_cache.clear()
        _density_cache[key] = d
    return d


def _mask_density(bias_mat):
    f = _mask_facts(bias_mat)
    return None if f is None else f[0]


def mask_list_density_limit(cfg, bias_mat):
    """Density up to which the list kernels are the faster family for this mask and layer shape."""
    f = _mask_facts(bias_mat)
    B, N = bias_mat.shape[:2]
    if f is not None and not f[1] and cfg.H * cfg.dp == 64 and cfg.dp >= 8 and graph_resident_batch(B, N):
        return GRAPH_LIST_DENSITY
    return MASK_LIST_DENSITY


# (heads, padded width) pairs the matrix-core kernels are built for (GATMI_DENSE_SHAPES in dense_attn.hip)
_MFMA_SHAPES = {(8, 8), (4, 8), (2, 8), (1, 8), (1, 4), (2, 4), (4, 4), (8, 4), (1, 16), (2, 16), (4, 16), (1, 32), (2, 32), (1, 64)}


def _use_mask_lists(cfg, bias_mat):
    """Pick the kernels for a [B,N,N] mask: compacted lists for sparse masks (and for rows wider than
    the matrix-core kernels are built for), matrix cores otherwise."""
    ok = cfg.H * cfg.dp >= 32 and bias_mat.shape[1] <= 32767
    if cfg.mask_path == "lists":
        if not ok:
            raise ValueError("mask_path='lists' needs heads * padded width >= 32 and N <= 32767")
        return True
    if cfg.mask_path == "dense" or not ok:
        return False
    if (cfg.H, cfg.dp) not in _MFMA_SHAPES:
        return True
    d = _mask_density(bias_mat)
    return d is not None and d <= mask_list_density_limit(cfg, bias_mat)


def _widen_for_lists(cfg, bias_mat):
    """Narrow rows (fewer than 32 floats: a single 7-class output head) cannot use the list kernels,
    and on one large graph the matrix-core kernels get few blocks.  For a very sparse mask it pays to
    zero-pad the head width so that a row has 32 floats and take the list kernels: returns such a
    copy of cfg, or cfg itself.  Only in mask_path="auto" (padding changes the keys of the in-kernel
    feature dropout, which the forced paths keep stable)."""
    N = bias_mat.shape[1]
    if cfg.mask_path != "auto" or cfg.H * cfg.dp >= 32 or not (512 <= N <= 32767):
        return cfg
    d = _mask_density(bias_mat)
    if d is None or d > 0.05:
        return cfg
    wide = _Cfg()
    for k in _Cfg.__slots__:
        setattr(wide, k, getattr(cfg, k))
    wide.dp = 32 // cfg.H
    return wide


class DenseGATLayerFn(torch.autograd.Function):
    """Same layer on [B, N, F] inputs with an additive [B, N, N] mask (utils/layers.py:6-34)."""

    @staticmethod
    def forward(ctx, x, W, a1, a2, b1, b2, bias, W_res, b_res, bias_mat, cfg):
        _require_cuda(x, W, None if isinstance(bias_mat, MaskLists) else bias_mat)
        if cfg.reduce != "concat":
            raise ValueError("dense layer: heads are concatenated by the kernel; average outside")
        x = _f32(x).contiguous()
        B, N, F = x.shape
        mlists = bias_mat if isinstance(bias_mat, MaskLists) else None
        if mlists is not None:
            # prebuilt lists of a mask that does not change: no mask read, no compaction (same results bit for bit)
            if (mlists.B, mlists.N) != (B, N):
                raise ValueError("the lists were built for [%d,%d,%d] masks, the input is [%d,%d,.]" % (mlists.B, mlists.N, mlists.N, B, N))
            if cfg.H * cfg.dp != 64 or cfg.dp < 8 or not mlists.fits(cfg.H, cfg.dp):
                raise ValueError("MaskLists go with the graph-resident kernels: rows of 64 floats in heads of >= 8, at least 256 "
                                 "graphs whose lists fit beside their rows in LDS (heads=%d width=%d, %r)" % (cfg.H, cfg.dp, mlists))
            bias_mat = None
        else:
            bias_mat = _f32(bias_mat).contiguous()
            if tuple(bias_mat.shape) != (B, N, N):
                raise ValueError("bias_mat must be [B,N,N] = [%d,%d,%d], got %s" % (B, N, N, tuple(bias_mat.shape)))
            cfg = _widen_for_lists(cfg, bias_mat)
        H, D, dp = cfg.H, cfg.D, cfg.dp
        hd = H * dp
        st = _lib.current_stream()
        x2 = x.reshape(B * N, F)
        Wp, a1p, a2p, Wrp, Wh, f1, f2, res = _project(x2, W, a1, a2, b1, b2, cfg, W_res, b_res)
        biasp = None if bias is None else _pad_last(bias.reshape(H, D), D, dp).reshape(hd)
        need_grad = any(ctx.needs_input_grad)
        dropping = cfg.coef_keep < 1.0 or cfg.feat_keep < 1.0        # the dropout kernels write the saved tensors
        out = _new((B * N, hd), x)
        agg = pos = lse = q = None
        if need_grad or dropping:
            agg, pos = _new((B * N, hd), x), _new((B * N, hd), x)
            lse, q = _new((B * N, H), x), _new((B * N, H), x)
        lists = mlists is None and _use_mask_lists(cfg, bias_mat)
        present = None
        if mlists is not None:
            _lib.call("gatmi_dense_attn_fwd_lists", B, N, H, dp, ptr(Wh), ptr(f1), ptr(f2), ptr(mlists.lptr), ptr(mlists.lcol),
                      ptr(mlists.lord), mlists.max_fwd, ptr(biasp), ptr(res), _ACT[cfg.activation], cfg.slope,
                      cfg.coef_keep, cfg.feat_keep, cfg.seed, ptr(out), ptr(agg), ptr(pos), ptr(lse), ptr(q), st)
        elif lists:
            if agg is not None:      # visited-entry bits for the backward pass
                present = torch.empty((B * N, _lib.lib.gatmi_dense_mask_words(N)), dtype=torch.int32, device=x.device)
            _lib.call("gatmi_dense_attn_fwd_masked", B, N, H, dp, ptr(Wh), ptr(f1), ptr(f2), ptr(bias_mat), ptr(biasp), ptr(res),
                      _ACT[cfg.activation], cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed,
                      ptr(out), ptr(agg), ptr(pos), ptr(lse), ptr(q), ptr(present), st)
        else:
            _lib.call("gatmi_dense_attn_fwd", B, N, H, dp, ptr(Wh), ptr(f1), ptr(f2), ptr(bias_mat), ptr(biasp), ptr(res),
                      _ACT[cfg.activation], cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed,
                      ptr(out), ptr(agg), ptr(pos), ptr(lse), ptr(q), st)
        if need_grad:
            ctx.cfg, ctx.dims, ctx.present, ctx.mlists = cfg, (B, N), present, mlists
            ctx.has_bias, ctx.has_res = bias is not None, W_res is not None
            ctx.save_for_backward(x2, Wp, a1p, a2p, Wrp, Wh, f1, f2, out, agg, pos, lse, q, bias_mat, biasp)
        o = out.reshape(B, N, H, dp)
        return (o[..., :D] if D != dp else o).reshape(B, N, H * D)

    @staticmethod
    def backward(ctx, dout):
        x2, Wp, a1p, a2p, Wrp, Wh, f1, f2, out, agg, pos, lse, q, bias_mat, biasp = ctx.saved_tensors
        cfg = ctx.cfg
        B, N = ctx.dims
        H, D, dp = cfg.H, cfg.D, cfg.dp
        hd = H * dp
        n = B * N
        st = _lib.current_stream()
        dout = _pad_last(dout.reshape(n, H, D), D, dp).reshape(n, hd).contiguous()
        df1 = _new((n, H), x2)
        dbias = _new((hd,), x2)
        dWh = _new((n, hd), x2)
        df2 = _new((n, H), x2)
        ml = ctx.mlists
        if ml is not None or (ctx.present is not None and _lib.lib.gatmi_dense_attn_bwd_fused_ok(B, N, H, dp, cfg.coef_keep, cfg.feat_keep)):
            # batches of many small graphs: the block that owns a graph forms g, the row statistics and df1 of its
            # nodes while it stages them (no per-destination launch, no round trip of g / stats through memory)
            g = _new((n, hd), x2) if ctx.has_res else None          # only the residual branch reads g afterwards
            ws = _new((_lib.lib.gatmi_dense_attn_bwd_fused_ws_floats(B, H, dp),), x2)
            lists = (None, None, None, 0) if ml is None else (ptr(ml.cptr), ptr(ml.crow), ptr(ml.cord), ml.max_bwd)
            # over prebuilt lists the kernel also sums da1, da2, db1, db2 while it holds the graph's Wh rows and df
            attn_grads = None
            if _lib.lib.gatmi_dense_attn_bwd_fused_param_grads(N, H, dp, 0 if ml is None else 1):
                attn_grads = (_new((hd,), x2), _new((hd,), x2), _new((H,), x2), _new((H,), x2))
            _lib.call("gatmi_dense_attn_bwd_fused", B, N, H, dp, ptr(dout), ptr(out) if ctx.has_res else None, ptr(agg), ptr(pos),
                      ptr(q), ptr(f1), ptr(lse), ptr(biasp), _ACT[cfg.activation], ptr(Wh), ptr(f2), ptr(bias_mat), ptr(ctx.present),
                      *lists, ptr(a1p), ptr(a2p), cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed,
                      ptr(g), ptr(df1), ptr(dbias), *((None,) * 4 if attn_grads is None else [ptr(t) for t in attn_grads]),
                      ptr(ws), ptr(dWh), ptr(df2), st)
            dx, dW, da1, da2, db1, db2, dWr = _project_backward(x2, Wp, Wrp, Wh, dWh, df1, df2, g, cfg, ctx.needs_input_grad[0], attn_grads)
            if dx is not None:
                dx = dx.reshape(B, N, -1)
            db = dbias.reshape(H, dp)[:, :D] if ctx.has_bias else None
            dbr = dbias.reshape(H, dp)[:, :D] if ctx.has_res else None
            return dx, dW, da1, da2, db1, db2, db, dWr, dbr, None, None
        g = _new((n, hd), x2)
        stats = _new((n, 3 * H), x2)
        ws = _new((_lib.lib.gatmi_gat_bwd_node_ws_floats(H, dp),), x2)
        _lib.call("gatmi_gat_bwd_node", n, H, dp, ptr(dout), ptr(out) if ctx.has_res else None, ptr(agg), ptr(pos), ptr(q), ptr(f1),
                  ptr(lse), ptr(biasp), _ACT[cfg.activation], _lib.REDUCE_CONCAT, cfg.slope, ptr(g), hd, None, 0, ptr(stats), 3 * H, ptr(df1),
                  ptr(dbias), ptr(ws), st)
        if ctx.present is not None:
            _lib.call("gatmi_dense_attn_bwd_masked", B, N, H, dp, ptr(g), ptr(stats), ptr(Wh), ptr(f2), ptr(bias_mat),
                      ptr(df1), ptr(a1p), ptr(a2p), ptr(ctx.present), cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed,
                      ptr(dWh), ptr(df2), st)
        else:
            _lib.call("gatmi_dense_attn_bwd", B, N, H, dp, ptr(g), ptr(stats), ptr(Wh), ptr(f2), ptr(bias_mat),
                      ptr(df1), ptr(a1p), ptr(a2p), cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed, ptr(dWh), ptr(df2), st)
        dx, dW, da1, da2, db1, db2, dWr = _project_backward(x2, Wp, Wrp, Wh, dWh, df1, df2, g, cfg, ctx.needs_input_grad[0])
        if dx is not None:
            dx = dx.reshape(B, N, -1)
        db = dbias.reshape(H, dp)[:, :D] if ctx.has_bias else None
        dbr = dbias.reshape(H, dp)[:, :D] if ctx.has_res else None
        return dx, dW, da1, da2, db1, db2, db, dWr, dbr, None, None


def make_cfg(H, D, activation="elu", reduce="concat", coef_drop=0.0, in_drop=0.0, training=False, seed=None,
             slope=LEAKY_SLOPE, storage="f32", mask_path="auto", x_drop=0.0):
    """Static description of one layer call.  coef_drop / in_drop are the reference's dropout
    rates (utils/layers.py:20,22); they act only when training is true, and rate 0 is an exact
    identity (no mask is drawn).  x_drop: the same in_drop rate applied to the layer's INPUT (utils/layers.py:9)
    inside the projection kernels, for inputs that allow it (input_dropout_fusable; the caller then does not drop
    the input itself).  storage="bf16" keeps the two gathered tables of the CSR path (Wh
    rows in the forward sweep, g rows in the backward sweep) in bf16 — half the gather traffic,
    fp32 arithmetic, results within the 2e-2 class instead of 1e-5.  mask_path (dense [B,N,N] masks
    only): "auto" picks the compacted-list kernels for sparse masks and the matrix-core kernels for
    dense ones; "lists" / "dense" force one of them (same results)."""
    if activation not in _ACT:
        raise ValueError("activation must be 'elu', 'relu' or 'identity', got %r" % (activation,))
    if not supported_shape(H, D):
        raise ValueError("heads=%d width=%d: the fused kernels need a power-of-two head count and heads * "
                         "padded width <= 256; call the layer per head instead" % (H, D))
    c = _Cfg()
    c.H, c.D, c.dp = int(H), int(D), padded_width(int(D))
    c.activation, c.reduce, c.slope = activation, reduce, float(slope)
    c.coef_keep = 1.0 - float(coef_drop) if training else 1.0
    c.feat_keep = 1.0 - float(in_drop) if training else 1.0
    c.x_keep = 1.0 - float(x_drop) if training else 1.0
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (c.coef_keep < 1.0 or c.feat_keep < 1.0 or c.x_keep < 1.0) else 0
    c.seed = int(seed)
    if storage not in ("f32", "bf16"):
        raise ValueError("storage must be 'f32' or 'bf16'")
    c.storage = storage
    if mask_path not in ("auto", "lists", "dense"):
        raise ValueError("mask_path must be 'auto', 'lists' or 'dense'")
    c.mask_path = mask_path
    return c


def sparse_gat_layer(x, W, a1, a2, b1, b2, bias, graph, cfg, W_res=None, b_res=None):
    out = SparseGATLayerFn.apply(x, W, a1, a2, b1, b2, bias, W_res, b_res, graph, cfg)
    if cfg.reduce == "mean" and cfg.D != cfg.dp:
        out = out[:, :cfg.D]                             # a view of the padded rows (see SparseGATLayerFn.forward)
    return out


def padded_base(logits, C):
    """(tensor, row stride) under which a fused loss can read `logits` [n, C] in place: when it is the leading C
    columns of a contiguous [n, ld] tensor it was sliced from (the averaged head of sparse_gat_layer: ld = padded
    width), that tensor — an autograd node itself, so the gradient can go to it directly, in its own layout —
    otherwise the logits made contiguous."""
    base = logits._base
    if (base is not None and base.dim() == 2 and base.is_contiguous() and logits.dim() == 2 and base.shape[0] == logits.shape[0]
            and logits.data_ptr() == base.data_ptr() and logits.stride() == (base.shape[1], 1)
            and C <= base.shape[1] <= padded_width(C) and base.dtype == logits.dtype):
        return base
    return logits.contiguous()


class MaskedCrossEntropyFn(torch.autograd.Function):
    """models/base_gattn.py:140-146 and :158-165 in one sweep (gatmi_masked_ce): (loss, accuracy) of logits [n, ld]
    whose first C columns count, labels [n, C] (one-hot or soft), mask [n] floats or None.  The gradient of the loss
    is formed in the same sweep and handed back in the logits' own layout."""

    @staticmethod
    def forward(ctx, logits, C, labels, mask):
        _require_cuda(logits, labels, mask)
        logits, labels = _f32(logits).contiguous(), _f32(labels).contiguous()
        n, ld = logits.shape
        if tuple(labels.shape) != (n, C) or (mask is not None and tuple(mask.shape) != (n,)):
            raise ValueError("masked cross-entropy: logits [%d,%d(%d)], labels %s, mask %s" % (n, C, ld, tuple(labels.shape), None if mask is None else tuple(mask.shape)))
        if mask is not None:
            mask = mask.to(torch.float32).contiguous()
        need = ctx.needs_input_grad[0]
        dl = torch.empty_like(logits) if need else None
        out3 = _new((3,), logits)
        ws = _new((_lib.lib.gatmi_masked_ce_ws_floats(),), logits)
        _lib.call("gatmi_masked_ce", n, C, ptr(logits), ld, ptr(labels), C, ptr(mask), ptr(dl), ptr(out3), ptr(ws), _lib.current_stream())
        if need:
            ctx.save_for_backward(dl)
            ctx.scaled = False
        loss, acc = out3[0], out3[1]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, gl, _ga):
        (dl,) = ctx.saved_tensors
        if ctx.scaled:                                   # a second sweep over the same tape: the saved rows were scaled in place
            raise RuntimeError("masked cross-entropy: backward through the same forward pass twice is not supported")
        ctx.scaled = True
        gl = gl.to(torch.float32).contiguous()
        _lib.call("gatmi_scale_by_device_scalar", ptr(dl), dl.numel(), ptr(gl), _lib.current_stream())
        return dl, None, None, None


def masked_cross_entropy(logits, labels, mask=None):
    """(loss, accuracy) as 0-dim tensors; logits [n, C] on the GPU."""
    C = logits.shape[-1]
    return MaskedCrossEntropyFn.apply(padded_base(logits, C), C, labels, mask)


def dense_gat_layer(x, W, a1, a2, b1, b2, bias, bias_mat, cfg, W_res=None, b_res=None):
    return DenseGATLayerFn.apply(x, W, a1, a2, b1, b2, bias, W_res, b_res, bias_mat, cfg)


def _bnf_head_sets(H):
    """The kernels sweep two heads per block: (first head, heads per group, groups) for the pairs, which share one
    launch, and for a last single head."""
    sets = []
    if H // 2:
        sets.append((0, 2, H // 2))
    if H % 2:
        sets.append((H - 1, 1, 1))
    return sets


class BnfAggregateFn(torch.autograd.Function):
    """Inner step of the reference's attn_e2v_BNF_V2 (utils/layers.py:173-179, coefficients from attn_v2e_BNF_V2
    :229-237) for H heads sharing x and the weighted adjacency:
        out[b,i,h,:] = sum_j (m2_h [A_ij>0] + A_ij) (m1_h + softmax_j LeakyReLU(f1_ih + f2_jh)) x[b,j,:]
    x [B,N,F], net_mat [B,N,N], W [F,H,Dz], a1,a2 [H,Dz], b1,b2,m1,m2 [H]  ->  [B,N,H,F].
    The logit halves: seq_fts = conv1d(seq, Dz, 1) (:231, no bias) feeds nothing but f_s = conv1d(seq_fts, 1, 1)
    (:232-233), so the two width-1 convolutions are applied as one, f_s[., h] = x . (W[:, h, :] a_s[h]) + b_s[h]: an [F]
    vector per head and half instead of an [n, Dz] table per head, forward and backward.
    All heads of a step are issued from this one autograd node (slices of the variables taken per pair of heads outside
    would each be a node of their own, with a zero-filled full-size gradient and an accumulation per slice), and the
    pairs of heads share one launch of the attention kernels: at the reference's batch of 50 a launch for two heads is
    200 blocks, less than one per compute unit."""

    @staticmethod
    def forward(ctx, x, net_mat, W, a1, a2, b1, b2, m1, m2, slope):
        _require_cuda(x, net_mat, W)
        B, N, F = x.shape
        H, Dz = a1.shape
        if F > 64:
            raise ValueError("BnfAggregateFn: feature width %d > 64" % F)
        fp = padded_width(F)
        n = B * N
        st = _lib.current_stream()
        x2d = _f32(x).reshape(n, F)
        xp = _pad_last(x2d, F, fp)
        A = _f32(net_mat).contiguous()
        a12 = torch.stack([a1.reshape(H, Dz), a2.reshape(H, Dz)])            # [2, H, Dz]
        Wc = torch.einsum("fhd,shd->shf", W.reshape(F, H, Dz), a12).reshape(2 * H, F)
        f12 = torch.addmm(torch.cat([b1.reshape(H), b2.reshape(H)]), x2d, Wc.t()).reshape(n, 2, H)
        m1c, m2c = m1.reshape(H).contiguous(), m2.reshape(H).contiguous()
        out = _new((H, n, fp), x)                 # head-major: what the per-head convolution after it reads, and returns gradients in
        need_grad = any(ctx.needs_input_grad)
        saved = []
        for h0, hg, groups in _bnf_head_sets(H):
            # [2, groups, n, hg]: every group's f1 rows, then every group's f2 rows
            f = f12[:, :, h0:h0 + groups * hg].reshape(n, 2, groups, hg).permute(1, 2, 0, 3).contiguous()
            t2n = lse = q = None
            if need_grad:
                t2n, lse, q = _new((groups, n, hg * fp), x), _new((groups, n, hg), x), _new((groups, n, hg, 2), x)
            wsf = _new((_lib.lib.gatmi_bnf_attn_fwd_ws_floats(B, N, fp, groups),), x) if groups > 1 else None
            _lib.call("gatmi_bnf_attn_fwd", B, N, hg, fp, ptr(xp), ptr(A), ptr(f[0]), ptr(f[1]), ptr(m1c[h0:]), ptr(m2c[h0:]), slope,
                      ptr(out[h0:]), ptr(t2n), ptr(lse), ptr(q), ptr(wsf), groups, st)
            saved += [f, t2n, lse, q]
        if need_grad:
            ctx.dims, ctx.slope = (B, N, F, H, fp), slope
            ctx.save_for_backward(xp, A, m1c, m2c, W, a12, Wc, *saved)
        return out.reshape(H, B, N, fp).permute(1, 2, 0, 3)[..., :F]

    @staticmethod
    def backward(ctx, dout):
        xp, A, m1c, m2c, W, a12, Wc = ctx.saved_tensors[:7]
        per_set = ctx.saved_tensors[7:]
        B, N, F, H, fp = ctx.dims
        slope = ctx.slope
        n = B * N
        st = _lib.current_stream()
        G_all = _pad_last(_f32(dout).permute(2, 0, 1, 3).reshape(H, n, F), F, fp)      # [H, n, fp]; no copy when the gradient arrives head-major
        dx, dA = _new((n, fp), xp), _new((B, N, N), xp)
        df12 = _new((n, 2, H), xp)
        dm = _new((2, H), xp)
        for si, (h0, hg, groups) in enumerate(_bnf_head_sets(H)):
            f, t2n, lse, q = per_set[4 * si:4 * si + 4]
            heads = groups * hg
            G = G_all[h0:h0 + heads]
            df = _new((2, groups, n, hg), xp)                                 # df1 of every group, then df2
            dmg = _new((groups, 2, hg), xp)
            ws2 = _new((max(1, _lib.lib.gatmi_bnf_attn_bwd_ws_floats(B, N, hg, fp, groups)),), xp)
            _lib.call("gatmi_bnf_attn_bwd", B, N, hg, fp, ptr(G), ptr(t2n), ptr(f[0]), ptr(lse), ptr(q), ptr(xp), ptr(A), ptr(f[1]), ptr(m1c[h0:]), ptr(m2c[h0:]), slope,
                      ptr(dx), ptr(dA), int(si > 0), ptr(df[0]), ptr(df[1]), ptr(dmg), ptr(ws2), groups, st)
            dm[:, h0:h0 + heads] = dmg.transpose(0, 1).reshape(2, heads)
            df12[:, :, h0:h0 + heads] = df.permute(2, 0, 1, 3).reshape(n, 2, heads)
        # the logit halves: dWc = x^T df12 and the bias sums over the node rows, then the chain rule through Wc = W . a
        df12 = df12.reshape(n, 2 * H)
        dWc, db = _new((fp, 2 * H), xp), _new((2 * H,), xp)
        wsl = _new((_lib.lib.gatmi_linear_bwd_ws_floats(n, fp, 2 * H),), xp)
        _lib.call("gatmi_linear_bwd_params", ptr(xp), n, fp, fp, ptr(df12), 2 * H, ptr(dWc), ptr(db), ptr(wsl), st)
        dWc = dWc[:F].reshape(F, 2, H)
        dW = torch.einsum("fsh,shd->fhd", dWc, a12)
        da12 = torch.einsum("fhd,fsh->shd", W.reshape(F, H, -1), dWc)
        dxs = torch.addmm(dx[:, :F], df12, Wc)
        return dxs.reshape(B, N, F), dA, dW.reshape(W.shape), da12[0], da12[1], db[:H], db[H:], dm[0], dm[1], None


def bnf_aggregate(x, net_mat, W, a1, a2, b1, b2, m1, m2, slope=LEAKY_SLOPE):
    return BnfAggregateFn.apply(x, net_mat, W, a1, a2, b1, b2, m1, m2, float(slope))


_DENSE_ACT = {"identity": _lib.ACT_IDENTITY, "elu": _lib.ACT_ELU, "relu": _lib.ACT_RELU}


def _torch_act(x, activation):
    if activation == "relu":
        return torch.relu(x)
    return torch.nn.functional.elu(x) if activation == "elu" else x


class TallLinearFn(torch.autograd.Function):
    """y = act(x . W^T + b) for x [n, f_in] (rows may be strided), W [w, f_in] and b [w] laid out as torch.nn.Linear
    keeps them: the width-1 convolutions around the brain-network attention step (tf.layers.conv1d(., w, 1, activation),
    utils/layers.py:180,183,187,223-224).  The products are ordinary GEMMs (torch; bias and ReLU ride in the GEMM's
    epilogue); the activation's backward and the bias gradient are one pass (gatmi_bias_act_bwd) and the weight gradient,
    a reduction over the n rows, runs on gatmi_linear_bwd_params."""

    @staticmethod
    def forward(ctx, x, W, b, activation):
        _require_cuda(x, W)
        act = _DENSE_ACT[activation]
        if act == _lib.ACT_IDENTITY:
            y = torch.addmm(b, x, W.t()) if b is not None else x @ W.t()
        elif act == _lib.ACT_RELU and b is not None and hasattr(torch, "_addmm_activation"):
            y = torch._addmm_activation(b, x, W.t(), use_gelu=False)      # bias and ReLU in the GEMM's epilogue
        else:
            pre = x @ W.t()
            y = torch.empty_like(pre)
            bb = b if b is not None else torch.zeros(W.shape[0], dtype=pre.dtype, device=pre.device)
            _lib.call("gatmi_bias_act_fwd", ptr(pre), pre.shape[0], pre.shape[1], pre.stride(0), ptr(bb), act, ptr(y), _lib.current_stream())
        ctx.save_for_backward(x, W, y if act != _lib.ACT_IDENTITY else None)
        ctx.has_bias, ctx.act = b is not None, act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        n, f_in = x.shape
        w = W.shape[0]
        st = _lib.current_stream()
        dy = _f32(dy).contiguous()
        db = None
        if ctx.act != _lib.ACT_IDENTITY:                 # through the activation, with the bias gradient in the same pass
            dpre, db = torch.empty_like(dy), _new((w,), dy)
            ws = _new((_lib.lib.gatmi_column_sums_ws_floats(n, w),), dy)
            _lib.call("gatmi_bias_act_bwd", ptr(dy), ptr(y), n, w, w, ctx.act, ptr(dpre), ptr(db), ptr(ws), st)
            dy = dpre
        dx = dy @ W if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if x.stride(1) != 1:
                x = x.contiguous()
            dWt = _new((f_in, w), dy)
            dbl = _new((w,), dy) if (ctx.has_bias and db is None) else None
            ws = _new((_lib.lib.gatmi_linear_bwd_ws_floats(n, f_in, w),), dy)
            _lib.call("gatmi_linear_bwd_params", ptr(x), n, f_in, x.stride(0), ptr(dy), w, ptr(dWt), ptr(dbl), ptr(ws), st)
            dW = dWt.t()
            db = db if db is not None else dbl
        return dx, dW, (db if ctx.has_bias else None), None


TALL_LINEAR_ROWS = 1024      # below this many rows the layer stays with torch's own backward


def _rows(x):
    n = 1
    for d in x.shape[:-1]:
        n *= int(d)
    return n


def _tall(x):
    return x.is_cuda and x.dtype == torch.float32 and _rows(x) >= TALL_LINEAR_ROWS and torch.is_grad_enabled()


def tall_linear(x, weight, bias=None, activation="identity"):
    """act(torch.nn.functional.linear(x, weight, bias)) for inputs of many rows and modest widths (x [..., f_in] on the
    GPU, fp32; activation "identity", "relu" or "elu"): same result, parameter gradients from the library's fixed-order
    reductions."""
    if activation not in _DENSE_ACT:
        raise ValueError("activation must be 'identity', 'relu' or 'elu', got %r" % (activation,))
    if not _tall(x):
        return _torch_act(torch.nn.functional.linear(x, weight, bias), activation)
    if x.dim() == 2:
        return TallLinearFn.apply(x, weight, bias, activation)
    return TallLinearFn.apply(x.reshape(_rows(x), x.shape[-1]), weight, bias, activation).reshape(*x.shape[:-1], weight.shape[0])


class RowBiasFn(torch.autograd.Function):
    """y = act(x + b) for x [n, w]: bias and activation of a width-1 convolution (utils/layers.py:180) in one pass each
    way (gatmi_bias_act_fwd / _bwd: the backward forms dy * act'(y) and the bias gradient, its column sums, together)."""

    @staticmethod
    def forward(ctx, x, b, activation):
        _require_cuda(x, b)
        act = _DENSE_ACT[activation]
        y = torch.empty_like(x)
        _lib.call("gatmi_bias_act_fwd", ptr(x), x.shape[0], x.shape[1], x.stride(0), ptr(b), act, ptr(y), _lib.current_stream())
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        n, w = y.shape
        d = _f32(dy).contiguous()
        dpre, db = torch.empty_like(d), _new((w,), d)
        ws = _new((_lib.lib.gatmi_column_sums_ws_floats(n, w),), d)
        _lib.call("gatmi_bias_act_bwd", ptr(d), ptr(y), n, w, w, ctx.act, ptr(dpre), ptr(db), ptr(ws), _lib.current_stream())
        return dpre, db, None


def add_row_bias(x, bias, activation="identity"):
    """act(x [..., w] + bias [w]) with the activation's backward and the bias gradient from one pass of the library
    (many rows, fp32, GPU, contiguous); torch's own broadcast add and activation otherwise."""
    if activation not in _DENSE_ACT:
        raise ValueError("activation must be 'identity', 'relu' or 'elu', got %r" % (activation,))
    w = x.shape[-1]
    if not _tall(x) or not x.is_contiguous():
        return _torch_act(x + bias, activation)
    return RowBiasFn.apply(x.reshape(_rows(x), w), bias.reshape(w).contiguous(), activation).reshape(x.shape)


class InnerProductDecoderFn(torch.autograd.Function):
    """utils/layers.py:307-313: act(dropout(z) . dropout(z)^T) per graph, z [B,N,F] -> [B,N,N]."""

    @staticmethod
    def forward(ctx, z, activation, keep, seed):
        _require_cuda(z)
        B, N, F = z.shape
        zc = _f32(z).contiguous()
        out = _new((B, N, N), zc)
        _lib.call("gatmi_ipd_fwd", B, N, F, ptr(zc), activation, keep, seed, ptr(out), _lib.current_stream())
        if ctx.needs_input_grad[0]:
            ctx.args = (activation, keep, seed)
            ctx.save_for_backward(zc, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        zc, out = ctx.saved_tensors
        activation, keep, seed = ctx.args
        B, N, F = zc.shape
        dz = _new((B, N, F), zc)
        _lib.call("gatmi_ipd_bwd", B, N, F, ptr(_f32(dout).contiguous()), ptr(out), ptr(zc), activation, keep, seed,
                  ptr(dz), _lib.current_stream())
        return dz, None, None, None


def inner_product_decoder(z, activation="sigmoid", dropout=0.0, seed=None):
    """sigmoid (or identity) of the Gram matrix of each graph's node embeddings; dropout is the
    reference's rate on the inputs (layers.py:308), 0 = exact identity."""
    act = {"sigmoid": _lib.ACT_SIGMOID, "identity": _lib.ACT_IDENTITY}.get(activation)
    if act is None:
        raise ValueError("activation must be 'sigmoid' or 'identity', got %r" % (activation,))
    if z.shape[-1] > 128:
        raise ValueError("inner_product_decoder: embedding width %d > 128" % z.shape[-1])
    keep = 1.0 - float(dropout)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if keep < 1.0 else 0
    return InnerProductDecoderFn.apply(z, act, keep, int(seed))
