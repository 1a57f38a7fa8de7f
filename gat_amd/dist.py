"""Row-partitioned graph-attention layer over several GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI).

The reference has no multi-device code at all; what shards here is its per-destination structure:
out_i needs only {Wh_j, f2_j : j in N(i)} (utils/layers.py:16-24).  Rank p owns a contiguous block
of destination rows, their X rows and their CSR rows.  Sources owned by other ranks ("halo") are
listed once per (owner, requester) pair, sorted by global id, and appended after the owned rows:

  forward   K1 on owned rows -> pack rows peers asked for -> all-to-all(v) of Wh and f2 rows ->
            fused edge-softmax + aggregation over owned + halo source rows
  backward  per-destination kernel on owned rows -> per-source kernel over owned + halo rows ->
            reverse all-to-all(v) of the halo rows' dWh / df2 contributions -> add at the owner ->
            finish owned rows -> parameter gradients on owned rows (+ one small all-reduce)

For a FIRST layer the exchange can be traded for memory: its input X is constant over training and
needs no gradient, so every rank fetches the X rows of its halo sources once (`replicate_rows`) and
from then on projects owned + halo rows itself (halo="replicate").  The forward pass then needs no
communication at all, and the backward pass none either: each rank adds its share of dW = X^T.dWh
and of da2 / db2 over owned + halo rows and the small all-reduce of the parameter gradients
completes the sums.  The price is the projection and the weight gradient over n_own + n_halo rows
instead of n_own, and (n_halo x F) floats of the 288 GB of HBM.  Deeper layers (input produced by
the previous layer, gradient needed) use the exchange.

Each peer pair exchanges directly (xGMI is point to point: an all-to-all keeps all 7 links busy,
a ring would be bound by one).  With the "gloo" backend (CPU rehearsal, or several ranks sharing
one GPU in tests) buffers are staged through host memory.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, ops
from ._lib import ptr
from .graph import CSRGraph, DEFAULT_CHUNK


# ------------------------------------------------------------------------------------------------
# planning (integer work)
# ------------------------------------------------------------------------------------------------
def partition_rows(n, nparts, rowptr=None, balance=0):
    """gatmi_partition_rows -> numpy int64 [nparts+1]."""
    part = np.empty(nparts + 1, dtype=np.int64)
    rp = None if rowptr is None else np.ascontiguousarray(rowptr, dtype=np.int64)
    _lib.call("gatmi_partition_rows", n, nparts, ptr(rp), balance, ptr(part))
    return part


def halo_plan_host(part, rank, col_global):
    """gatmi_halo_plan on host arrays -> (col_local int32, halo_ids int64, counts int64 [P])."""
    part = np.ascontiguousarray(part, dtype=np.int64)
    P = len(part) - 1
    cg = np.ascontiguousarray(col_global, dtype=np.int64)
    cl = np.empty(max(len(cg), 1), dtype=np.int32)
    halo = np.empty(max(len(cg), 1), dtype=np.int64)
    counts = np.empty(P, dtype=np.int64)
    nh = ctypes.c_int64()
    _lib.call("gatmi_halo_plan", P, rank, ptr(part), len(cg), ptr(cg), ptr(cl), ptr(halo), ptr(counts), ctypes.byref(nh))
    return cl[:len(cg)].copy(), halo[:nh.value].copy(), counts


def halo_plan_device(part, rank, col_global):
    """Same plan computed with device sorts for graphs that already live on the GPU
    (col_global: int64 device tensor).  Identical output to halo_plan_host."""
    lo, hi = int(part[rank]), int(part[rank + 1])
    own = (col_global >= lo) & (col_global < hi)
    halo = torch.unique(col_global[~own])                     # sorted ascending
    pos = torch.searchsorted(halo, col_global)
    col_local = torch.where(own, col_global - lo, pos + (hi - lo)).to(torch.int32)
    bounds = torch.searchsorted(halo, torch.as_tensor(np.asarray(part), device=halo.device))
    counts = (bounds[1:] - bounds[:-1]).cpu().numpy().astype(np.int64)
    counts[rank] = 0
    return col_local, halo, counts


def _pack_rows(table, idx, n_idx, width, out):
    """out[k] = table[idx[k]] on the current HIP stream (gatmi_halo_pack)."""
    _lib.call("gatmi_halo_pack", ptr(table), ptr(idx), n_idx, width, ptr(out), _lib.current_stream())


def _unpack_add_rows(rows, idx, n_idx, width, table):
    """table[idx[k]] += rows[k] on the current HIP stream (gatmi_halo_unpack_add)."""
    _lib.call("gatmi_halo_unpack_add", ptr(rows), ptr(idx), n_idx, width, ptr(table), _lib.current_stream())


class HaloExchange:
    """Send/receive lists of one rank plus the two collectives built on them."""

    def __init__(self, rank, world, n_own, halo_ids, recv_counts, lo, device, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.n_own, self.n_halo = int(n_own), int(len(halo_ids))
        self.device = torch.device(device)
        self.recv_counts = [int(c) for c in recv_counts]
        self.backend = dist.get_backend(group) if world > 1 else "none"
        # tell every owner how many and which of its rows we read
        if world > 1:
            rc = torch.tensor(self.recv_counts, dtype=torch.int64)
            sc = torch.empty(world, dtype=torch.int64)
            self._a2a(sc, rc, None, None)
            self.send_counts = [int(c) for c in sc.tolist()]
            want = torch.as_tensor(halo_ids, dtype=torch.int64).cpu().contiguous()
            asked = torch.empty(sum(self.send_counts), dtype=torch.int64)
            self._a2a(asked, want, self.send_counts, self.recv_counts)
            self.send_idx = (asked - lo).to(self.device)
            if len(asked) and (int(self.send_idx.min()) < 0 or int(self.send_idx.max()) >= n_own):
                raise RuntimeError("halo plan: a peer asked rank %d for a row it does not own" % rank)
        else:
            self.send_counts = [0] * world
            self.send_idx = torch.empty(0, dtype=torch.int64, device=self.device)
        self.n_send = int(self.send_idx.numel())

    def _a2a(self, out, inp, out_splits, in_splits):
        """all_to_all_single (every rank calls it, also with nothing to send); buffers are staged
        through the memory the backend works on: host for gloo, device for nccl (= RCCL)."""
        if self.backend == "gloo" and (inp.is_cuda or out.is_cuda):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        elif self.backend == "nccl" and not (inp.is_cuda and out.is_cuda):
            o = torch.empty(out.shape, dtype=out.dtype, device=self.device)
            dist.all_to_all_single(o, inp.to(self.device), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def gather_halo(self, table, width):
        """table [n_own + n_halo, width]: owned rows valid; fills the halo rows from their owners."""
        if self.world == 1:
            return
        send = torch.empty((self.n_send, width), dtype=table.dtype, device=table.device)
        if self.n_send:
            _pack_rows(table, self.send_idx, self.n_send, width, send)
        self._a2a(table[self.n_own:], send, self.recv_counts, self.send_counts)

    def scatter_add_halo(self, table, width):
        """Reverse: halo rows of table go back to their owners and are added to the owned rows."""
        if self.world == 1:
            return
        recv = torch.empty((self.n_send, width), dtype=table.dtype, device=table.device)
        self._a2a(recv, table[self.n_own:].contiguous(), self.send_counts, self.recv_counts)
        off = 0
        for c in self.send_counts:          # one launch per peer: inside a peer's list rows are distinct
            if c:
                _unpack_add_rows(recv[off:], self.send_idx[off:], c, width, table)
            off += c


class PartitionedGraph:
    """One rank's shard: local CSRGraph over owned + halo sources, and its HaloExchange."""

    def __init__(self, graph, exchange, part, rank):
        self.graph, self.exchange, self.part, self.rank = graph, exchange, np.asarray(part), rank
        self.n_own, self.n_halo = exchange.n_own, exchange.n_halo

    @classmethod
    def from_global_csr(cls, rowptr, col, part, rank, world, device, chunk=DEFAULT_CHUNK, group=None):
        """Every rank holds the whole (small) CSR on the host and cuts out its rows: tests, Cora."""
        lo, hi = int(part[rank]), int(part[rank + 1])
        rp = np.asarray(rowptr[lo:hi + 1], dtype=np.int64) - int(rowptr[lo])
        cg = np.asarray(col[int(rowptr[lo]):int(rowptr[hi])], dtype=np.int64)
        cl, halo, counts = halo_plan_host(part, rank, cg)
        n_own = hi - lo
        # rows sorted by LOCAL source id (owned first, then halo), the order the device planner yields
        rows = np.repeat(np.arange(n_own), np.diff(rp))
        cl = cl[np.lexsort((cl, rows))]
        g = CSRGraph.from_csr(rp, cl, n_own + len(halo), device, chunk)
        ex = HaloExchange(rank, world, n_own, halo, counts, lo, device, group)
        return cls(g, ex, part, rank)

    @classmethod
    def from_device_edges(cls, dst_global, src_global, part, rank, world, chunk=DEFAULT_CHUNK, group=None):
        """Large shards generated on the device: distinct (dst, src) pairs with dst in this rank's range."""
        lo, hi = int(part[rank]), int(part[rank + 1])
        n_own = hi - lo
        cl, halo, counts = halo_plan_device(part, rank, src_global)
        g = CSRGraph.from_device_coo(dst_global - lo, cl.to(torch.int64), n_own, n_own + int(halo.numel()), chunk)
        ex = HaloExchange(rank, world, n_own, halo, counts, lo, dst_global.device, group)
        return cls(g, ex, part, rank)

    def replicate_rows(self, x_own):
        """x_own [n_own, F] -> [n_own + n_halo, F]: the owned rows followed by the halo sources' rows
        fetched from their owners (one exchange, at set-up).  Input of the layer with halo="replicate"."""
        x_own = ops._f32(x_own).contiguous()
        full = torch.empty((self.n_own + self.n_halo, x_own.shape[1]), dtype=torch.float32, device=x_own.device)
        full[:self.n_own] = x_own
        self.exchange.gather_halo(full, x_own.shape[1])
        return full


# ------------------------------------------------------------------------------------------------
# the layer
# ------------------------------------------------------------------------------------------------
class DistSparseGATLayerFn(torch.autograd.Function):
    """SparseGATLayerFn with the halo exchange between the projection and the aggregation."""

    @staticmethod
    def forward(ctx, x, W, a1, a2, b1, b2, bias, pgraph, cfg, replicate=False):
        ops._require_cuda(x, W)
        x = ops._f32(x).contiguous()
        graph, ex = pgraph.graph, pgraph.exchange
        F = x.shape[1]
        n, n_src = graph.n_dst, graph.n_src
        if replicate:
            if x.shape[0] != n_src:
                raise ValueError("halo='replicate': x must hold owned + halo rows (%d, from replicate_rows), got %d"
                                 % (n_src, x.shape[0]))
            if ctx.needs_input_grad[0]:
                raise ValueError("halo='replicate' is for constant inputs (a first layer): the gradient of replicated "
                                 "rows would have to travel back to their owners; use the exchange")
        elif x.shape[0] != n:
            raise ValueError("x has %d rows but this rank owns %d" % (x.shape[0], n))
        H, D, dp = cfg.H, cfg.D, cfg.dp
        hd = H * dp
        st = _lib.current_stream()
        Wp = ops._pad_last(W.reshape(F, H, D), D, dp).reshape(F, hd)
        a1p = ops._pad_last(a1.reshape(H, D), D, dp).reshape(hd)
        a2p = ops._pad_last(a2.reshape(H, D), D, dp).reshape(hd)
        n_proj = n_src if replicate else n          # rows this rank projects itself
        feat = ops._new((n_src, hd), x)
        f2 = ops._new((n_src, H), x)
        f1 = ops._new((n_proj, H), x)
        _lib.call("gatmi_proj_fwd", ptr(x), n_proj, F, x.stride(0), ptr(Wp), ptr(a1p), ptr(a2p), ptr(b1.contiguous()),
                  ptr(b2.contiguous()), H, dp, ptr(feat), ptr(f1), ptr(f2), None, None, None, st)
        bf16 = cfg.storage == "bf16"
        gtab, gdt = feat, _lib.DTYPE_F32
        if replicate:
            if bf16:
                gtab, gdt = torch.empty((n_src, hd), dtype=torch.bfloat16, device=x.device), _lib.DTYPE_BF16
                _lib.call("gatmi_to_bf16", ptr(feat), n_src * hd, ptr(gtab), st)
        elif bf16:
            # bf16 mode: the gathered table and therefore the halo rows on the wire are bf16 (half the
            # forward exchange); the owned fp32 rows stay for the parameter gradients
            gtab, gdt = torch.empty((n_src, hd), dtype=torch.bfloat16, device=x.device), _lib.DTYPE_BF16
            _lib.call("gatmi_to_bf16", ptr(feat), n * hd, ptr(gtab), st)
            ex.gather_halo(gtab.view(torch.float32), hd // 2)
            feat[n:].copy_(gtab[n:])        # the per-source backward sweep reads its own Wh row in fp32 (exact widening)
        else:
            ex.gather_halo(feat, hd)
        if not replicate:
            ex.gather_halo(f2, H)
        biasp = None if bias is None else ops._pad_last(bias.reshape(H, D), D, dp).reshape(hd)
        need_grad = any(ctx.needs_input_grad)
        out = ops._new((n, dp if cfg.reduce == "mean" else hd), x)
        agg = pos = lse = q = None
        if need_grad:
            agg, pos = ops._new((n, hd), x), ops._new((n, hd), x)
            lse, q = ops._new((n, H), x), ops._new((n, H), x)
        ws = None
        if len(graph.split_csr):
            ws = ops._new((_lib.lib.gatmi_gat_fwd_ws_floats(graph.n_slots_csr, H, dp),), x)
        _lib.call("gatmi_gat_fwd", n, H, dp, ptr(graph.items_csr), len(graph.items_csr),
                  ptr(graph.split_csr) if len(graph.split_csr) else None, len(graph.split_csr),
                  ptr(graph.col), ptr(gtab), gdt, ptr(f1), ptr(f2), ptr(biasp), None,
                  ops._ACT[cfg.activation], ops._RED[cfg.reduce], cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed,
                  ptr(out), ptr(agg), ptr(pos), ptr(lse), ptr(q), ptr(ws), st)
        if need_grad:
            ctx.pgraph, ctx.cfg, ctx.has_bias, ctx.replicate = pgraph, cfg, bias is not None, replicate
            ctx.save_for_backward(x, Wp, a1p, a2p, feat, f1, f2, out, agg, pos, lse, q)
        if cfg.reduce == "mean":
            return out[:, :D] if D != dp else out
        return out.reshape(n, H, dp)[:, :, :D].reshape(n, H * D) if D != dp else out

    @staticmethod
    def backward(ctx, dout):
        x, Wp, a1p, a2p, feat, f1, f2, out, agg, pos, lse, q = ctx.saved_tensors
        pgraph, cfg = ctx.pgraph, ctx.cfg
        graph, ex = pgraph.graph, pgraph.exchange
        H, D, dp = cfg.H, cfg.D, cfg.dp
        hd = H * dp
        n, n_src = graph.n_dst, graph.n_src
        replicate = ctx.replicate
        st = _lib.current_stream()
        if cfg.reduce == "mean":
            dout = ops._pad_last(dout, D, dp)
        else:
            dout = ops._pad_last(dout.reshape(n, H, D), D, dp).reshape(n, hd)
        bf16 = cfg.storage == "bf16"
        g = None if bf16 else ops._new((n, hd), x)
        g16 = torch.empty((n, hd), dtype=torch.bfloat16, device=x.device) if bf16 else None
        stats = ops._new((n, 3 * H), x)
        if replicate:
            # halo rows are no destinations here: their logit gradient df1 belongs to their owner
            df1 = torch.zeros((n_src, H), dtype=torch.float32, device=x.device)
        else:
            df1 = ops._new((n, H), x)
        dbias = ops._new((hd,), x)
        ws = ops._new((_lib.lib.gatmi_gat_bwd_node_ws_floats(H, dp),), x)
        _lib.call("gatmi_gat_bwd_node", n, H, dp, ptr(dout), ptr(out), ptr(agg), ptr(pos), ptr(q), ptr(f1), ptr(lse),
                  ops._ACT[cfg.activation], ops._RED[cfg.reduce], cfg.slope, ptr(g), ptr(g16), ptr(stats), ptr(df1), ptr(dbias), ptr(ws), st)
        dfeat = ops._new((n_src, hd), x)
        df2 = ops._new((n_src, H), x)
        ws2 = None
        if len(graph.split_csc):
            ws2 = ops._new((_lib.lib.gatmi_gat_bwd_ws_floats(graph.n_slots_csc, H, dp),), x)
        _lib.call("gatmi_gat_bwd_csc", n_src, H, dp, ptr(graph.items_csc), len(graph.items_csc),
                  ptr(graph.split_csc) if len(graph.split_csc) else None, len(graph.split_csc),
                  ptr(graph.rowidx), ptr(graph.perm), ptr(g16 if bf16 else g), _lib.DTYPE_BF16 if bf16 else _lib.DTYPE_F32,
                  ptr(stats), ptr(feat), ptr(f2), ptr(df1) if replicate else None, ptr(a1p) if replicate else None,
                  ptr(a2p) if replicate else None,
                  cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed, ptr(dfeat), ptr(df2), ptr(ws2), st)
        if replicate:
            # dfeat already holds dWh (the logit terms df1.a1 + df2.a2 were added by the sweep): this rank's
            # share of every sum over sources, on owned + halo rows; the all-reduce of the parameter
            # gradients adds the shares of the other ranks
            _, dW, da1, da2, db1, db2, _ = ops._project_backward(x, Wp, None, feat, dfeat, df1, df2, g, cfg, False)
            db = dbias.reshape(H, dp)[:, :D] if ctx.has_bias else None
            return None, dW, da1, da2, db1, db2, db, None, None, None
        ex.scatter_add_halo(dfeat, hd)
        ex.scatter_add_halo(df2, H)
        _lib.call("gatmi_gat_bwd_finalize", n, H, dp, ptr(df1), ptr(df2), ptr(a1p), ptr(a2p), ptr(dfeat), st)
        dWh = dfeat[:n]
        dx, dW, da1, da2, db1, db2, _ = ops._project_backward(x, Wp, None, feat, dWh, df1, df2, g, cfg, ctx.needs_input_grad[0])
        db = dbias.reshape(H, dp)[:, :D] if ctx.has_bias else None
        return dx, dW, da1, da2, db1, db2, db, None, None, None


def dist_sparse_gat_layer(x, W, a1, a2, b1, b2, bias, pgraph, cfg, halo="exchange"):
    """halo="exchange": x [n_own, F], Wh / f2 rows of halo sources travel every step (any layer).
    halo="replicate": x [n_own + n_halo, F] from pgraph.replicate_rows (constant first-layer input):
    no communication besides the all-reduce of the parameter gradients."""
    if halo not in ("exchange", "replicate"):
        raise ValueError("halo must be 'exchange' or 'replicate'")
    return DistSparseGATLayerFn.apply(x, W, a1, a2, b1, b2, bias, pgraph, cfg, halo == "replicate")


def allreduce_grads(params, group=None):
    """Sum the (tiny) parameter gradients over ranks in ONE flat all-reduce."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        h = flat.cpu()
        dist.all_reduce(h, group=group)
        flat.copy_(h)
    else:
        dist.all_reduce(flat, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def build_bench_step(rank, world, n_own, e_own, f_in, heads, width, device, chunk=DEFAULT_CHUNK, storage="f32",
                     halo="replicate"):
    """bench.py's N > 1 workload: every rank owns n_own destination rows of an R-MAT graph with
    world * n_own nodes and generates its own e_own in-edges (sources over the whole id range)."""
    from . import synth
    from .layers import GATHeads
    n_global = n_own * world
    part = np.arange(world + 1, dtype=np.int64) * n_own
    dst, src = synth.rmat_graph(n_global, e_own, 4 + 1000 * rank, device, row_range=(int(part[rank]), int(part[rank + 1])))
    pg = PartitionedGraph.from_device_edges(dst, src, part, rank, world, chunk)
    del dst, src
    x = synth.normal((n_own, f_in), 5 + 1000 * rank, device)
    if halo == "replicate":
        x = pg.replicate_rows(x)            # set-up, outside the timed step: the input features are constant
    dout = synth.normal((n_own, heads * width), 8 + 1000 * rank, device)
    m = GATHeads(f_in, heads, width, device=device)
    m.load(synth.glorot_layer(f_in, heads, width, 7, device))
    m.train()
    params = list(m.parameters())
    cfg = ops.make_cfg(heads, width, "elu", "concat", storage=storage)

    def step():
        for p in params:
            p.grad = None
        out = dist_sparse_gat_layer(x, m.W, m.a1, m.a2, m.b1, m.b2, m.bias, pg, cfg, halo)
        out.backward(dout)
        allreduce_grads(params)

    info = {"graph": repr(pg.graph), "n_halo": pg.n_halo, "n_send": pg.exchange.n_send, "halo": halo}
    return step, pg.graph.nnz, info
