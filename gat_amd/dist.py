"""Row-partitioned graph-attention layer over several GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI).

The reference has no multi-device code at all; what shards here is its per-destination structure:
out_i needs only {Wh_j : j in N(i)} (utils/layers.py:16-24; f2_j = <Wh_j, a2> + b2 follows from the row).  Rank p owns a contiguous block
of destination rows, their X rows and their CSR rows.  Sources owned by other ranks ("halo") are
listed once per (owner, requester) pair, sorted by global id, and appended after the owned rows:

  forward   K1 on owned rows -> pack rows peers asked for -> all-to-all(v) of the Wh rows on a
            communication stream WHILE the aggregation kernel sweeps the entries whose sources are
            owned -> second sweep over the entries that read halo rows, taking up each row's parked
            softmax state (two-phase work items, gatmi_work_items_phased)
  backward  per-destination kernel on owned rows -> per-source kernel over the HALO columns ->
            reverse all-to-all(v) of their dWh / df2 contributions WHILE the per-source kernel does
            the owned columns -> one launch adds every peer's rows at the owner -> finish owned
            rows -> parameter gradients on owned rows (+ one small all-reduce)

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


def _unpack_sum_rows(recv, plan, width, table):
    """table[rows[u]] += sum of recv[pos[k]] over each row's list, all peers in one launch, fixed order
    (gatmi_halo_unpack_sum).  plan = (rows, ptr, pos) int64 tensors from HaloExchange."""
    rows, rptr, pos = plan
    _lib.call("gatmi_halo_unpack_sum", ptr(recv), ptr(rows), ptr(rptr), ptr(pos), rows.numel(), width, ptr(table),
              _lib.current_stream())


class _Pending:
    """An exchange in flight on the communication stream: wait() orders the current stream behind it."""

    def __init__(self, event=None, keep=None):
        self.event, self.keep = event, keep

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
        keep, self.keep, self.event = self.keep, None, None
        return keep


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
        self._finish_setup()

    def _finish_setup(self):
        self.n_send = int(self.send_idx.numel())
        # where peer p's rows sit: in the halo range (rows this rank reads, grouped by owner) and in the
        # send list / the receive buffer of the reverse exchange (rows peers read here, grouped by reader)
        self.recv_off = [0] + np.cumsum(self.recv_counts).tolist()
        self.send_off = [0] + np.cumsum(self.send_counts).tolist()
        # the send list inverted: for every owned row some peer reads, where its returned copies sit in
        # the receive buffer of the reverse exchange (rows sorted, copies in peer order)
        order = torch.argsort(self.send_idx, stable=True)
        rows, counts = torch.unique_consecutive(self.send_idx[order], return_counts=True)
        rptr = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=self.device)
        rptr[1:] = torch.cumsum(counts, 0)
        self.unpack_plan = (rows.contiguous(), rptr, order.contiguous())
        # the same lists with a range for EVERY owned row (empty for rows nobody reads): gatmi_gat_bwd_finalize_halo
        full = torch.zeros(self.n_own + 1, dtype=torch.int64, device=self.device)
        if self.n_send:
            full[1:] = torch.cumsum(torch.bincount(self.send_idx, minlength=self.n_own), 0)
        self.row_ptr_all = full
        self._comm_stream = None
        self.timings = None          # list of (start event, end event, bytes sent, bytes received) when enabled

    def enable_timing(self):
        """Bracket every all-to-all with HIP events on the communication stream (bench.py)."""
        self.timings = []

    def a2a_summary(self):
        """(total milliseconds, bytes sent, bytes received, number of all-to-alls) recorded since
        enable_timing(); the events must have completed (synchronise first)."""
        if not self.timings:
            return 0.0, 0, 0, 0
        return (sum(a.elapsed_time(b) for a, b, _, _ in self.timings), sum(t[2] for t in self.timings),
                sum(t[3] for t in self.timings), len(self.timings))

    def _comm(self):
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(self.device)
        return self._comm_stream

    def _mark(self, pairs):
        """Event on the current stream behind which the exchange of `pairs` may start (None for host buffers)."""
        if not pairs or not pairs[0][1].is_cuda:
            return None
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        return ready

    def _exchange(self, pairs, out_splits, in_splits, ready):
        """All-to-all of every (out, inp) pair.  Device buffers: issued on the communication stream
        behind `ready`, so kernels enqueued on the current stream after `ready` was recorded run beside
        it; returns a _Pending to wait on.  Host buffers: done on return."""
        if ready is None:
            for out, inp in pairs:
                self._a2a(out, inp, out_splits, in_splits)
            return _Pending()
        cs = self._comm()
        cs.wait_event(ready)
        with torch.cuda.stream(cs):
            # Stream order this relies on: the nccl backend runs a collective on a stream of its own that first
            # waits for the stream CURRENT at the call (cs: the inputs are ready there) and, when the call is
            # synchronous or its work object is waited for, makes that same current stream wait for the
            # collective.  `done`, recorded on cs afterwards, therefore lies behind the transfer.  Both hold only
            # while cs is the current stream at the call, hence the assertion; _a2a waits for every work object.
            assert torch.cuda.current_stream(self.device) == cs, "the exchange must be issued with the communication stream current"
            for out, inp in pairs:
                if self.timings is not None:
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record(cs)
                self._a2a(out, inp, out_splits, in_splits)
                if self.timings is not None:
                    t1.record(cs)
                    self.timings.append((t0, t1, inp.numel() * inp.element_size(), out.numel() * out.element_size()))
            done = torch.cuda.Event()
            done.record(cs)
        return _Pending(done, [t for pr in pairs for t in pr])

    def peer_steps(self):
        """[(to, frm)] for the world - 1 steps of a pairwise exchange: in step k every rank r sends to r + k and
        receives from r - k, so each step is a perfect matching of senders and receivers and all links are busy."""
        return [((self.rank + k) % self.world, (self.rank - k) % self.world) for k in range(1, self.world)]

    def _sendrecv(self, sends, to, recvs, frm):
        """One step of a pairwise exchange, issued on the current stream: every non-empty tensor of `sends` goes
        to rank `to`, every non-empty tensor of `recvs` is filled by rank `frm` (both ends know the counts, so an
        empty tensor is skipped by both).  nccl: one grouped launch (batch_isend_irecv); the current stream waits
        for it.  gloo: staged through host memory, done on return."""
        sends = [t for t in sends if t.numel()]
        recvs = [t for t in recvs if t.numel()]
        if not sends and not recvs:
            return
        raw = lambda t: t.view(torch.uint8) if t.dtype == torch.bfloat16 else t      # moved as bytes: not every backend knows bfloat16
        stage = self.backend == "gloo" and any(t.is_cuda for t in sends + recvs)
        if stage:
            torch.cuda.current_stream(self.device).synchronize()
            s_buf = [raw(t).cpu() for t in sends]
            r_buf = [torch.empty(raw(t).shape, dtype=raw(t).dtype) for t in recvs]
        else:
            s_buf, r_buf = [raw(t) for t in sends], [raw(t) for t in recvs]
        ops = [dist.P2POp(dist.isend, t, to, self.group) for t in s_buf] + [dist.P2POp(dist.irecv, t, frm, self.group) for t in r_buf]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if stage:
            for t, h in zip(recvs, r_buf):
                raw(t).copy_(h)

    def _step_on_comm(self, ready, sends, to, recvs, frm):
        """_sendrecv on the communication stream behind `ready` (device buffers) or at once (host buffers)."""
        if ready is None:
            self._sendrecv(sends, to, recvs, frm)
            return
        cs = self._comm()
        cs.wait_event(ready)
        with torch.cuda.stream(cs):
            assert torch.cuda.current_stream(self.device) == cs
            if self.timings is not None:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(cs)
            self._sendrecv(sends, to, recvs, frm)
            if self.timings is not None:
                t1.record(cs)
                self.timings.append((t0, t1, sum(t.numel() * t.element_size() for t in sends),
                                     sum(t.numel() * t.element_size() for t in recvs)))

    def _comm_done(self, keep, device_buffers):
        """A _Pending behind everything issued on the communication stream so far."""
        if not device_buffers:
            return _Pending()
        done = torch.cuda.Event()
        done.record(self._comm())
        return _Pending(done, keep)

    def _a2a(self, out, inp, out_splits, in_splits):
        """all_to_all_single (every rank calls it, also with nothing to send); buffers are staged
        through the memory the backend works on: host for gloo, device for nccl (= RCCL).  The work object is
        waited for before returning: with nccl that orders the current stream behind the transfer (the host does
        not block), with gloo the transfer is complete."""
        if inp.dtype == torch.bfloat16:          # moved as bytes (rows stay rows): not every backend knows bfloat16
            inp, out = inp.view(torch.uint8), out.view(torch.uint8)
        if self.backend == "gloo" and (inp.is_cuda or out.is_cuda):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group, async_op=True).wait()
            out.copy_(o)
        elif self.backend == "nccl" and not (inp.is_cuda and out.is_cuda):
            o = torch.empty(out.shape, dtype=out.dtype, device=self.device)
            dist.all_to_all_single(o, inp.to(self.device), out_splits, in_splits, group=self.group, async_op=True).wait()
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=True).wait()

    def gather_halo_pack(self, tables):
        """tables: list of (table [n_own + n_halo, width], width) whose owned rows are valid.  Packs the
        rows the peers asked for (current stream) and returns a ticket for gather_halo_start.  Kernels
        launched between the two calls run beside the exchange."""
        if self.world == 1:
            return None
        pairs = []
        for table, width in tables:
            send = torch.empty((self.n_send, width), dtype=table.dtype, device=table.device)
            if self.n_send:
                _pack_rows(table, self.send_idx, self.n_send, width, send)
            pairs.append((table[self.n_own:], send))
        return pairs, self._mark(pairs)

    def gather_halo_start(self, ticket):
        """Starts the all-to-all(v) that fills the halo rows; returns a _Pending.  Until its wait()
        nothing may read the halo rows."""
        if ticket is None:
            return _Pending()
        pairs, ready = ticket
        return self._exchange(pairs, self.recv_counts, self.send_counts, ready)

    def gather_halo_pairwise(self, tables):
        """The forward exchange as world - 1 pairwise steps instead of one all-to-all: the rows one peer asked for
        are packed (current stream) and leave (communication stream) while the next peer's rows are being packed,
        so only the first pack stands in front of the transfers.  tables as for gather_halo_pack; returns a
        _Pending; until its wait() nothing may read the halo rows."""
        if self.world == 1:
            return _Pending()
        keep, on_device = [], False
        for to, frm in self.peer_steps():
            s0, s1 = self.send_off[to], self.send_off[to + 1]
            r0, r1 = self.n_own + self.recv_off[frm], self.n_own + self.recv_off[frm + 1]
            sends, recvs = [], []
            for table, width in tables:
                send = torch.empty((s1 - s0, width), dtype=table.dtype, device=table.device)
                if s1 > s0:
                    _pack_rows(table, self.send_idx[s0:s1], s1 - s0, width, send)
                sends.append(send)
                recvs.append(table[r0:r1])
                on_device = on_device or table.is_cuda
            keep += sends
            self._step_on_comm(self._mark([(None, sends[0])]) if on_device else None, sends, to, recvs, frm)
        return self._comm_done(keep, on_device)

    def scatter_add_halo_pairwise(self, tables, sweep_owner, wire_bf16=()):
        """The reverse exchange as world - 1 pairwise steps, each started as soon as its rows are final:
        sweep_owner(p) is called (current stream) to produce the halo rows owned by rank p in every table, then
        those rows leave for p on the communication stream while sweep_owner of the next peer runs.  Returns
        a _Pending whose .recvs hold the rows the peers returned (the layout scatter_finalize /
        scatter_add_halo_finish expect: grouped by reader, in send-list order)."""
        if self.world == 1:
            return None
        recvs_all, keep, on_device = [], [], False
        for k, (table, width) in enumerate(tables):
            dt = torch.bfloat16 if (k in wire_bf16 and table.is_cuda) else table.dtype
            recvs_all.append(torch.empty((self.n_send, width), dtype=dt, device=table.device))
            on_device = on_device or table.is_cuda
        for to, frm in self.peer_steps():
            sweep_owner(to)
            h0, h1 = self.n_own + self.recv_off[to], self.n_own + self.recv_off[to + 1]
            s0, s1 = self.send_off[frm], self.send_off[frm + 1]
            sends = []
            for k, (table, width) in enumerate(tables):
                send = table[h0:h1]
                if recvs_all[k].dtype != table.dtype:
                    send = send.to(recvs_all[k].dtype)
                sends.append(send)
            keep += sends
            self._step_on_comm(self._mark([(None, sends[0])]) if on_device else None, sends, to, [r[s0:s1] for r in recvs_all], frm)
        pend = self._comm_done(keep + recvs_all, on_device)
        pend.recvs = recvs_all
        return pend

    def gather_halo(self, table, width):
        """table [n_own + n_halo, width]: owned rows valid; fills the halo rows from their owners."""
        self.gather_halo_start(self.gather_halo_pack([(table, width)])).wait()

    def scatter_add_halo_mark(self, tables, wire_bf16=()):
        """Reverse direction, step 1: the halo rows of every (table, width) are final as of now on the
        current stream (they must not change until scatter_add_halo_finish); returns a ticket.
        wire_bf16: indices of tables whose rows travel rounded to bf16 (half the bytes; the owner widens
        them again before adding: for storage modes that are in the bf16 error class anyway)."""
        if self.world == 1:
            return None
        pairs = []
        for k, (table, width) in enumerate(tables):
            send = table[self.n_own:]
            if k in wire_bf16 and send.is_cuda:
                send = send.to(torch.bfloat16)
            recv = torch.empty((self.n_send, width), dtype=send.dtype, device=table.device)
            pairs.append((recv, send))
        return pairs, self._mark(pairs)

    def scatter_add_halo_start(self, ticket):
        """Step 2: starts the all-to-all(v) that carries those rows back to their owners."""
        if ticket is None:
            return _Pending()
        pairs, ready = ticket
        pend = self._exchange(pairs, self.send_counts, self.recv_counts, ready)
        pend.recvs = [r for r, _ in pairs]
        return pend

    def scatter_add_halo_finish(self, pend, tables):
        """Step 3: waits for the reverse exchange and adds the returned rows to the owned rows (one
        launch per table for all peers, fixed summation order)."""
        if self.world == 1:
            return
        recvs = pend.recvs
        pend.wait()
        for (table, width), recv in zip(tables, recvs):
            if self.n_send:
                _unpack_sum_rows(recv if recv.dtype == table.dtype else recv.to(table.dtype), self.unpack_plan, width, table)

    def scatter_finalize(self, pend, dfeat, df2, df1, a1p, a2p, H, dp):
        """Step 3 of the layer's backward exchange fused with its last step: waits for the reverse exchange of
        (dfeat [.., H*dp], df2 [.., H]) and, in one pass over the owned rows, adds the returned rows in
        fixed order and then the logit terms df1.a1 + df2.a2 (gatmi_gat_bwd_finalize_halo)."""
        n = self.n_own
        st = _lib.current_stream()
        if self.world == 1 or self.n_send == 0:
            if self.world > 1:
                pend.wait()
            _lib.call("gatmi_gat_bwd_finalize", n, H, dp, ptr(df1), ptr(df2), ptr(a1p), ptr(a2p), ptr(dfeat), st)
            return
        recv_feat, recv_df2 = pend.recvs
        pend.wait()
        if recv_feat.dtype != dfeat.dtype:
            recv_feat = recv_feat.to(dfeat.dtype)       # bf16 on the wire: widened before the sum
        _lib.call("gatmi_gat_bwd_finalize_halo", n, H, dp, ptr(df1), ptr(df2), ptr(a1p), ptr(a2p), ptr(dfeat),
                  ptr(self.row_ptr_all), ptr(self.unpack_plan[2]), ptr(recv_feat), ptr(recv_df2), st)

    def scatter_add_halo(self, table, width):
        """Halo rows of table go back to their owners and are added to the owned rows."""
        tables = [(table, width)]
        self.scatter_add_halo_finish(self.scatter_add_halo_start(self.scatter_add_halo_mark(tables)), tables)


class PartitionedGraph:
    """One rank's shard: local CSRGraph over owned + halo sources, and its HaloExchange."""

    def __init__(self, graph, exchange, part, rank):
        self.graph, self.exchange, self.part, self.rank = graph, exchange, np.asarray(part), rank
        self.n_own, self.n_halo = exchange.n_own, exchange.n_halo
        self._phase_lists()

    def _phase_lists(self):
        """Work lists that let the exchanges run beside the kernels.  Forward: every CSR row keeps its
        owned sources first (rows are sorted by local id, halo ids follow the owned ones), so the row is
        cut there: phase A = entries reading owned rows, phase B = entries reading halo rows
        (gatmi_work_items_phased).  Backward: the per-source items are divided into halo columns (swept
        first: their results travel) and owned columns (swept while those travel)."""
        g, dev = self.graph, self.graph.device
        n = g.n_dst
        rowptr = g.rowptr_host
        if g.nnz:
            c = torch.zeros(g.nnz + 1, dtype=torch.int64, device=dev)
            c[1:] = torch.cumsum(g.col < self.n_own, 0, dtype=torch.int64)
            rp = torch.from_numpy(rowptr).to(dev)
            cut = (rp[:-1] + c[rp[1:]] - c[rp[:-1]]).cpu().numpy()
            del c, rp
        else:
            cut = rowptr[:-1].copy()
        cut = np.ascontiguousarray(cut, dtype=np.int64)
        na, nb, ns, nl = (ctypes.c_int64() for _ in range(4))
        _lib.call("gatmi_work_items_phased_count", n, ptr(rowptr), ptr(cut), g.chunk, ctypes.byref(na), ctypes.byref(nb),
                  ctypes.byref(ns), ctypes.byref(nl))
        ia, ib = np.empty((na.value, 4), dtype=np.int32), np.empty((nb.value, 4), dtype=np.int32)
        sp = np.empty((ns.value, 4), dtype=np.int32)
        _lib.call("gatmi_work_items_phased", n, ptr(rowptr), ptr(cut), g.chunk, ptr(ia), ptr(ib), ptr(sp))
        to = lambda a: torch.from_numpy(a).to(dev)
        self.items_a, self.items_b, self.split_ab, self.n_slots_ab = to(ia), to(ib), to(sp), nl.value
        self.local_entries = int((cut - rowptr[:-1]).sum())
        own = g.items_csc[:, 0] < self.n_own
        self.items_csc_own, self.items_csc_halo = g.items_csc[own].contiguous(), g.items_csc[~own].contiguous()
        own = g.split_csc[:, 0] < self.n_own
        self.split_csc_own, self.split_csc_halo = g.split_csc[own].contiguous(), g.split_csc[~own].contiguous()
        # the halo columns once more, grouped by the rank that owns them (halo rows are sorted by global id, so an
        # owner's columns are one range of local ids): the pairwise reverse exchange sweeps one owner's columns,
        # sends their results on their way and goes on with the next owner's
        ex = self.exchange
        edges = torch.tensor([self.n_own + o for o in ex.recv_off[1:]], dtype=self.items_csc_halo.dtype, device=dev)
        io = torch.bucketize(self.items_csc_halo[:, 0].contiguous(), edges, right=True)
        so = torch.bucketize(self.split_csc_halo[:, 0].contiguous(), edges, right=True)
        self.items_csc_by_owner = [self.items_csc_halo[io == p].contiguous() for p in range(ex.world)]
        self.split_csc_by_owner = [self.split_csc_halo[so == p].contiguous() for p in range(ex.world)]

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
    def forward(ctx, x, W, a1, a2, b1, b2, bias, pgraph, cfg, replicate=False, overlap=True):
        ops._require_cuda(x, W)
        x = ops._f32(x).contiguous()
        graph, ex = pgraph.graph, pgraph.exchange
        F = x.shape[1]
        n, n_src = graph.n_dst, graph.n_src
        two_phase_fwd = overlap == "two_phase"
        overlap = overlap not in (False, "off", None, 0)
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
        f2 = ops._new((n_src if cfg.storage == "bf16" else n_proj, H), x)
        f1 = ops._new((n_proj, H), x)
        b2c = b2.reshape(H).contiguous()
        bf16 = cfg.storage == "bf16"
        gtab, gdt, ldf = feat, _lib.DTYPE_F32, hd
        if bf16:
            # bf16 mode: the gathered table (a second store of the projection's epilogue: one record
            # [hd bf16 | the row's fp32 f2 | pad] per row) and therefore the halo rows on the wire are bf16
            # (0.6 of the forward exchange); the fp32 rows stay for the backward sweep and the parameter gradients
            ldf = ops._bf16_record_elems(H, dp)
            gtab, gdt = torch.empty((n_src, ldf), dtype=torch.bfloat16, device=x.device), _lib.DTYPE_BF16
        _lib.call("gatmi_proj_fwd", ptr(x), n_proj, F, x.stride(0), ptr(Wp), ptr(a1p), ptr(a2p), ptr(b1.contiguous()),
                  ptr(b2c), H, dp, ptr(feat), ptr(gtab) if bf16 else None, ldf, ptr(f1), ptr(f2), None, None, None, 1.0, 0, st)
        ticket = pend_fwd = None
        if not replicate and ex.world > 1:
            # fp32: only the Wh rows travel, the receiving sweep forms <Wh_j, a2> + b2 from them; bf16 rows
            # would add their rounding to the logits, so their records carry the exact f2 along
            travel = [(gtab.view(torch.float32), ldf // 2)] if bf16 else [(feat, hd)]
            if overlap:
                pend_fwd = ex.gather_halo_pairwise(travel)      # per peer: pack, send; the sweeps below run beside it
            else:
                ticket = ex.gather_halo_pack(travel)

        def widen_halo():
            # the per-source backward sweep reads its own Wh row and f2 in fp32: exact widening of what arrived
            feat[n:].copy_(gtab[n:, :hd])
            f2[n:].copy_(gtab.view(torch.float32)[n:, hd // 2:hd // 2 + H])
        biasp = None if bias is None else ops._pad_last(bias.reshape(H, D), D, dp).reshape(hd)
        need_grad = any(ctx.needs_input_grad)
        out = ops._new((n, dp if cfg.reduce == "mean" else hd), x)
        agg = pos = lse = q = None
        if need_grad:
            agg, pos = ops._new((n, hd), x), ops._new((n, hd), x)
            lse, q = ops._new((n, H), x), ops._new((n, H), x)

        half = (None, None, None) if bf16 else (None, ptr(a2p), ptr(b2c))           # the source half of the logits comes with the rows

        def sweep(items, split, ws):
            _lib.call("gatmi_gat_fwd", n, H, dp, ptr(items), len(items), ptr(split) if len(split) else None, len(split),
                      ptr(graph.col), ptr(gtab), gdt, ldf, ptr(f1), *half, ptr(biasp), None,
                      ops._ACT[cfg.activation], ops._RED[cfg.reduce], cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed,
                      ptr(out), ptr(agg), ptr(pos), ptr(lse), ptr(q), ptr(ws), st)
        two_phase = pend_fwd is not None and two_phase_fwd and len(pgraph.items_a) and len(pgraph.items_b)
        if two_phase:
            ws = ops._new((max(1, _lib.lib.gatmi_gat_fwd_ws_floats(pgraph.n_slots_ab, H, dp)),), x)
            sweep(pgraph.items_a, pgraph.split_ab[:0], ws)       # entries reading owned rows, beside the exchange
            pend_fwd.wait()
            if bf16:
                widen_halo()
            sweep(pgraph.items_b, pgraph.split_ab, ws)           # entries reading halo rows; merges rows cut further
        else:
            if ticket is not None:
                pend_fwd = ex.gather_halo_start(ticket)
            if pend_fwd is not None:
                pend_fwd.wait()
                if bf16:
                    widen_halo()
            ws = None
            if len(graph.split_csr):
                ws = ops._new((_lib.lib.gatmi_gat_fwd_ws_floats(graph.n_slots_csr, H, dp),), x)
            sweep(graph.items_csr, graph.split_csr, ws)
        if need_grad:
            ctx.pgraph, ctx.cfg, ctx.has_bias, ctx.replicate, ctx.overlap = pgraph, cfg, bias is not None, replicate, overlap
            ctx.save_for_backward(x, Wp, a1p, a2p, feat, f1, f2, b2c, out, agg, pos, lse, q, biasp)
        if cfg.reduce == "mean":
            return out                                   # padded rows [n, dp]; dist_sparse_gat_layer slices (see ops.SparseGATLayerFn)
        return out.reshape(n, H, dp)[:, :, :D].reshape(n, H * D) if D != dp else out

    @staticmethod
    def backward(ctx, dout):
        x, Wp, a1p, a2p, feat, f1, f2, b2c, out, agg, pos, lse, q, biasp = ctx.saved_tensors
        pgraph, cfg = ctx.pgraph, ctx.cfg
        graph, ex = pgraph.graph, pgraph.exchange
        H, D, dp = cfg.H, cfg.D, cfg.dp
        hd = H * dp
        n, n_src = graph.n_dst, graph.n_src
        replicate = ctx.replicate
        st = _lib.current_stream()
        if cfg.reduce == "mean":
            dout = dout.contiguous()
        else:
            dout = ops._pad_last(dout.reshape(n, H, D), D, dp).reshape(n, hd)
        bf16 = cfg.storage == "bf16"
        g = g16 = None
        if bf16:
            ldg = ops._bf16_grad_record_elems(H, dp)             # [g: hd bf16 | f1 lse c | pad] per row
            g16 = torch.empty((n, ldg), dtype=torch.bfloat16, device=x.device)
            stats = g16.view(torch.float32)[:, hd // 2:hd // 2 + 3 * H]
        else:
            rec = ops._new((n, ops._record_width(H, dp)), x)    # [g | f1 lse c | pad] per row: one gather per entry
            g, stats = rec[:, :hd], rec[:, hd:hd + 3 * H]
            ldg = rec.stride(0)
        lds = stats.stride(0)
        if replicate:
            # halo rows are no destinations here: their logit gradient df1 belongs to their owner
            df1 = torch.zeros((n_src, H), dtype=torch.float32, device=x.device)
        else:
            df1 = ops._new((n, H), x)
        dbias = ops._new((hd,), x)
        ws = ops._new((_lib.lib.gatmi_gat_bwd_node_ws_floats(H, dp),), x)
        _lib.call("gatmi_gat_bwd_node", n, H, dp, ptr(dout), None, ptr(agg), ptr(pos), ptr(q), ptr(f1), ptr(lse), ptr(biasp),
                  ops._ACT[cfg.activation], ops._RED[cfg.reduce], cfg.slope, ptr(g), ldg, ptr(g16), ldg, ptr(stats), lds, ptr(df1), ptr(dbias), ptr(ws), st)
        dfeat = ops._new((n_src, hd), x)
        df2 = ops._new((n_src, H), x)
        ws2 = None
        if len(graph.split_csc):
            ws2 = ops._new((_lib.lib.gatmi_gat_bwd_ws_floats(graph.n_slots_csc, H, dp),), x)

        half = (ptr(f2), None, None) if bf16 else (None, ptr(a2p), ptr(b2c))

        def sweep(items, split):
            if len(items) == 0:
                return
            _lib.call("gatmi_gat_bwd_csc", n_src, H, dp, ptr(items), len(items), ptr(split) if len(split) else None, len(split),
                      ptr(graph.rowidx), ptr(graph.perm), ptr(g16 if bf16 else g), _lib.DTYPE_BF16 if bf16 else _lib.DTYPE_F32, ldg,
                      ptr(stats), lds, ptr(feat), *half, ptr(df1) if replicate else None, ptr(a1p) if replicate else None,
                      ptr(a2p) if replicate else None,
                      cfg.slope, cfg.coef_keep, cfg.feat_keep, cfg.seed, ptr(dfeat), ptr(df2), ptr(ws2), st)
        if replicate:
            sweep(graph.items_csc, graph.split_csc)
            # dfeat already holds dWh (the logit terms df1.a1 + df2.a2 were added by the sweep): this rank's
            # share of every sum over sources, on owned + halo rows; the all-reduce of the parameter
            # gradients adds the shares of the other ranks
            _, dW, da1, da2, db1, db2, _ = ops._project_backward(x, Wp, None, feat, dfeat, df1, df2, g, cfg, False)
            db = dbias.reshape(H, dp)[:, :D] if ctx.has_bias else None
            return None, dW, da1, da2, db1, db2, db, None, None, None, None
        tables = [(dfeat, hd), (df2, H)]
        wire16 = (0,) if bf16 else ()        # bf16 storage: the returned dWh rows travel as bf16 too (df2 stays fp32: 32 B per row)
        if ctx.overlap and ex.world > 1:
            # one owner's halo columns at a time: their results leave for the owner while the next owner's columns
            # (and, behind the last owner, the owned columns) are swept
            pend = ex.scatter_add_halo_pairwise(
                tables, lambda p: sweep(pgraph.items_csc_by_owner[p], pgraph.split_csc_by_owner[p]), wire16)
            sweep(pgraph.items_csc_own, pgraph.split_csc_own)
        else:
            sweep(graph.items_csc, graph.split_csc)
            pend = ex.scatter_add_halo_start(ex.scatter_add_halo_mark(tables, wire16))
        ex.scatter_finalize(pend, dfeat, df2, df1, a1p, a2p, H, dp)
        dWh = dfeat[:n]
        dx, dW, da1, da2, db1, db2, _ = ops._project_backward(x, Wp, None, feat, dWh, df1, df2, g, cfg, ctx.needs_input_grad[0])
        db = dbias.reshape(H, dp)[:, :D] if ctx.has_bias else None
        return dx, dW, da1, da2, db1, db2, db, None, None, None, None


def dist_sparse_gat_layer(x, W, a1, a2, b1, b2, bias, pgraph, cfg, halo="exchange", overlap=True):
    """halo="exchange": x [n_own, F], Wh rows of halo sources travel every step (any layer).
    overlap=False ("off"): one all-to-all per direction, exchanges and kernels in turn.
    overlap=True ("pipeline"): each direction is world - 1 pairwise steps on a communication stream — forward,
    a peer's rows leave while the next peer's are packed; backward, the halo columns are swept owner by owner and
    an owner's results leave while the next owner's columns (last: the owned columns) are swept.  Same results
    bit for bit as "off" (the kernels see the same lists in the same order per column).
    overlap="two_phase": "pipeline", and the forward sweep is cut in two — entries that read owned rows run
    beside the exchange, every row's softmax state is parked, a second launch takes it up over the entries that
    read halo rows (same results up to summation order; pays only with slow links and a large owned share).
    halo="replicate": x [n_own + n_halo, F] from pgraph.replicate_rows (constant first-layer input):
    no communication besides the all-reduce of the parameter gradients."""
    if halo not in ("exchange", "replicate"):
        raise ValueError("halo must be 'exchange' or 'replicate'")
    if overlap not in (True, False, "off", "pipeline", "two_phase"):
        raise ValueError("overlap must be False / 'off', True / 'pipeline' or 'two_phase'")
    out = DistSparseGATLayerFn.apply(x, W, a1, a2, b1, b2, bias, pgraph, cfg, halo == "replicate", overlap)
    if cfg.reduce == "mean" and cfg.D != cfg.dp:
        out = out[:, :cfg.D]
    return out


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


class BenchShard:
    """bench.py's N > 1 workload: every rank owns n_own destination rows of an R-MAT graph with
    world * n_own nodes and generates its own e_own in-edges (sources over the whole id range, node
    labels scrambled so that the hubs spread over all owners)."""

    def __init__(self, rank, world, n_own, e_own, f_in, heads, width, device, chunk=DEFAULT_CHUNK, storage="f32"):
        from . import synth
        from .layers import GATHeads
        self.rank, self.world, self.device = rank, world, device
        part = np.arange(world + 1, dtype=np.int64) * n_own
        dst, src = synth.rmat_graph(n_own * world, e_own, 4 + 1000 * rank, device, row_range=(int(part[rank]), int(part[rank + 1])))
        self.pg = PartitionedGraph.from_device_edges(dst, src, part, rank, world, chunk)
        del dst, src
        self.E_rank = self.pg.graph.nnz
        self.x = synth.normal((n_own, f_in), 5 + 1000 * rank, device)
        self.x_rep = None
        self.dout = synth.normal((n_own, heads * width), 8 + 1000 * rank, device)
        self.m = GATHeads(f_in, heads, width, device=device)
        self.m.load(synth.glorot_layer(f_in, heads, width, 7, device))
        self.m.train()
        self.cfg = ops.make_cfg(heads, width, "elu", "concat", storage=storage)

    def info(self):
        pg = self.pg
        return {"graph": repr(pg.graph), "n_halo": pg.n_halo, "n_send": pg.exchange.n_send,
                "two_phase_items": [len(pg.items_a), len(pg.items_b), len(pg.split_ab), pg.n_slots_ab]}

    def step_fn(self, halo, overlap=True):
        m, pg, cfg, dout = self.m, self.pg, self.cfg, self.dout
        params = list(m.parameters())
        if halo == "replicate":
            if self.x_rep is None:
                self.x_rep = pg.replicate_rows(self.x)       # set-up, outside the timed steps: the input features are constant
            x = self.x_rep
        else:
            x = self.x

        def step():
            for p in params:
                p.grad = None
            out = dist_sparse_gat_layer(x, m.W, m.a1, m.a2, m.b1, m.b2, m.bias, pg, cfg, halo, overlap)
            out.backward(dout)
            allreduce_grads(params)
        return step

    def release(self, halo):
        if halo == "replicate":
            self.x_rep = None
