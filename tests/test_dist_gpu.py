"""Two ranks sharing the one visible GPU (gloo backend, buffers staged through the host): the
row-partitioned layer with the real HIP kernels must reproduce the single-GPU layer, forward and
backward.  (RCCL refuses two ranks on one device; the collectives' plumbing is the same.)"""
import datetime
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, light, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
    try:
        from gat_amd import CSRGraph, dist as gdist, graph, ops
        from tests.util import assert_close, layer_params, random_graph
        dev = torch.device("cuda:0")
        n, F, H, D = 700, 20, 8, 8
        dst, src = random_graph(n, 5.0, 11, hubs=3, hub_deg=400)
        rowptr, col, _ = graph.csr_from_coo_host(n, dst, src)
        part = gdist.partition_rows(n, world, rowptr, balance=1)
        lo, hi = int(part[rank]), int(part[rank + 1])
        p = layer_params(F, H, D, seed=5)
        x = np.random.default_rng(6).normal(0, 1, (n, F)).astype(np.float32)
        dout = np.random.default_rng(7).normal(0, 1, (n, H * D)).astype(np.float32)
        cfg = ops.make_cfg(H, D, "elu", "concat")

        def leaves():
            return {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        # single-GPU result (every rank computes it for itself)
        tp = leaves()
        xs = torch.tensor(x, device=dev, requires_grad=True)
        ref = ops.sparse_gat_layer(xs, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"],
                                   CSRGraph.from_coo(dst, src, n, dev, chunk=64), cfg)
        ref.backward(torch.tensor(dout, device=dev))
        # partitioned
        pg = gdist.PartitionedGraph.from_global_csr(rowptr, col, part, rank, world, dev, chunk=64)
        dp_ = leaves()
        xd = torch.tensor(x[lo:hi], device=dev, requires_grad=True)
        out = gdist.dist_sparse_gat_layer(xd, dp_["W"], dp_["a1"], dp_["a2"], dp_["b1"], dp_["b2"], dp_["bias"], pg, cfg)
        out.backward(torch.tensor(dout[lo:hi], device=dev))
        gdist.allreduce_grads(list(dp_.values()))
        torch.cuda.synchronize()
        assert_close(out.detach().cpu().numpy(), ref.detach().cpu().numpy()[lo:hi], rtol=2e-6, atol_scale=1e-6, what="out")
        assert_close(xd.grad.cpu().numpy(), xs.grad.cpu().numpy()[lo:hi], rtol=1e-5, atol_scale=2e-6, what="dX")
        for k in p:
            assert_close(dp_[k].grad.cpu().numpy(), tp[k].grad.cpu().numpy(), rtol=1e-5, atol_scale=5e-6, what="d" + k)
        # the default is the pairwise pipeline.  Without overlap (one all-to-all per direction, exchange and kernels
        # in turn) every kernel sees the same lists in the same order: bit-identical.  With the two-phase forward
        # sweep the sums run in another order.
        assert len(pg.items_a) and len(pg.items_b) and pg.n_slots_ab > 0 and len(pg.items_csc_halo) >= pg.n_halo > 0
        assert sum(len(t) for t in pg.items_csc_by_owner) == len(pg.items_csc_halo) and len(pg.items_csc_by_owner[rank]) == 0
        assert sum(len(t) for t in pg.split_csc_by_owner) == len(pg.split_csc_halo)
        for ov in (False, "two_phase"):
            s_ = leaves()
            xq = torch.tensor(x[lo:hi], device=dev, requires_grad=True)
            oseq = gdist.dist_sparse_gat_layer(xq, s_["W"], s_["a1"], s_["a2"], s_["b1"], s_["b2"], s_["bias"], pg, cfg, overlap=ov)
            oseq.backward(torch.tensor(dout[lo:hi], device=dev))
            gdist.allreduce_grads(list(s_.values()))
            torch.cuda.synchronize()
            if ov is False:
                assert torch.equal(oseq, out) and torch.equal(xq.grad, xd.grad), "pipelined and sequential exchange differ"
                for k in p:
                    assert torch.equal(s_[k].grad, dp_[k].grad), "pipelined vs sequential d" + k
            else:
                assert_close(oseq.detach().cpu().numpy(), ref.detach().cpu().numpy()[lo:hi], rtol=2e-6, atol_scale=1e-6, what="two-phase out")
                assert_close(xq.grad.cpu().numpy(), xs.grad.cpu().numpy()[lo:hi], rtol=1e-5, atol_scale=2e-6, what="two-phase dX")
                for k in p:
                    assert_close(s_[k].grad.cpu().numpy(), tp[k].grad.cpu().numpy(), rtol=1e-5, atol_scale=5e-6, what="two-phase d" + k)
        if light:
            q.put((rank, "ok"))
            return
        # first-layer mode: halo sources' input rows replicated once, no exchange inside the step
        r_ = leaves()
        xr = pg.replicate_rows(torch.tensor(x[lo:hi], device=dev))
        assert xr.shape[0] == pg.n_own + pg.n_halo
        halo_ids = np.setdiff1d(np.unique(col[int(rowptr[lo]):int(rowptr[hi])]), np.arange(lo, hi))
        assert torch.equal(xr[pg.n_own:].cpu(), torch.tensor(x[halo_ids]))
        orep = gdist.dist_sparse_gat_layer(xr, r_["W"], r_["a1"], r_["a2"], r_["b1"], r_["b2"], r_["bias"], pg, cfg, halo="replicate")
        orep.backward(torch.tensor(dout[lo:hi], device=dev))
        gdist.allreduce_grads(list(r_.values()))
        assert_close(orep.detach().cpu().numpy(), ref.detach().cpu().numpy()[lo:hi], rtol=2e-6, atol_scale=1e-6, what="replicate out")
        for k in p:
            assert_close(r_[k].grad.cpu().numpy(), tp[k].grad.cpu().numpy(), rtol=1e-5, atol_scale=5e-6, what="replicate d" + k)
        o16r = gdist.dist_sparse_gat_layer(xr, r_["W"], r_["a1"], r_["a2"], r_["b1"], r_["b2"], r_["bias"], pg,
                                           ops.make_cfg(H, D, "elu", "concat", storage="bf16"), halo="replicate")
        assert_close(o16r.detach().cpu().numpy(), ref.detach().cpu().numpy()[lo:hi], rtol=2e-2, atol_scale=2e-2, what="replicate bf16 out")
        try:
            gdist.dist_sparse_gat_layer(xr.clone().requires_grad_(True), r_["W"], r_["a1"], r_["a2"], r_["b1"], r_["b2"],
                                        r_["bias"], pg, cfg, halo="replicate")
            raise AssertionError("replicate mode accepted an input that needs a gradient")
        except ValueError:
            pass
        # bf16 storage: bf16 rows on the wire, same result class as the single-GPU bf16 layer
        cfg16 = ops.make_cfg(H, D, "elu", "concat", storage="bf16")
        b_ = leaves()
        xb = torch.tensor(x[lo:hi], device=dev, requires_grad=True)
        o16 = gdist.dist_sparse_gat_layer(xb, b_["W"], b_["a1"], b_["a2"], b_["b1"], b_["b2"], b_["bias"], pg, cfg16)
        o16.backward(torch.tensor(dout[lo:hi], device=dev))
        gdist.allreduce_grads(list(b_.values()))
        assert_close(o16.detach().cpu().numpy(), ref.detach().cpu().numpy()[lo:hi], rtol=2e-2, atol_scale=2e-2, what="bf16 out")
        assert_close(b_["W"].grad.cpu().numpy(), tp["W"].grad.cpu().numpy(), rtol=2e-2, atol_scale=2e-2, what="bf16 dW")
        # other shapes, a third of the rows on one rank, the identity / mean output layer
        for (n2, F2, H2, D2, act2, red2, seed2) in ((333, 7, 1, 7, "identity", "mean", 1), (1000, 33, 4, 16, "relu", "concat", 2), (64, 5, 2, 4, "elu", "concat", 3)):
            dst2, src2 = random_graph(n2, 4.0, seed2, hubs=1, hub_deg=min(n2, 300))
            rowptr2, col2, _ = graph.csr_from_coo_host(n2, dst2, src2)
            part2 = np.array([0, n2 // 3, n2], dtype=np.int64) if seed2 == 1 else gdist.partition_rows(n2, world, rowptr2, balance=seed2 % 2)
            lo2, hi2 = int(part2[rank]), int(part2[rank + 1])
            p2 = layer_params(F2, H2, D2, seed=seed2)
            x2 = np.random.default_rng(seed2).normal(0, 1, (n2, F2)).astype(np.float32)
            cfg2 = ops.make_cfg(H2, D2, act2, red2)
            mk = lambda: {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p2.items()}
            t2 = mk()
            xs2 = torch.tensor(x2, device=dev, requires_grad=True)
            ref2 = ops.sparse_gat_layer(xs2, t2["W"], t2["a1"], t2["a2"], t2["b1"], t2["b2"], t2["bias"],
                                        CSRGraph.from_coo(dst2, src2, n2, dev, chunk=32), cfg2)
            do2 = np.random.default_rng(seed2 + 9).normal(0, 1, tuple(ref2.shape)).astype(np.float32)
            ref2.backward(torch.tensor(do2, device=dev))
            pg2 = gdist.PartitionedGraph.from_global_csr(rowptr2, col2, part2, rank, world, dev, chunk=32)
            for mode in ("exchange", "replicate"):
                d2 = mk()
                xd2 = torch.tensor(x2[lo2:hi2], device=dev, requires_grad=(mode == "exchange"))
                xin = xd2 if mode == "exchange" else pg2.replicate_rows(xd2)
                o2 = gdist.dist_sparse_gat_layer(xin, d2["W"], d2["a1"], d2["a2"], d2["b1"], d2["b2"], d2["bias"], pg2, cfg2, halo=mode)
                o2.backward(torch.tensor(do2[lo2:hi2], device=dev))
                gdist.allreduce_grads(list(d2.values()))
                tag = "n=%d %s" % (n2, mode)
                assert_close(o2.detach().cpu().numpy(), ref2.detach().cpu().numpy()[lo2:hi2], rtol=2e-6, atol_scale=1e-6, what=tag + " out")
                if mode == "exchange":
                    assert_close(xd2.grad.cpu().numpy(), xs2.grad.cpu().numpy()[lo2:hi2], rtol=1e-5, atol_scale=2e-6, what=tag + " dX")
                for k in p2:
                    assert_close(d2[k].grad.cpu().numpy(), t2[k].grad.cpu().numpy(), rtol=1e-5, atol_scale=5e-6, what=tag + " d" + k)
        # the device-side planner gives the same shard
        e0, e1 = int(rowptr[lo]), int(rowptr[hi])
        dl = torch.tensor(np.repeat(np.arange(lo, hi), np.diff(rowptr[lo:hi + 1])), dtype=torch.int64, device=dev)
        sg = torch.tensor(col[e0:e1].astype(np.int64), device=dev)
        pg2 = gdist.PartitionedGraph.from_device_edges(dl, sg, part, rank, world, chunk=64)
        assert torch.equal(pg2.graph.col, pg.graph.col) and torch.equal(pg2.exchange.send_idx, pg.exchange.send_idx)
        assert pg2.exchange.recv_counts == pg.exchange.recv_counts
        q.put((rank, "ok"))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL %s\n%s" % (exc, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_gpu():
    from tests.util import run_ranks
    run_ranks(_worker, 2, (False,), timeout=300)


def test_four_ranks_on_one_gpu_match_single_gpu():
    """Four ranks in one process group on the one GPU: three pairwise steps per direction with distinct send and
    receive partners (with two ranks they coincide), pipelined, sequential and two-phase, against the single-GPU layer."""
    from tests.util import run_ranks
    run_ranks(_worker, 4, (True,), timeout=300)


def _nccl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=200))
        from gat_amd import dist as gdist
        # the collectives of the partitioned layer in the forms it issues them, on RCCL with one rank
        ex = gdist.HaloExchange.__new__(gdist.HaloExchange)
        ex.rank, ex.world, ex.group, ex.device, ex.backend = 0, 1, None, dev, dist.get_backend()
        assert ex.backend == "nccl"
        rows = torch.randn(37, 64, device=dev)
        out = torch.empty_like(rows)
        ex._a2a(out, rows, [37], [37])                         # device rows, uneven-split form
        ids = torch.arange(11, dtype=torch.int64)
        got = torch.empty(11, dtype=torch.int64)
        ex._a2a(got, ids, [11], [11])                          # host int64 ids staged through the device
        cnt = torch.empty(1, dtype=torch.int64)
        ex._a2a(cnt, torch.tensor([5], dtype=torch.int64), None, None)
        w = torch.nn.Parameter(torch.ones(3, device=dev))
        w.grad = torch.full((3,), 2.0, device=dev)
        dist.all_reduce(w.grad)
        t = torch.tensor([1.5, 2.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(out, rows) and torch.equal(got, ids) and int(cnt) == 5 and float(w.grad[0]) == 2.0
        q.put((0, "ok"))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((0, "FAIL %s\n%s" % (exc, traceback.format_exc())))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rccl_initialises_and_runs_the_collective_forms_on_one_rank():
    """One GPU cannot host two RCCL ranks, but a one-rank communicator still goes through RCCL's
    initialisation and the exact call forms of the partitioned layer (all_to_all_single with split
    lists on device rows and on staged int64 ids, all_reduce SUM / MAX, barrier)."""
    from tests.util import run_ranks
    run_ranks(_nccl_worker, 1, timeout=240)


@pytest.mark.parametrize("H,D", [(8, 8), (2, 16), (1, 4)])
def test_fused_halo_finalize_equals_unpack_then_finalize(H, D):
    """gatmi_gat_bwd_finalize_halo against gatmi_halo_unpack_sum (on dWh and on df2) followed by
    gatmi_gat_bwd_finalize, through the C ABI: same terms in the same order, so bit-identical; rows that no
    peer read (empty ranges) and rows several peers returned are both in the lists."""
    from gat_amd import _lib
    from gat_amd._lib import ptr
    dev = torch.device("cuda:0")
    n, n_send, hd = 500, 1300, H * D
    gen = torch.Generator(device="cpu").manual_seed(5)
    send_idx = torch.randint(0, n - 40, (n_send,), generator=gen).to(dev)           # the last 40 rows are read by nobody
    order = torch.argsort(send_idx, stable=True)
    rows, counts = torch.unique_consecutive(send_idx[order], return_counts=True)
    rptr = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=dev)
    rptr[1:] = torch.cumsum(counts, 0)
    full = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    full[1:] = torch.cumsum(torch.bincount(send_idx, minlength=n), 0)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)
    dfeat0, df20, df1 = rnd(n, hd), rnd(n, H), rnd(n, H)
    recv_feat, recv_df2 = rnd(n_send, hd), rnd(n_send, H)
    a1, a2 = rnd(hd), rnd(hd)
    st = _lib.current_stream()
    want_f, want_2 = dfeat0.clone(), df20.clone()
    _lib.call("gatmi_halo_unpack_sum", ptr(recv_feat), ptr(rows), ptr(rptr), ptr(order), rows.numel(), hd, ptr(want_f), st)
    _lib.call("gatmi_halo_unpack_sum", ptr(recv_df2), ptr(rows), ptr(rptr), ptr(order), rows.numel(), H, ptr(want_2), st)
    _lib.call("gatmi_gat_bwd_finalize", n, H, D, ptr(df1), ptr(want_2), ptr(a1), ptr(a2), ptr(want_f), st)
    got_f, got_2 = dfeat0.clone(), df20.clone()
    _lib.call("gatmi_gat_bwd_finalize_halo", n, H, D, ptr(df1), ptr(got_2), ptr(a1), ptr(a2), ptr(got_f),
              ptr(full), ptr(order), ptr(recv_feat), ptr(recv_df2), st)
    torch.cuda.synchronize()
    assert torch.equal(got_2, want_2) and torch.equal(got_f, want_f)
    assert not torch.equal(got_f, dfeat0)
