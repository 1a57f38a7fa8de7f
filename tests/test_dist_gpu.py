"""Two ranks sharing the one visible GPU (gloo backend, buffers staged through the host): the
row-partitioned layer with the real HIP kernels must reproduce the single-GPU layer, forward and
backward.  (RCCL refuses two ranks on one device; the collectives' plumbing is the same.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
