"""world_size-2 rehearsal of the row-partitioned layer on CPU (gloo): partition, halo plan,
send-list handshake, forward halo gather and reverse scatter-add (the send list inverted into one
summation plan for all peers).  The HIP pack / unpack kernels
cannot run without a GPU, so this test swaps them for torch index ops (stand-ins that exist only
here) and runs the oracle's math on each rank's shard: the partitioned result must equal the
single-graph oracle result, forward and backward — with the per-step exchange of projected rows and
with the first-layer mode that fetches the halo sources' input rows once and exchanges nothing else."""
import datetime
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_pack(table, idx, n_idx, width, out):
    out.copy_(table[idx[:n_idx]])


def _cpu_unpack_sum(recv, plan, width, table):
    rows, rptr, pos = plan
    table.index_add_(0, torch.repeat_interleave(rows, rptr[1:] - rptr[:-1]), recv[pos])


def _worker(rank, world, port, n, seed, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        from gat_amd import dist as gdist, graph
        from oracle import ref_numpy, ref_torch
        from tests.util import layer_params, random_graph
        gdist._pack_rows, gdist._unpack_sum_rows = _cpu_pack, _cpu_unpack_sum

        F, H, D = 7, 2, 4
        dst, src = random_graph(n, 4.0, seed, hubs=2, hub_deg=n // 2)
        rowptr, col, _ = graph.csr_from_coo_host(n, dst, src)
        part = gdist.partition_rows(n, world, rowptr, balance=1)
        lo, hi = int(part[rank]), int(part[rank + 1])
        cg = col[rowptr[lo]:rowptr[hi]].astype(np.int64)
        cl, halo, counts = gdist.halo_plan_host(part, rank, cg)
        ex = gdist.HaloExchange(rank, world, hi - lo, halo, counts, lo, "cpu")
        assert sum(ex.recv_counts) == len(halo)

        p = layer_params(F, H, D, seed=1)
        x = np.random.default_rng(2).normal(0, 1, (n, F))
        tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
        tx = torch.tensor(x, requires_grad=True)

        # forward gather: a table holding global ids must come back as exactly the halo ids
        ids = torch.zeros((hi - lo + len(halo), 1), dtype=torch.float64)
        ids[: hi - lo, 0] = torch.arange(lo, hi, dtype=torch.float64)
        ex.gather_halo(ids, 1)
        assert (ids[hi - lo:, 0].numpy() == halo).all()

        # the layer on the shard with the oracle's math: Wh rows of owned nodes + gathered halo rows
        wh = (tx[lo:hi] @ tp["W"].reshape(F, H * D))
        table = torch.cat([wh.detach(), torch.zeros(len(halo), H * D, dtype=torch.float64)])
        ex.gather_halo(table, H * D)
        wh_full = (torch.tensor(x) @ tp["W"].detach().reshape(F, H * D))
        assert torch.equal(table[hi - lo:], wh_full[torch.as_tensor(halo)])

        # reverse path: every rank contributes 1 to each row it reads remotely; owners must end up
        # with the number of remote readers of each owned row
        cnt = torch.zeros((hi - lo + len(halo), 1), dtype=torch.float64)
        cnt[hi - lo:] = 1.0
        ex.scatter_add_halo(cnt, 1)
        readers = np.zeros(n)
        for r in range(world):
            l2, h2 = int(part[r]), int(part[r + 1])
            c2 = np.unique(col[rowptr[l2]:rowptr[h2]])
            c2 = c2[(c2 < l2) | (c2 >= h2)]
            readers[c2] += 1
        assert (cnt[: hi - lo, 0].numpy() == readers[lo:hi]).all()

        # end to end with autograd through a differentiable twin of the exchange
        class Gather(torch.autograd.Function):
            @staticmethod
            def forward(ctx, own):
                t = torch.cat([own, torch.zeros(len(halo), own.shape[1], dtype=own.dtype)])
                ex.gather_halo(t, own.shape[1])
                return t

            @staticmethod
            def backward(ctx, gt):
                gt = gt.clone()
                ex.scatter_add_halo(gt, gt.shape[1])
                return gt[: hi - lo]

        rp_local = torch.tensor(rowptr[lo:hi + 1] - rowptr[lo])
        col_local = torch.tensor(cl)
        xs = tx[lo:hi]
        wh = (xs @ tp["W"].reshape(F, H * D)).reshape(-1, H, D)
        f1 = (wh * tp["a1"]).sum(-1) + tp["b1"]
        f2 = (wh * tp["a2"]).sum(-1) + tp["b2"]
        wh_ext = Gather.apply(wh.reshape(-1, H * D)).reshape(-1, H, D)
        f2_ext = Gather.apply(f2)
        dsts = torch.repeat_interleave(torch.arange(hi - lo), rp_local[1:] - rp_local[:-1])
        e = torch.nn.functional.leaky_relu(f1[dsts] + f2_ext[col_local.long()], 0.2)
        emax = torch.full((hi - lo, H), -1e30, dtype=torch.float64).scatter_reduce(0, dsts[:, None].expand(-1, H), e.detach(), "amax")
        pe = torch.exp(e - emax[dsts])
        den = torch.zeros((hi - lo, H), dtype=torch.float64).index_add(0, dsts, pe)
        a = pe / den[dsts]
        out = torch.zeros((hi - lo, H, D), dtype=torch.float64).index_add(0, dsts, a[:, :, None] * wh_ext[col_local.long()])
        out = torch.nn.functional.elu(out + tp["bias"]).reshape(hi - lo, H * D)
        dout = torch.tensor(np.random.default_rng(3).normal(0, 1, (n, H * D)))
        out.backward(dout[lo:hi])
        gdist.allreduce_grads(list(tp.values()))
        gx = tx.grad.clone()
        dist.all_reduce(gx)

        rp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
        rx = torch.tensor(x, requires_grad=True)
        ref = ref_torch.layer_csr(rx, rp, torch.tensor(rowptr), torch.tensor(col), "elu", "concat")
        ref.backward(dout)
        np.testing.assert_allclose(out.detach().numpy(), ref.detach().numpy()[lo:hi], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(gx.numpy(), rx.grad.numpy(), rtol=1e-10, atol=1e-12)
        for k in p:
            np.testing.assert_allclose(tp[k].grad.numpy(), rp[k].grad.numpy(), rtol=1e-10, atol=1e-12, err_msg=k)
        # first-layer mode: the halo sources' INPUT rows fetched once, everything else local; each rank's
        # parameter gradients are its shares over owned + halo rows and the all-reduce completes them
        xfull = torch.cat([torch.tensor(x[lo:hi]), torch.zeros(len(halo), F, dtype=torch.float64)])
        ex.gather_halo(xfull, F)
        assert torch.equal(xfull[hi - lo:], torch.tensor(x)[torch.as_tensor(halo)])
        tp2 = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
        wh2 = (xfull @ tp2["W"].reshape(F, H * D)).reshape(-1, H, D)
        f1b = (wh2[: hi - lo] * tp2["a1"]).sum(-1) + tp2["b1"]
        f2b = (wh2 * tp2["a2"]).sum(-1) + tp2["b2"]
        e2 = torch.nn.functional.leaky_relu(f1b[dsts] + f2b[col_local.long()], 0.2)
        m2 = torch.full((hi - lo, H), -1e30, dtype=torch.float64).scatter_reduce(0, dsts[:, None].expand(-1, H), e2.detach(), "amax")
        p2 = torch.exp(e2 - m2[dsts])
        a2_ = p2 / torch.zeros((hi - lo, H), dtype=torch.float64).index_add(0, dsts, p2)[dsts]
        out2 = torch.zeros((hi - lo, H, D), dtype=torch.float64).index_add(0, dsts, a2_[:, :, None] * wh2[col_local.long()])
        out2 = torch.nn.functional.elu(out2 + tp2["bias"]).reshape(hi - lo, H * D)
        out2.backward(dout[lo:hi])
        gdist.allreduce_grads(list(tp2.values()))
        np.testing.assert_allclose(out2.detach().numpy(), ref.detach().numpy()[lo:hi], rtol=1e-11, atol=1e-12)
        for k in p:
            np.testing.assert_allclose(tp2[k].grad.numpy(), rp[k].grad.numpy(), rtol=1e-10, atol=1e-12, err_msg="replicate " + k)
        q.put((rank, "ok"))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL %s\n%s" % (exc, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,seed", [(2, 61, 0), (2, 200, 1), (3, 97, 2)])
def test_partitioned_layer_equals_single_graph(world, n, seed):
    from tests.util import run_ranks
    run_ranks(_worker, world, (n, seed), timeout=240)


def test_device_and_host_halo_plans_agree_on_cpu_tensors():
    sys.path.insert(0, ROOT)
    from gat_amd import dist as gdist, graph
    from tests.util import random_graph
    n, P = 300, 4
    dst, src = random_graph(n, 5.0, 3)
    rowptr, col, _ = graph.csr_from_coo_host(n, dst, src)
    part = gdist.partition_rows(n, P)
    for r in range(P):
        cg = col[rowptr[part[r]]:rowptr[part[r + 1]]].astype(np.int64)
        cl, halo, counts = gdist.halo_plan_host(part, r, cg)
        cl2, halo2, counts2 = gdist.halo_plan_device(part, r, torch.tensor(cg))
        assert (cl == cl2.numpy()).all() and (halo == halo2.numpy()).all() and (counts == counts2).all()
