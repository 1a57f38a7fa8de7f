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


def _hub_graph(n, world, seed):
    """A graph for the plan at the target world size (even blocks of n / world rows): one hub row in rank 0's
    block read by every node of every other rank (its gradient comes back from all world - 1 peers), ranks
    1 .. world-4 with random sources among blocks 0 .. world-4, rank world-3 reading nothing but the hub (empty
    lists for every other peer), rank world-2 reading the last block only, the last rank reading block 1 only."""
    rng = np.random.default_rng(seed)
    blk = n // world
    hub = 3
    dst, src = [np.arange(n)], [np.arange(n)]
    dst.append(np.arange(blk, n)); src.append(np.full(n - blk, hub))
    lo_n = (world - 3) * blk
    m = 4 * lo_n
    dst.append(rng.integers(0, lo_n, m)); src.append(rng.integers(0, lo_n, m))
    r6 = np.arange((world - 2) * blk, (world - 1) * blk)
    dst.append(np.repeat(r6, 3)); src.append(rng.integers((world - 1) * blk, n, 3 * blk))
    r7 = np.arange((world - 1) * blk, n)
    dst.append(np.repeat(r7, 2)); src.append(rng.integers(blk, 2 * blk, 2 * blk))
    key = np.unique(np.concatenate(dst).astype(np.int64) * n + np.concatenate(src))
    return (key // n).astype(np.int32), (key % n).astype(np.int32)


def _worker(rank, world, port, n, seed, kind, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        from gat_amd import dist as gdist, graph
        from oracle import ref_numpy, ref_torch
        from tests.util import layer_params, random_graph
        gdist._pack_rows, gdist._unpack_sum_rows = _cpu_pack, _cpu_unpack_sum

        F, H, D = 7, 2, 4
        if kind == "hub":
            dst, src = _hub_graph(n, world, seed)
            rowptr, col, _ = graph.csr_from_coo_host(n, dst, src)
            part = gdist.partition_rows(n, world)
            assert (np.diff(part) == n // world).all()
        else:
            dst, src = random_graph(n, 4.0, seed, hubs=2, hub_deg=n // 2)
            rowptr, col, _ = graph.csr_from_coo_host(n, dst, src)
            part = gdist.partition_rows(n, world, rowptr, balance=1)
        lo, hi = int(part[rank]), int(part[rank + 1])
        cg = col[rowptr[lo]:rowptr[hi]].astype(np.int64)
        cl, halo, counts = gdist.halo_plan_host(part, rank, cg)
        ex = gdist.HaloExchange(rank, world, hi - lo, halo, counts, lo, "cpu")
        assert sum(ex.recv_counts) == len(halo)
        if kind == "hub":
            blk = n // world
            if rank == 0:          # the hub row goes to every peer and comes back from every peer
                assert all(c >= 1 for c in ex.send_counts[1:])
                rows, rptr, _ = ex.unpack_plan
                k = int((rows == 3).nonzero()[0])
                assert int(rptr[k + 1] - rptr[k]) == world - 1
                assert int(ex.row_ptr_all[4] - ex.row_ptr_all[3]) == world - 1
            if rank == world - 3:  # reads the hub and nothing else: every other pair is empty in both directions' counts
                assert ex.recv_counts == [1] + [0] * (world - 1) and list(halo) == [3]
            if rank == world - 2:
                assert [c > 0 for c in ex.recv_counts] == [True] + [False] * (world - 2) + [True]
            if rank == world - 1:
                assert ex.recv_counts[0] == 1 and ex.recv_counts[1] > 0 and sum(ex.recv_counts[2:]) == 0
                assert ex.send_counts[world - 2] > 0 and sum(ex.send_counts[:world - 2]) == 0

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

        # the same two exchanges as world - 1 pairwise steps (the pipelined form of the layer)
        ids2 = torch.zeros_like(ids)
        ids2[: hi - lo] = ids[: hi - lo]
        ex.gather_halo_pairwise([(ids2, 1)]).wait()
        assert torch.equal(ids2, ids)
        cnt2 = torch.zeros((hi - lo + len(halo), 1), dtype=torch.float64)
        swept = []

        def sweep_owner(p_):                     # the layer's per-owner sweep: here it just fills that owner's halo rows
            swept.append(p_)
            cnt2[hi - lo + ex.recv_off[p_]: hi - lo + ex.recv_off[p_ + 1]] = 1.0
        pend = ex.scatter_add_halo_pairwise([(cnt2, 1)], sweep_owner)
        ex.scatter_add_halo_finish(pend, [(cnt2, 1)])
        assert sorted(swept) == [r for r in range(world) if r != rank] and swept[0] == (rank + 1) % world
        assert torch.equal(cnt2[: hi - lo], cnt[: hi - lo])

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
    run_ranks(_worker, world, (n, seed, "random"), timeout=240)


def test_partitioned_layer_at_world_eight_with_empty_pairs_and_a_row_every_peer_returns():
    """The plan at the target world size: peer pairs that exchange nothing, a rank whose whole halo is one row,
    and a row whose gradient comes back from all seven peers (fixed-order sum at the owner)."""
    from tests.util import run_ranks
    run_ranks(_worker, 8, (384, 5, "hub"), timeout=400)


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
