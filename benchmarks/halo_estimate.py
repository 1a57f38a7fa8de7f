"""One products-sized shard (or, with a leading `rmat2b` argument, one 12.5 M-row / 262.5 M-entry shard) of a world-times larger R-MAT graph, seen from rank 0 and run alone:
  * unique remote source rows (halo): what each rank receives per forward exchange and returns per
    backward exchange in halo="exchange" mode, and what it stores once in halo="replicate" mode;
  * the compute time of the step in replicate mode (owned + halo rows projected locally), measured
    in this one process with random rows standing in for the peers' input rows.  This is rank 0's
    step without the all-reduce of the parameter gradients (a few KB);
  * the step in exchange mode with the peers emulated: every all-to-all is a device-to-device copy
    of the same size on the communication stream (as many rows sent as received, random send list).
    That prices everything of the exchange step except the links: pack, the two-phase forward sweep
    (park and resume of each row's state), the split backward sweep and the one-launch unpack, with
    and without the stream overlap."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gat_amd import CSRGraph, ops, synth
from gat_amd import dist as gdist
from gat_amd.layers import GATHeads


class _NoPeers:
    """Stand-in for HaloExchange when the peers are not running: nothing to send or receive."""
    def __init__(self, n_own, n_halo):
        self.n_own, self.n_halo, self.n_send, self.world = n_own, n_halo, 0, 1
        self.recv_off = [0, n_halo]

    def gather_halo(self, table, width):
        pass

    def scatter_add_halo(self, table, width):
        pass


class _LoopbackPeers(gdist.HaloExchange):
    """HaloExchange whose peers are emulated on this GPU: as many rows leave for a peer as arrive from it
    (the real per-owner counts of this shard), and every transfer is a device-to-device copy of that size
    on the communication stream."""

    def __init__(self, n_own, n_halo, device, counts):
        self.rank, self.world, self.group, self.backend = 0, len(counts), None, "loopback"
        self.n_own, self.n_halo, self.device = n_own, n_halo, torch.device(device)
        self.recv_counts, self.send_counts = [int(c) for c in counts], [int(c) for c in counts]
        self._pool = torch.empty(max(int(max(counts)), 1) * 80, dtype=torch.float32, device=device)     # a peer's rows, either direction
        gen = torch.Generator(device=device)
        gen.manual_seed(11)
        # what each peer reads here: a random subset of the owned rows of the size it sends, in ascending order, as the
        # real plan lists them (a peer's halo ids are sorted): the pack gathers and the returned rows are sorted streams
        parts = []
        for c in self.send_counts:
            c = int(c)
            reps = -(-c // max(n_own, 1))
            ids = torch.cat([torch.randperm(n_own, device=device, generator=gen) for _ in range(max(reps, 1))])[:c]
            parts.append(torch.sort(ids).values)
        self.send_idx = torch.cat(parts).contiguous()
        self._finish_setup()

    def _a2a(self, out, inp, out_splits, in_splits):
        out.copy_(inp)

    def _sendrecv(self, sends, to, recvs, frm):
        for t in sends:
            if t.numel():
                self._pool.view(t.dtype)[:t.numel()].copy_(t.reshape(-1))
        for t in recvs:
            if t.numel():
                t.copy_(self._pool.view(t.dtype)[:t.numel()].view(t.shape))


def _kernel_breakdown(step, iters=5):
    """Milliseconds per step of every C-ABI entry point (HIP events around each call, all launches of a step added)."""
    from gat_amd import _lib
    records = {}
    raw = _lib.call

    def timed(name, *args):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        raw(name, *args)
        b.record()
        records.setdefault(name, []).append((a, b))
    _lib.call = timed
    try:
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
    finally:
        _lib.call = raw
    return {k.replace("gatmi_", ""): round(sum(a.elapsed_time(b) for a, b in v) / iters, 3) for k, v in sorted(records.items())}


def main():
    dev = torch.device("cuda:0")
    n_own, e_own = synth.PRODUCTS_NODES, synth.PRODUCTS_EDGES
    argv = sys.argv[1:]
    if argv and argv[0] == "rmat2b":                     # one GPU's share of BASELINE.json configs[4] instead of the products shape
        n_own, e_own, argv = 12_500_000, 250_000_000, argv[1:]
    F, H, D = 100, 8, 8
    for world in ([int(a) for a in argv] or (1, 2, 4, 8)):
        n = n_own * world
        part = np.arange(world + 1, dtype=np.int64) * n_own
        dst, src = synth.rmat_graph(n, e_own, 4, dev, row_range=(0, n_own))
        cl, halo, counts = gdist.halo_plan_device(part, 0, src)
        remote = float((src >= n_own).sum()) / src.numel()
        n_halo = int(halo.numel())
        # What replicating hub rows on every rank could save: halo rows by the number of this shard's entries that read
        # them.  The bytes of an exchange are proportional to ROWS; hubs carry the entries, not the rows.
        refs = torch.bincount(cl[cl >= n_own].to(torch.int64) - n_own, minlength=n_halo)
        n_remote = int(refs.sum())
        hub_table = {}
        for t in (2, 4, 16, 64, 256, 1024):
            sel = refs >= t
            hub_table["refs>=%d" % t] = {"halo_rows": int(sel.sum()), "share_of_halo_rows": round(float(sel.sum()) / max(1, n_halo), 4),
                                          "share_of_remote_entries": round(float(refs[sel].sum()) / max(1, n_remote), 4)}
        del refs
        g = CSRGraph.from_device_coo(dst, cl.to(torch.int64), n_own, n_own + n_halo, chunk=0)
        del dst, src, cl
        pg = gdist.PartitionedGraph(g, _NoPeers(n_own, n_halo), part, 0)
        x = synth.normal((n_own + n_halo, F), 5, dev)
        dout = synth.normal((n_own, H * D), 8, dev)
        m = GATHeads(F, H, D, device=dev)
        m.load(synth.glorot_layer(F, H, D, 7, dev))
        params = list(m.parameters())
        cfg = ops.make_cfg(H, D, "elu", "concat")

        def step():
            for p in params:
                p.grad = None
            gdist.dist_sparse_gat_layer(x, m.W, m.a1, m.a2, m.b1, m.b2, m.bias, pg, cfg, "replicate").backward(dout)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        line = {"world": world, "entries": g.nnz, "remote_entry_frac": round(remote, 3), "n_halo": n_halo, "halo_rows_by_reference_count": hub_table,
                "exchange_MB_forward_f32": round(n_halo * 256 / 1e6), "exchange_MB_backward_f32": round(n_halo * 288 / 1e6), "replicated_input_MB": round(n_halo * F * 4 / 1e6),
                "replicate_step_ms": round(ms, 3), "edges_per_s_per_gpu": round(g.nnz / ms * 1e3)}
        if world > 1:
            pe = gdist.PartitionedGraph(g, _LoopbackPeers(n_own, n_halo, dev, counts), part, 0)
            xo = x[:n_own].contiguous()
            for overlap in ("two_phase", "pipeline", False):
                def estep():
                    for p in params:
                        p.grad = None
                    gdist.dist_sparse_gat_layer(xo, m.W, m.a1, m.a2, m.b1, m.b2, m.bias, pe, cfg, "exchange", overlap).backward(dout)
                for _ in range(3):
                    estep()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    estep()
                torch.cuda.synchronize()
                line["exchange_step_ms_links_emulated_%s" % (overlap or "sequential")] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
                if overlap != "two_phase":
                    line["exchange_%s_kernels_ms" % (overlap or "sequential")] = _kernel_breakdown(estep)
            line["two_phase_items"] = [len(pe.items_a), len(pe.items_b), len(pe.split_ab), pe.n_slots_ab]
            line["local_entry_frac"] = round(pe.local_entries / g.nnz, 3)
            del pe, xo
        print(json.dumps(line), flush=True)
        del g, pg, x, dout, m


if __name__ == "__main__":
    main()
