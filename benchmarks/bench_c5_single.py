#!/usr/bin/env python3
"""BASELINE.json configs[4] WHOLE on one MI355X: the 100 M-node / 2 B-edge R-MAT graph (2.1 G stored
entries with the self-loops), 100 features, 8 heads x 8, forward + backward of one layer (first
layer: no dX) — the one-GPU denominator of the eight-GPU scaling target.

The graph is the union of the eight shards `bench.py --gpus 8 --workload rmat2b` gives its ranks
(same generator, same seeds, same scrambled source ids), built shard by shard.  The autograd wrapper
would hold about 300 GB here, so the step goes through the C ABI with every buffer allocated once
and reused where the data flow allows (the caller owns all buffers of libgatmi):

    out      doubles as dOut: the step differentiates loss = 1/2 |out|^2, whose dOut IS out
    agg      is dead after gatmi_gat_bwd_node and receives dWh
    q        likewise receives df2

which leaves x 40 + Wh 25.6 + out 25.6 + agg 25.6 + pos 25.6 + records 38.4 + graph 29 + small = about 225 GB
resident.  Prints one JSON line.  --dump writes sampled rows of the forward pass with everything they depend on;
tests/check_sampled_rows.py compares them with the CPU oracle.

    python benchmarks/bench_c5_single.py [--shards 8] [--steps 3] [--warmup 1] [--dump /tmp/c5_rows.npz]
    python tests/check_sampled_rows.py /tmp/c5_rows.npz
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NODES_PER_SHARD = 12_500_000
EDGES_PER_SHARD = 250_000_000
F_IN, H, D = 100, 8, 8
HD = H * D


def say(*a):
    print("[c5]", *a, file=sys.stderr, flush=True)


def gib():
    free, total = torch.cuda.mem_get_info()
    return "%.0f GiB in use" % ((total - free) / 2 ** 30)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=8, help="how many of the eight shards to join (8 = configs[4])")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--dump", default="", help="write the sampled rows and their inputs here (.npz) for tests/check_sampled_rows.py")
    args = ap.parse_args()
    from gat_amd import CSRGraph, _lib, ops, synth
    from gat_amd._lib import ptr
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    P = args.shards
    n = NODES_PER_SHARD * P
    t0 = time.time()
    parts = []
    for r in range(P):
        dst, src = synth.rmat_graph(n, EDGES_PER_SHARD, 4 + 1000 * r, dev, row_range=(r * NODES_PER_SHARD, (r + 1) * NODES_PER_SHARD))
        key = dst * n + src
        del dst, src
        parts.append(torch.sort(key)[0])                      # shards hold disjoint ascending row ranges
        del key
        torch.cuda.empty_cache()
        say("shard %d generated, %s, %.0f s" % (r, gib(), time.time() - t0))
    key = torch.cat(parts)
    del parts
    torch.cuda.empty_cache()
    nnz = key.numel()
    if nnz >= 2 ** 31:
        raise SystemExit("%d entries do not fit 32-bit offsets" % nnz)
    # CSR / CSC on the device, as CSRGraph.from_device_coo does, minus its global sort (the keys are sorted already)
    d = torch.div(key, n, rounding_mode="floor")
    col = (key - d * n).to(torch.int32)
    del key
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(d, minlength=n), 0)
    torch.cuda.empty_cache()
    say("CSR done, %s" % gib())
    _, order = torch.sort(col.to(torch.int64), stable=True)
    rowidx = d[order].to(torch.int32)
    del d
    perm = order.to(torch.int32)
    del order
    colptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    colptr[1:] = torch.cumsum(torch.bincount(col.to(torch.int64), minlength=n), 0)
    torch.cuda.empty_cache()
    say("CSC done, %s, %.0f s" % (gib(), time.time() - t0))
    g = CSRGraph(n, n, rowptr.cpu().numpy(), col, colptr.cpu().numpy(), rowidx, perm, dev, 0)
    del rowptr, colptr
    torch.cuda.empty_cache()
    say("graph: %r, %s, %.0f s" % (g, gib(), time.time() - t0))

    p = synth.glorot_layer(F_IN, H, D, 7, dev)
    cfg = ops.make_cfg(H, D, "elu")
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    x = new(n, F_IN)
    for r in range(P):                                        # each shard's rows as its rank would draw them
        x[r * NODES_PER_SHARD:(r + 1) * NODES_PER_SHARD] = synth.normal((NODES_PER_SHARD, F_IN), 5 + 1000 * r, dev)
    W = p["W"].reshape(F_IN, HD).contiguous()
    a1, a2 = p["a1"].reshape(HD).contiguous(), p["a2"].reshape(HD).contiguous()
    b1, b2, bias = p["b1"].contiguous(), p["b2"].contiguous(), p["bias"].reshape(HD).contiguous()
    ldr = ops._record_width(H, D)
    Wh, f1 = new(n, HD), new(n, H)
    out, agg, pos, lse, q = new(n, HD), new(n, HD), new(n, HD), new(n, H), new(n, H)
    rec, df1 = new(n, ldr), new(n, H)
    dWh, df2 = agg, q                                         # dead after gatmi_gat_bwd_node
    f2 = df1                                                  # the projection writes the f2 table; nothing reads it (the
                                                              # gather kernels form f2 from the rows), df1 is written later
    ws_f = new(_lib.lib.gatmi_gat_fwd_ws_floats(g.n_slots_csr, H, D)) if len(g.split_csr) else None
    ws_n = new(_lib.lib.gatmi_gat_bwd_node_ws_floats(H, D))
    ws_b = new(_lib.lib.gatmi_gat_bwd_ws_floats(g.n_slots_csc, H, D)) if len(g.split_csc) else None
    ws_p = new(_lib.lib.gatmi_proj_bwd_ws_floats(F_IN, H, D))
    dW, da1, da2, db1, db2, dbias = new(F_IN, HD), new(HD), new(HD), new(H), new(H), new(HD)
    say("buffers allocated, %s" % gib())
    st = _lib.current_stream()

    def forward():
        _lib.call("gatmi_proj_fwd", ptr(x), n, F_IN, F_IN, ptr(W), ptr(a1), ptr(a2), ptr(b1), ptr(b2), H, D,
                  ptr(Wh), None, 0, ptr(f1), ptr(f2), None, None, None, 1.0, 0, st)
        _lib.call("gatmi_gat_fwd", n, H, D, ptr(g.items_csr), len(g.items_csr), ptr(g.split_csr) if len(g.split_csr) else None,
                  len(g.split_csr), ptr(g.col), ptr(Wh), _lib.DTYPE_F32, HD, ptr(f1), None, ptr(a2), ptr(b2), ptr(bias), None,
                  _lib.ACT_ELU, _lib.REDUCE_CONCAT, cfg.slope, 1.0, 1.0, 0, ptr(out), ptr(agg), ptr(pos), ptr(lse), ptr(q), ptr(ws_f), st)

    def backward():
        _lib.call("gatmi_gat_bwd_node", n, H, D, ptr(out), None, ptr(agg), ptr(pos), ptr(q), ptr(f1), ptr(lse), ptr(bias),
                  _lib.ACT_ELU, _lib.REDUCE_CONCAT, cfg.slope, ptr(rec), ldr, None, 0, ptr(rec[:, HD:]), ldr, ptr(df1), ptr(dbias), ptr(ws_n), st)
        _lib.call("gatmi_gat_bwd_csc", n, H, D, ptr(g.items_csc), len(g.items_csc), ptr(g.split_csc) if len(g.split_csc) else None,
                  len(g.split_csc), ptr(g.rowidx), ptr(g.perm), ptr(rec), _lib.DTYPE_F32, ldr, ptr(rec[:, HD:]), ldr, ptr(Wh), None, ptr(a2), ptr(b2),
                  ptr(df1), ptr(a1), ptr(a2), cfg.slope, 1.0, 1.0, 0, ptr(dWh), ptr(df2), ptr(ws_b), st)
        _lib.call("gatmi_proj_bwd_params", ptr(x), n, F_IN, F_IN, ptr(dWh), ptr(Wh), ptr(df1), ptr(df2), H, D,
                  ptr(dW), ptr(da1), ptr(da2), ptr(db1), ptr(db2), ptr(ws_p), 1.0, 0, st)

    forward()
    torch.cuda.synchronize()
    say("first forward pass done")
    checked = None
    if not args.no_check:
        # (1) sampled rows of the forward pass (the two longest, the last rows of the entry array, random ones) with the
        # inputs they depend on, written for tests/check_sampled_rows.py, which evaluates the CPU oracle on the induced
        # sub-graph; (2) a whole-graph identity of the backward pass
        deg = np.diff(g.rowptr_host)
        rows = np.unique(np.concatenate([np.argsort(deg)[-2:], np.nonzero(deg > 1)[0][-20:],
                                         np.random.default_rng(2).choice(n, 150, replace=False)]))
        lo, lens = g.rowptr_host[rows], deg[rows]
        epos = np.repeat(lo - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(int(lens.sum()))
        cols = g.col[torch.as_tensor(epos, device=dev)].cpu().numpy().astype(np.int64)
        nodes = np.unique(np.concatenate([rows, cols]))
        sub_len = np.zeros(len(nodes), dtype=np.int64)
        sub_len[np.searchsorted(nodes, rows)] = lens
        if args.dump:
            np.savez(args.dump, x=x[torch.as_tensor(nodes, device=dev)].cpu().numpy(),
                     rowptr=np.concatenate([[0], np.cumsum(sub_len)]), col=np.searchsorted(nodes, cols).astype(np.int32),
                     rows=np.searchsorted(nodes, rows), got=out[torch.as_tensor(rows, device=dev)].cpu().numpy(),
                     **{"p_" + k: v.detach().cpu().numpy() for k, v in p.items()})
            say("sampled rows written to %s (%d rows, %d nodes, longest row %d)" % (args.dump, len(rows), len(nodes), deg.max()))
        backward()
        torch.cuda.synchronize()
        # sum_j df2_j and sum_i df1_i are both the sum of every entry's logit gradient (one comes from the per-source
        # sweep, the other from the forward pass's positive-part sums)
        s1, s2 = df1.double().sum(0), df2.double().sum(0)
        scale = df1.double().abs().sum(0) + df2.double().abs().sum(0)
        rel = float(((s1 - s2).abs() / scale).max())
        if not (rel <= 1e-6 and bool(torch.isfinite(dW).all())):
            raise SystemExit("backward identities are off: %.3g" % rel)
        checked = {"sampled_rows": int(len(rows)), "longest_row": int(deg.max()), "sum_df1_vs_sum_df2_rel": rel}
        say("checks passed:", checked)
        forward()

    def step():
        forward()
        backward()

    for _ in range(args.warmup):
        step()
    names = ["proj_fwd", "gat_fwd", "gat_bwd_node", "gat_bwd_csc", "proj_bwd_params"]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(args.steps)]
    raw = _lib.call
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k in range(args.steps):
        it = iter(ev[k])
        next(it).record()

        def call(name, *a, _it=it):
            raw(name, *a)
            next(_it).record()
        _lib.call = call
        step()
    _lib.call = raw
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t1) / args.steps
    kern = {nm: round(sum(e[i].elapsed_time(e[i + 1]) for e in ev) / args.steps, 3) for i, nm in enumerate(names)}
    print(json.dumps({
        "workload": "configs[4] on ONE GPU: R-MAT %d nodes, %d stored entries (self-loops included), %d features, %d heads x %d, "
                    "fwd+bwd of one layer through the C ABI with reused buffers" % (n, g.nnz, F_IN, H, D),
        "shards_joined": P, "ms_per_step": round(sec * 1e3, 2), "edges_per_s": g.nnz / sec, "kernels_ms": kern,
        "resident": gib(), "graph": repr(g), "checks": checked}))


if __name__ == "__main__":
    main()
