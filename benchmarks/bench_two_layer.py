#!/usr/bin/env python3
"""The reference's two-layer network (models/sp_gat.py:7-34: 8 heads x 8 with ELU, then one averaged
output head) as ONE training step at the ogbn-products shape: 2.45 M nodes, 64.3 M stored entries,
100 input features, 47 classes; masked softmax cross-entropy on a tenth of the nodes + L2, backward
through both layers (the second layer needs dX of its input, the first does not), Adam update.
The single-layer benchmark (bench.py) times the first layer alone; this shows the composed step.

    python benchmarks/bench_two_layer.py [--steps 10] [--classes 47]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--classes", type=int, default=47)
    ap.add_argument("--dropout", type=float, default=0.0)
    args = ap.parse_args()
    from gat_amd import CSRGraph, _lib, layers, synth
    from gat_amd.models import SpGAT
    dev = torch.device("cuda:0")
    n, e = synth.PRODUCTS_NODES, synth.PRODUCTS_EDGES
    dst, src = synth.rmat_graph(n, e, 4, dev)
    graph = CSRGraph.from_device_coo(dst, src, n, n)
    del dst, src
    x = synth.normal((1, n, 100), 5, dev)
    gen = torch.Generator(device=dev).manual_seed(9)
    y = torch.nn.functional.one_hot(torch.randint(0, args.classes, (n,), device=dev, generator=gen), args.classes).float()
    mask = torch.rand(n, device=dev, generator=gen) < 0.1
    store = layers.VariableStore()
    kw = dict(hid_units=[8], n_heads=[8, 1], activation=layers.elu, residual=False)

    def forward():
        with layers.variable_scope(store):
            return SpGAT.inference(x, args.classes, n, True, args.dropout, args.dropout, bias_mat=graph, **kw)[0]

    forward()
    opt, l2 = SpGAT.training(store.parameters(), lr=0.005, l2_coef=0.0005)
    loss_out = torch.zeros((), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = SpGAT.masked_softmax_cross_entropy(forward(), y, mask) + l2()
        loss.backward()
        opt.step()
        loss_out.copy_(loss.detach())

    for _ in range(args.warmup):
        step()
    first = float(loss_out)
    records = {}
    raw = _lib.call

    def timed(name, *a):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        raw(name, *a)
        t1.record()
        records.setdefault(name, []).append((t0, t1))
    _lib.call = timed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    _lib.call = raw
    kern = {k: round(sum(a.elapsed_time(b) for a, b in v) / args.steps, 3) for k, v in sorted(records.items())}
    print(json.dumps({
        "workload": "two-layer SpGAT training step (8x8 ELU, then 1 x %d averaged), R-MAT %d nodes / %d stored entries, 100 features, "
                    "masked CE + L2 + Adam, dropout %.1f" % (args.classes, n, graph.nnz, args.dropout),
        "ms_per_step": round(ms, 3), "edges_per_s_both_layers": 2 * graph.nnz / (ms * 1e-3),
        "entry_points_ms": kern, "entry_points_total_ms": round(sum(kern.values()), 3),
        "loss_first": first, "loss_last": float(loss_out)}))


if __name__ == "__main__":
    main()
