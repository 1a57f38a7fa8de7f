#!/usr/bin/env python3
"""Secondary measurement: BASELINE.json configs[2] — batched dense small graphs (B x 264 nodes,
8 heads x 8, 64 input features), forward + backward of one layer through DenseGATLayerFn.
Prints one JSON line (pair-slots/s = B*N^2 / t, and the HBM fraction from the survey's
0.96 MB/graph byte model)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=264)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--mask-path", choices=["auto", "dense", "lists", "csr", "prebuilt"], default="auto",
                    help="dense: matrix-core kernels over all N x N entries; lists: compacted lists of the contributing entries; "
                         "csr: the mask converted once (untimed) to a block-diagonal CSRGraph, CSR kernels; prebuilt: the mask's "
                         "lists built once (untimed, MaskLists: what the layer's memo does on the second unchanged sighting of a "
                         "mask), graph-resident kernels that only walk them")
    ap.add_argument("--dropout", type=float, default=0.0, help="coefficient and feature dropout rate (the reference trains with 0.6 / 0.6)")
    ap.add_argument("--density", type=float, default=0.10, help="fraction of node pairs that are edges (symmetric)")
    a = ap.parse_args()
    from gat_amd import _lib, ops, synth
    dev = torch.device("cuda:0")
    B, N, F, H, D = a.batch, a.nodes, 64, 8, 8
    g = torch.Generator(device=dev).manual_seed(2)
    adj = torch.rand(B, N, N, device=dev, generator=g) < 1.0 - (1.0 - a.density) ** 0.5
    adj = adj | adj.transpose(1, 2) | torch.eye(N, dtype=torch.bool, device=dev)[None]
    bias_mat = torch.where(adj, 0.0, -1e9).to(torch.float32)
    del adj
    x = synth.normal((B, N, F), 3, dev)
    p = synth.glorot_layer(F, H, D, 7, dev)
    for v in p.values():
        v.requires_grad_(not a.fwd_only)
    dout = synth.normal((B, N, H * D), 8, dev)
    density = float((bias_mat[:64] > -1e8).float().mean())
    graph = None
    drop = dict(coef_drop=a.dropout, in_drop=a.dropout, training=a.dropout > 0, seed=1234)
    mask = bias_mat
    convert_ms = None
    if a.mask_path == "csr":
        from gat_amd import CSRGraph
        t0 = time.perf_counter()
        graph = CSRGraph.from_bias_mat(bias_mat)
        torch.cuda.synchronize()
        convert_ms = (time.perf_counter() - t0) * 1e3
        cfg = ops.make_cfg(H, D, "elu", **drop)
    elif a.mask_path == "prebuilt":
        from gat_amd.graph import MaskLists
        MaskLists(bias_mat)                                  # first build pays for torch's lazy initialisations
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mask = MaskLists(bias_mat)
        torch.cuda.synchronize()
        convert_ms = (time.perf_counter() - t0) * 1e3
        cfg = ops.make_cfg(H, D, "elu", **drop)
    else:
        cfg = ops.make_cfg(H, D, "elu", mask_path=a.mask_path, **drop)

    rec = {}
    raw = _lib.call

    def timed(name, *args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); raw(name, *args); e1.record()
        rec.setdefault(name, []).append((e0, e1))

    def step():
        for v in p.values():
            v.grad = None
        if graph is not None:
            out = ops.sparse_gat_layer(x.reshape(B * N, F), p["W"], p["a1"], p["a2"], p["b1"], p["b2"], p["bias"], graph, cfg).reshape(B, N, H * D)
        else:
            out = ops.dense_gat_layer(x, p["W"], p["a1"], p["a2"], p["b1"], p["b2"], p["bias"], mask, cfg)
        if not a.fwd_only:
            out.backward(dout)
    for _ in range(a.warmup):
        step()
    _lib.call = timed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    _lib.call = raw
    alg = B * (2 * N * N * 4 + 2 * N * F * 4 + 4 * N * H * D * 4)
    print(json.dumps({"workload": "dense batched graphs B=%d N=%d F=%d %dx%d %s" % (B, N, F, H, D, "fwd" if a.fwd_only else "fwd+bwd"),
                      "mask_path": a.mask_path, "mask_density": round(density, 4), "dropout": a.dropout, "mask_conversion_ms_once": round(convert_ms, 2) if convert_ms is not None else None, "ms_per_step": round(ms, 4), "pair_slots_per_s": B * N * N / (ms * 1e-3),
                      "hbm_frac_survey_model": round(alg / (ms * 1e-3) / 8e12, 4),
                      "kernels_ms": {k: round(sum(x0.elapsed_time(x1) for x0, x1 in v) / len(v), 4) for k, v in sorted(rec.items())}}))


if __name__ == "__main__":
    main()
