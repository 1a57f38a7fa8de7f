#!/usr/bin/env python3
"""Secondary measurement: BASELINE.json configs[0] and configs[1] — the two-layer Cora network
(2708 nodes, 1433 features, 8 heads x 8 hidden, 7 classes; execute_cora.py:13-22) through
GAT.inference (dense bias_mat) and SpGAT.inference (sparse adjacency), forward + backward, eager
and replayed from a HIP graph.  The graph is the reference's own random stand-in for Cora
(utils/process.py:98-123 style: symmetric, density 0.002, plus self-loops).  The CPU figure for the
same network is printed by `python bench.py --workload cora`."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, iters, sync):
    for _ in range(3):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    sync()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    from gat_amd import CSRGraph, graphs, layers, preprocess
    from gat_amd.features import SparseFeatures
    from gat_amd.models import GAT, SpGAT
    dev = torch.device("cuda:0")
    n, f, c = 2708, 1433, 7
    rng = np.random.RandomState(0)
    adj = sp.random(n, n, density=0.001, random_state=rng, format="csr")
    adj = ((adj + adj.T) > 0).astype(np.float64)
    feats, feats_tuple = preprocess.preprocess_features(sp.random(n, f, density=0.0127, random_state=rng, format="csr"))
    triple = preprocess.preprocess_adj_bias(adj)
    graph = CSRGraph.from_reference_triple(triple, dev)
    bias = torch.tensor(preprocess.adj_to_bias(np.asarray(adj.todense())[None], [n], 1), dtype=torch.float32, device=dev)
    x = torch.tensor(np.asarray(feats)[None], dtype=torch.float32, device=dev)
    xs = SparseFeatures.from_tuple(feats_tuple, dev)          # the same features, stored entries only
    kw = dict(hid_units=[8], n_heads=[8, 1], activation=layers.elu, residual=False)
    res = {"nodes": n, "entries": graph.nnz}
    res["feature_entries"] = xs.nnz
    for name, model, g, inp in (("sparse", SpGAT, graph, x), ("sparse_spfeat", SpGAT, graph, xs), ("dense", GAT, bias, x)):
        store = layers.VariableStore()

        def step():
            with layers.variable_scope(store):
                out = model.inference(inp, c, n, False, 0.0, 0.0, bias_mat=g, **kw)
            for p in store.parameters():
                p.grad = None
            out.sum().backward()
        res[name + "_gpu_ms"] = round(timed(step, 20, torch.cuda.synchronize), 4)
        # the same step captured once into a HIP graph and replayed (the eager step is launch-bound)
        replay, _ = graphs.capture(step)
        res[name + "_gpu_graph_ms"] = round(timed(replay, 50, torch.cuda.synchronize), 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
