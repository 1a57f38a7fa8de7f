#!/usr/bin/env python3
"""Benchmark of the graph-attention hot path: one 8-head GAT layer, forward + backward.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (N = 1): BASELINE.json configs[3] — a synthetic power-law (R-MAT) graph at the
ogbn-products shape, 2,449,029 nodes, 61,859,140 directed edges + one self-loop per node,
100 input features, 8 heads x 8 hidden, ELU, fp32.  A step = forward + backward of the layer
(all parameter gradients; no dX, it is a first layer), inputs resident in HBM, dropout off.
For N > 1 every rank owns a products-sized block of destination rows of an N-times larger
R-MAT graph (weak scaling) and exchanges halo rows with RCCL all-to-all.

Prints ONE JSON line on rank 0.  `value` = directed edge entries (self-loops included)
processed per second over all ranks.  `roofline` prices the slowest kernel of the step against
the 8 TB/s HBM peak from algorithmic bytes; `layer_roofline` does the same for the whole step
with the survey's byte model (944 B/edge + 2984 B/node at these dimensions).
`cpu_baseline` times the CPU restatement of the reference math (oracle/ref_torch.py, labelled
"port": TensorFlow, which the reference runs on, is not installed) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0
F_IN, HEADS, WIDTH = 100, 8, 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=0, help="override node count per GPU (debug)")
    ap.add_argument("--edges", type=int, default=0, help="override edge count per GPU (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--workload", choices=["products", "cora"], default="products",
                    help="products: the headline (default); cora: the two-layer Cora network, GPU vs the CPU restatement")
    ap.add_argument("--halo", choices=["replicate", "exchange"], default="replicate",
                    help="N > 1 only.  replicate (default): the layer is a first layer, its constant input rows of the halo "
                         "sources are fetched once at set-up and projected locally, so a step communicates only the all-reduce "
                         "of the parameter gradients; exchange: Wh / f2 rows and their gradients travel every step (RCCL all-to-all)")
    ap.add_argument("--storage", choices=["f32", "bf16"], default="f32",
                    help="bf16: gathered tables in bf16 (2e-2 parity class); the headline is f32")
    return ap.parse_args()


def layer_alg_bytes(E, N, f_in=F_IN, H=HEADS, D=WIDTH, s=4):
    """Survey byte model of one layer fwd+bwd (gather-inclusive)."""
    per_edge = 3 * H * D * s + 3 * H * 4 + 2 * H * 4 + 16
    per_node = 8 * f_in + 7 * H * D * 4 + 12 * H * 4 + 8
    return E * per_edge + N * per_node


def kernel_alg_bytes(name, E, N, H=HEADS, D=WIDTH, f_in=F_IN, s=4):
    """Algorithmic bytes of one launch of each entry point (what the math must touch once);
    s = bytes per element of the gathered table (4 fp32, 2 bf16)."""
    hd = H * D * 4
    hg = H * D * s
    if name == "gatmi_gat_fwd":       # gather Wh row + f2 + col id per edge; f1, item, out per node
        return E * (hg + H * 4 + 4) + N * (hd + H * 4 + 16)
    if name == "gatmi_gat_bwd_csc":   # gather g row + {f1,lse,c} + row id per edge; Wh, f2, df1 in, dWh, df2 out per node
        return E * (hg + 3 * H * 4 + 4) + N * (2 * hd + 3 * H * 4 + 16)
    if name == "gatmi_gat_bwd_node":  # dOut, out, agg, pos in; g out; small per-head vectors
        return N * (5 * hd + 8 * H * 4)
    if name == "gatmi_proj_fwd":
        return N * (f_in * 4 + hd + 2 * H * 4)
    if name == "gatmi_proj_bwd_params":
        return N * (f_in * 4 + 2 * hd + 2 * H * 4)
    return 0


class EventProfiler:
    """HIP events (torch's current stream = the stream the kernels are enqueued on) around every
    C-ABI call; durations are read after the timed region."""

    def __init__(self):
        self.records = {}

    def wrap(self, call):
        def timed(name, *args):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            call(name, *args)
            b.record()
            self.records.setdefault(name, []).append((a, b))
        return timed

    def summary(self):
        out = {}
        for name, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = (sum(ms) / len(ms), len(ms))
        return out


def cpu_baseline(sample_nodes=272_114, sample_edges=6_873_238, iters=5):
    """The CPU restatement (oracle) on an R-MAT sample with the benchmark's average degree."""
    from oracle import ref_torch
    from gat_amd import synth
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    dst, src = synth.rmat_graph(sample_nodes, sample_edges, 4, torch.device("cpu"))
    order = torch.argsort(dst * sample_nodes + src)
    dst, src = dst[order], src[order]
    rowptr = torch.zeros(sample_nodes + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=sample_nodes), 0)
    p = synth.glorot_layer(F_IN, HEADS, WIDTH, 7, torch.device("cpu"))
    x = synth.normal((sample_nodes, F_IN), 5, torch.device("cpu"))
    dout = synth.normal((sample_nodes, HEADS * WIDTH), 8, torch.device("cpu"))
    for v in p.values():
        v.requires_grad_(True)
    best = None
    for _ in range(iters):
        t0 = time.perf_counter()
        out = ref_torch.layer_csr(x, p, rowptr, src.to(torch.int32), "elu", "concat")
        out.backward(dout)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        for v in p.values():
            v.grad = None
    E = int(dst.numel())
    return {"value": E / best, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": "R-MAT %d nodes / %d edge entries (benchmark's average degree), fp32 torch-CPU restatement "
                      "of utils/layers.py:6-34 on stored edges, best of %d fwd+bwd" % (sample_nodes, E, iters)}


def run_cora(args):
    """BASELINE.json configs[0]/[1]: the two-layer Cora network (2708 nodes, 1433 features, 8 x 8 hidden,
    7 classes; execute_cora.py:13-22) forward + backward on a Cora-shaped random graph: SpGAT.inference
    and GAT.inference on the GPU next to the CPU restatement of the dense path (the reference's
    TensorFlow CPU path cannot run here)."""
    import numpy as np
    import scipy.sparse as sp
    from gat_amd import CSRGraph, layers, preprocess
    from gat_amd.models import GAT, SpGAT
    from oracle import ref_torch
    dev = torch.device("cuda:0")
    n, f, c = 2708, 1433, 7
    rng = np.random.RandomState(0)
    adj = sp.random(n, n, density=0.001, random_state=rng, format="csr")
    adj = ((adj + adj.T) > 0).astype(np.float64)
    feats, feats_tuple = preprocess.preprocess_features(sp.random(n, f, density=0.0127, random_state=rng, format="csr"))
    graph = CSRGraph.from_reference_triple(preprocess.preprocess_adj_bias(adj), dev)
    bias = torch.tensor(preprocess.adj_to_bias(np.asarray(adj.todense())[None], [n], 1), dtype=torch.float32, device=dev)
    x = torch.tensor(np.asarray(feats)[None], dtype=torch.float32, device=dev)
    from gat_amd.features import SparseFeatures
    xs = SparseFeatures.from_tuple(feats_tuple, dev)
    kw = dict(hid_units=[8], n_heads=[8, 1], activation=layers.elu, residual=False)

    def timed(fn, iters, sync):
        for _ in range(args.warmup):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        sync()
        return (time.perf_counter() - t0) / iters * 1e3
    res = {}
    for name, model, g, inp in (("sparse", SpGAT, graph, x), ("sparse_features", SpGAT, graph, xs), ("dense", GAT, bias, x)):
        store = layers.VariableStore()

        def step():
            with layers.variable_scope(store):
                out = model.inference(inp, c, n, False, 0.0, 0.0, bias_mat=g, **kw)
            for p in store.parameters():
                p.grad = None
            out.sum().backward()
        res[name] = timed(step, args.steps, torch.cuda.synchronize)
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    lp = [{k: v.detach().cpu().requires_grad_(True) for k, v in m.named_parameters()} for m in store.layers]
    xc, bc = x.cpu(), bias.cpu()
    cpu_ms = timed(lambda: ref_torch.inference_dense(xc, lp, bc).sum().backward(), 3, lambda: None)
    print(json.dumps({
        "metric": "Cora two-layer GAT fwd+bwd steps/sec", "value": 1e3 / res["sparse"], "unit": "steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(res["sparse"], 4), "higher_is_better": True,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Cora-shaped random graph: %d nodes, %d stored entries, %d features, 8x8 + 1x7, SpGAT.inference" % (n, graph.nnz, f)},
        "sparse_features_ms_per_step": round(res["sparse_features"], 4), "dense_ms_per_step": round(res["dense"], 4),
        "cpu_baseline": {"value": 1e3 / cpu_ms, "unit": "steps/s", "cores": threads, "kind": "port",
                         "sample": "the same two-layer network through the torch-CPU fp32 restatement of the dense path, mean of 3 fwd+bwd"}}),
        flush=True)


def main():
    args = parse()
    if args.workload == "cora":
        return run_cora(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the graph-attention kernels have no CPU path")
    # GATMI_DIST_BACKEND=gloo rehearses N > 1 on a box with fewer GPUs than ranks (ranks share devices,
    # buffers are staged through the host); the measured configuration is always nccl = RCCL.
    backend = os.environ.get("GATMI_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import gat_amd
    from gat_amd import _lib, synth, CSRGraph
    from gat_amd.layers import GATHeads

    n_own = args.nodes or synth.PRODUCTS_NODES
    e_own = args.edges or synth.PRODUCTS_EDGES
    chunk = args.chunk

    if world > 1:
        import torch.distributed as dist
        from gat_amd import dist as gdist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        layer_step, E_rank, info = gdist.build_bench_step(rank, world, n_own, e_own, F_IN, HEADS, WIDTH, dev, chunk, args.storage, args.halo)
    else:
        dst, src = synth.rmat_graph(n_own, e_own, 4, dev)
        graph = CSRGraph.from_device_coo(dst, src, n_own, n_own, chunk=chunk)
        del dst, src
        E_rank = graph.nnz
        x = synth.normal((n_own, F_IN), 5, dev)
        dout = synth.normal((n_own, HEADS * WIDTH), 8, dev)
        heads = GATHeads(F_IN, HEADS, WIDTH, device=dev, storage=args.storage)
        heads.load(synth.glorot_layer(F_IN, HEADS, WIDTH, 7, dev))
        heads.train()
        params = list(heads.parameters())
        info = {"graph": repr(graph)}

        def layer_step():
            for p in params:
                p.grad = None
            out = heads(x, graph, "elu", "concat")
            out.backward(dout)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        layer_step()
    prof = EventProfiler()
    raw_call = _lib.call
    _lib.call = prof.wrap(raw_call)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        layer_step()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.call = raw_call

    t = torch.tensor([elapsed, float(E_rank)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = t.clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        elapsed, E_total = float(tmax[0]), float(t[1])
    else:
        E_total = float(E_rank)
    ms_per_step = elapsed / args.steps * 1e3

    if rank == 0:
        ksum = prof.summary()
        dom = max(ksum, key=lambda k: ksum[k][0]) if ksum else None
        roofline = None
        if dom:
            ab = kernel_alg_bytes(dom, E_rank, n_own, s=4 if args.storage == "f32" else 2)
            gbps = ab / (ksum[dom][0] * 1e-3) / 1e9
            traffic, traffic_src = None, None
            try:   # PMC-measured HBM bytes per launch, taken in separate rocprofv3 --pmc runs of this same workload
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                if world == 1 and args.storage == "f32" and tj.get("workload_edges") == E_rank:
                    traffic, traffic_src = tj["kernels"].get(dom), tj.get("source")
            except (OSError, ValueError, KeyError):
                pass
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "alg_bytes_per_launch": ab, "avg_ms": round(ksum[dom][0], 4)}
        lb = layer_alg_bytes(E_rank, n_own, s=4 if args.storage == "f32" else 2)
        lg = lb / (ms_per_step * 1e-3) / 1e9
        line = {
            "metric": "GAT layer fwd+bwd edges/sec (8 heads) + % HBM roofline at 1/2/4/8 GPUs",
            "value": E_total / (ms_per_step * 1e-3), "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.storage == "f32" else "f32 arithmetic, bf16 gather tables", "data": "synthetic",
            "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
            "config": {"workload": "R-MAT power-law graph at ogbn-products shape per GPU: %d nodes, %d edge entries "
                                   "(self-loops included), %d-d features, %d heads x %d, ELU, fwd+bwd of one layer"
                                   % (n_own, E_rank, F_IN, HEADS, WIDTH),
                       "partitioning": "single GPU" if world == 1 else (
                           "row-partitioned x%d, " % world + (
                               "halo sources' input rows replicated at set-up and projected locally; per step one RCCL "
                               "all-reduce of the parameter gradients" if args.halo == "replicate" else
                               "RCCL all-to-all of halo Wh/f2 rows (forward) and of their gradients (backward) every step"))},
            "roofline": roofline,
            "layer_roofline": {"bound": "hbm", "alg_bytes": lb, "achieved": round(lg, 1), "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": round(lg / HBM_PEAK_GBPS, 4)},
            "kernels_ms": {k: round(v[0], 4) for k, v in sorted(ksum.items())},
            "info": info,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
