#!/usr/bin/env python3
"""Benchmark of the graph-attention hot path: one 8-head GAT layer, forward + backward.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (N = 1): BASELINE.json configs[3] — a synthetic power-law (R-MAT) graph at the
ogbn-products shape, 2,449,029 nodes, 61,859,140 directed edges + one self-loop per node,
100 input features, 8 heads x 8 hidden, ELU, fp32.  A step = forward + backward of the layer
(all parameter gradients; no dX, it is a first layer), inputs resident in HBM, dropout off.
For N > 1 every rank owns a products-sized block of destination rows of an N-times larger
R-MAT graph (weak scaling) and exchanges halo rows with RCCL all-to-all (--halo exchange, the
default: Wh / f2 rows forward, their gradients backward, every step, overlapped with the sweeps that
do not need them).  --workload rmat2b gives every rank its share of BASELINE.json configs[4]
(100 M nodes / 2 B edges over 8 GPUs: 12.5 M rows and 250 M edges + self-loops per rank).

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
    ap.add_argument("--workload", choices=["products", "rmat2b", "cora"], default="products",
                    help="products: the headline (default), per GPU the ogbn-products shape; rmat2b: per GPU one eighth of "
                         "configs[4] (12.5 M rows, 250 M edges + self-loops: at --gpus 8 the 100 M-node / 2 B-edge graph); "
                         "cora: the two-layer Cora network, GPU vs the CPU restatement")
    ap.add_argument("--halo", choices=["exchange", "replicate", "both"], default="both",
                    help="N > 1 only.  exchange (the measured configuration, the line's value): Wh / f2 rows of the halo sources and "
                         "their gradients travel every step (RCCL all-to-all); replicate: a first layer's constant input rows "
                         "of the halo sources are fetched once at set-up and projected locally, a step communicates only the "
                         "all-reduce of the parameter gradients; both (default): exchange is the line's value, the replicate-mode step "
                         "is timed afterwards and reported beside it under 'replicate_mode'")
    ap.add_argument("--overlap", choices=["auto", "on", "off"], default="auto",
                    help="N > 1, exchange: run the all-to-alls on a communication stream beside the sweeps that do not need "
                         "them (on), or exchanges and kernels in turn (off).  The overlap costs some kernel time (every row's "
                         "softmax state is parked between the two sweeps) and hides at most the sweep over owned sources, so "
                         "which is faster depends on the share of owned sources and on the links: auto (default) times three "
                         "steps of each at set-up, outside the timed region, and keeps the faster")
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
    if name == "gatmi_gat_fwd":       # gather Wh row + col id per edge (f2 is formed from the row); f1, item, out per node
        return E * (hg + 4) + N * (hd + H * 4 + 16)
    if name == "gatmi_gat_bwd_csc":   # gather g row + {f1,lse,c} + row id per edge; Wh, f2, df1 in, dWh, df2 out per node
        return E * (hg + 3 * H * 4 + 4) + N * (2 * hd + 3 * H * 4 + 16)
    if name == "gatmi_gat_bwd_node":  # dOut, agg, pos in (the activation's derivative comes from agg + bias); g out; small per-head vectors
        return N * (4 * hd + 8 * H * 4)
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
        """name -> (mean milliseconds per call, number of calls)."""
        out = {}
        for name, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = (sum(ms) / len(ms), len(ms))
        return out


# kernel symbol each timed entry point launches at the benchmark's shape (8 heads x 8: 16 lanes per
# row, 2 per head; training, no dropout), as rocprofv3 prints it: a PMC traffic figure is only
# attached to the line when it was counted on a kernel of exactly this name
KERNEL_SYMBOL = {
    "gatmi_gat_fwd": {"f32": "gat_fwd_kernel<16, 2, true, false, false, true>", "bf16": "gat_fwd_kernel<16, 2, true, false, true, true>"},
    "gatmi_gat_bwd_csc": {"f32": "gat_bwd_csc_kernel<16, 2, false, false>", "bf16": "gat_bwd_csc_kernel<16, 2, false, true>"},
    "gatmi_gat_bwd_node": {"f32": "gat_bwd_node_kernel<16, 2>", "bf16": "gat_bwd_node_kernel<16, 2>"},
}


def pmc_traffic(entry, storage, E_rank):
    """(bytes per launch, source) from profiles/traffic.json if it was taken on this workload and on the
    kernel this run launched; (None, reason) otherwise."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None, "profiles/traffic.json missing"
    want = KERNEL_SYMBOL.get(entry, {}).get(storage)
    have = tj.get("kernel_symbols", {}).get(entry)
    if tj.get("workload_edges") != E_rank:
        return None, "traffic.json is for another workload"
    if want is None or have != want:
        return None, "traffic.json counted %r, this run launched %r" % (have, want)
    return tj["kernels"].get(entry), tj.get("source")


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


RMAT2B_NODES_PER_GPU = 12_500_000      # configs[4]: 100 M nodes / 2 B edges row-partitioned over 8 GPUs
RMAT2B_EDGES_PER_GPU = 250_000_000


def timed_steps(step, args, world, dev, lib):
    """W warm-up steps, then K steps between two barrier + synchronize brackets, every C-ABI call
    bracketed by HIP events.  Returns (seconds, max over ranks; per-entry-point summary)."""
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof = EventProfiler()
    raw_call = lib.call
    lib.call = prof.wrap(raw_call)
    try:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
    finally:
        lib.call = raw_call
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0])
    return elapsed, prof.summary()


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

    if args.workload == "rmat2b":
        n_own, e_own = RMAT2B_NODES_PER_GPU, RMAT2B_EDGES_PER_GPU
        shape_name = "one eighth of the 100 M-node / 2 B-edge R-MAT graph per GPU"
    else:
        n_own, e_own = synth.PRODUCTS_NODES, synth.PRODUCTS_EDGES
        shape_name = "R-MAT power-law graph at ogbn-products shape per GPU"
    n_own = args.nodes or n_own
    e_own = args.edges or e_own
    chunk = args.chunk
    storage = args.storage
    dist_info = None
    replicate_line = None

    if world > 1:
        import torch.distributed as dist
        from gat_amd import dist as gdist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        bench = gdist.BenchShard(rank, world, n_own, e_own, F_IN, HEADS, WIDTH, dev, chunk, storage)
        E_rank = bench.E_rank
        info = bench.info()
        modes = ["exchange", "replicate"] if args.halo == "both" else [args.halo]
        results = {}
        overlap, overlap_trial = args.overlap == "on", None
        if args.overlap == "auto" and "exchange" in modes:
            trial = argparse.Namespace(warmup=2, steps=3)
            overlap_trial = {}
            for ov in (True, False):
                t_ov, _ = timed_steps(bench.step_fn("exchange", overlap=ov), trial, world, dev, _lib)     # max over ranks: same choice everywhere
                overlap_trial["on" if ov else "off"] = round(t_ov / trial.steps * 1e3, 4)
            overlap = overlap_trial["on"] <= overlap_trial["off"]
        replicate_error = None
        for mode in modes:
            try:
                step = bench.step_fn(mode, overlap=overlap)
                bench.pg.exchange.enable_timing()
                elapsed, ksum = timed_steps(step, args, world, dev, _lib)
                torch.cuda.synchronize()
                a2a_total_ms, sent, received, n_a2a = bench.pg.exchange.a2a_summary()
                per_step = args.steps + args.warmup          # the exchange's events cover the warm-up steps too
                results[mode] = (elapsed, ksum, {
                    "a2a_ms": round(a2a_total_ms / per_step, 4), "a2a_per_step": n_a2a // per_step,
                    "bytes_sent_per_step": sent // per_step, "bytes_received_per_step": received // per_step})
                bench.release(mode)
            except Exception as exc:          # the second leg of --halo both must not cost the line its measured value
                if mode == modes[0]:
                    raise
                replicate_error = "%s: %s" % (type(exc).__name__, exc)
        elapsed, ksum, comm = results[modes[0]]
        dist_info = dict(comm, backend=("rccl" if backend == "nccl" else backend), world_size=dist.get_world_size(),
                         halo=modes[0], overlap=(modes[0] == "exchange" and overlap), overlap_trial_ms=overlap_trial,
                         n_halo=bench.pg.n_halo, n_send=bench.pg.exchange.n_send,
                         local_entry_fraction=round(bench.pg.local_entries / max(1, E_rank), 4))
        if len(modes) == 2 and "replicate" in results:
            e2, k2, c2 = results["replicate"]
            replicate_line = (e2, k2, c2)
    else:
        dst, src = synth.rmat_graph(n_own, e_own, 4, dev)
        graph = CSRGraph.from_device_coo(dst, src, n_own, n_own, chunk=chunk)
        del dst, src
        E_rank = graph.nnz
        x = synth.normal((n_own, F_IN), 5, dev)
        dout = synth.normal((n_own, HEADS * WIDTH), 8, dev)
        heads = GATHeads(F_IN, HEADS, WIDTH, device=dev, storage=storage)
        heads.load(synth.glorot_layer(F_IN, HEADS, WIDTH, 7, dev))
        heads.train()
        params = list(heads.parameters())
        info = {"graph": repr(graph)}

        def layer_step():
            for p in params:
                p.grad = None
            out = heads(x, graph, "elu", "concat")
            out.backward(dout)
        elapsed, ksum = timed_steps(layer_step, args, world, dev, _lib)

    E_total = float(E_rank)
    if world > 1:
        t = torch.tensor([float(E_rank)], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        E_total = float(t[0])
    ms_per_step = elapsed / args.steps * 1e3

    if rank == 0:
        s_bytes = 4 if storage == "f32" else 2
        per_step = {k: v[0] * v[1] / args.steps for k, v in ksum.items()}       # ms per step of each entry point (all its launches)
        dom = max(per_step, key=per_step.get) if per_step else None
        roofline = None
        if dom:
            ab = kernel_alg_bytes(dom, E_rank, n_own, s=s_bytes)
            gbps = ab / (per_step[dom] * 1e-3) / 1e9
            traffic, traffic_src = (None, "single-GPU figure only") if world > 1 else pmc_traffic(dom, storage, E_rank)
            roofline = {"bound": "hbm", "kernel": dom, "kernel_symbol": KERNEL_SYMBOL.get(dom, {}).get(storage),
                        "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "alg_bytes_per_launch": ab, "avg_ms": round(per_step[dom], 4),
                        "launches_per_step": ksum[dom][1] // args.steps,
                        "note": "achieved = algorithmic bytes / time: every entry is charged its full gather, so rows gathered "
                                "many times (hubs) that L2 / Infinity Cache serve push it past what HBM itself delivers; "
                                "traffic_* prices the bytes the counters saw leave L2"}
            if traffic:
                tg = traffic / (per_step[dom] * 1e-3) / 1e9
                roofline["traffic_achieved"] = round(tg, 1)
                roofline["traffic_frac"] = round(tg / HBM_PEAK_GBPS, 4)
        lb = layer_alg_bytes(E_rank, n_own, s=s_bytes)
        lg = lb / (ms_per_step * 1e-3) / 1e9
        # compulsory lower bound: every array touched once, no gather counted (E index entries of 16 B + the per-node bytes)
        b_min = E_rank * 16 + n_own * (8 * F_IN + 7 * HEADS * WIDTH * 4 + 12 * HEADS * 4 + 8)
        impl = sum(kernel_alg_bytes(k, E_rank, n_own, s=s_bytes) for k in
                   ("gatmi_proj_fwd", "gatmi_gat_fwd", "gatmi_gat_bwd_node", "gatmi_gat_bwd_csc", "gatmi_proj_bwd_params"))
        ig = impl / (ms_per_step * 1e-3) / 1e9
        if world == 1:
            partitioning = "single GPU"
        elif dist_info["halo"] == "replicate":
            partitioning = ("row-partitioned x%d, halo sources' input rows replicated at set-up and projected locally; per step one "
                            "RCCL all-reduce of the parameter gradients" % world)
        else:
            partitioning = ("row-partitioned x%d, RCCL all-to-all of halo Wh rows (forward) and of their gradients (backward) every "
                            "step%s, one all-reduce of the parameter gradients"
                            % (world, ", on a communication stream beside the sweeps over owned sources / owned columns" if dist_info["overlap"] else ""))
        line = {
            "metric": "GAT layer fwd+bwd edges/sec (8 heads) + % HBM roofline at 1/2/4/8 GPUs",
            "value": E_total / (ms_per_step * 1e-3), "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if storage == "f32" else "f32 arithmetic, bf16 gather tables", "data": "synthetic",
            "backend": dist_info["backend"] if dist_info else None,
            "config": {"workload": "%s: %d nodes, %d edge entries (self-loops included), %d-d features, %d heads x %d, ELU, "
                                   "fwd+bwd of one layer" % (shape_name, n_own, E_rank, F_IN, HEADS, WIDTH),
                       "name": args.workload, "partitioning": partitioning},
            "roofline": roofline,
            "layer_roofline": {"bound": "hbm", "alg_bytes": lb, "achieved": round(lg, 1), "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": round(lg / HBM_PEAK_GBPS, 4),
                               "note": "survey byte model (944 B/edge + 2984 B/node), the common yardstick; impl_* prices the bytes "
                                       "this implementation's five kernels actually have to move",
                               "impl_bytes": impl, "impl_achieved": round(ig, 1), "impl_frac": round(ig / HBM_PEAK_GBPS, 4),
                               "b_min_bytes": b_min, "b_min_achieved": round(b_min / (ms_per_step * 1e-3) / 1e9, 1),
                               "b_min_note": "compulsory lower bound (each array touched once, gathers not counted): what an "
                                             "infinite cache would leave; the gap to impl_bytes is the gather traffic"},
            "kernels_ms": {k: round(v, 4) for k, v in sorted(per_step.items())},
            "info": info,
        }
        if dist_info:
            line["dist"] = dist_info
        if replicate_line:
            e2, k2, c2 = replicate_line
            ms2 = e2 / args.steps * 1e3
            line["replicate_mode"] = dict(c2, ms_per_step=round(ms2, 4), value=E_total / (ms2 * 1e-3),
                                          kernels_ms={k: round(v[0] * v[1] / args.steps, 4) for k, v in sorted(k2.items())})
        elif world > 1 and replicate_error:
            line["replicate_mode"] = {"error": replicate_error}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
