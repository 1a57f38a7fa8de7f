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
    ap.add_argument("--overlap", choices=["auto", "on", "pipeline", "two_phase", "off"], default="auto",
                    help="N > 1, exchange.  off: one all-to-all per direction, exchanges and kernels in turn.  pipeline (= on): "
                         "each direction as pairwise steps on a communication stream; a peer's rows travel while the next peer's "
                         "are packed (forward) or while the next owner's halo columns are swept (backward).  two_phase: pipeline, "
                         "and the forward sweep is cut into owned-source and halo-source phases (every row's softmax state is "
                         "parked in between).  auto (default): a full leg (W warm-up + K timed steps) of off, then of pipeline, "
                         "three trial steps of two_phase and a full leg of it only if those were the fastest yet; the line's "
                         "value is the fastest full leg, all legs are reported under dist.legs")
    ap.add_argument("--no-configs4", dest="configs4", action="store_false",
                    help="N > 1, --workload products: skip the second part of the run, which times every rank's share of "
                         "BASELINE.json configs[4] (12.5 M rows / 250 M edges + self-loops per rank) and reports it under 'configs4'")
    ap.add_argument("--c4-nodes", type=int, default=0, help="override the configs4 part's node count per GPU (debug / rehearsal)")
    ap.add_argument("--c4-edges", type=int, default=0, help="override the configs4 part's edge count per GPU (debug / rehearsal)")
    ap.add_argument("--deadline", type=int, default=300,
                    help="N > 1: seconds one leg may take before rank 0 prints the line with the legs measured so far and "
                         "ends the run with a non-zero status (a collective that never completes)")
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
# one-GPU denominators of the configs[4] target, measured by hand on one MI355X (fp32, same seeds)
CONFIGS4_DENOMINATORS = {
    "share_alone_ms": 36.1, "share_alone_source": "python bench.py --workload rmat2b (one rank's 12.5 M rows / 262.5 M entries, no halo)",
    "whole_graph_one_gpu_ms": 280.3, "whole_graph_one_gpu_edges_per_s": 7.49e9,
    "whole_graph_source": "profiles/r02_c5_single_gpu.json (benchmarks/bench_c5_single.py: all eight shards' 100 M nodes / 2.1 G entries on one GPU)",
    "target": "8 GPUs >= 6 x the one-GPU rate <=> <= 46.7 ms per step"}


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


class LineKeeper:
    """Rank 0's JSON line, built leg by leg and printed exactly once, whatever happens afterwards: at the normal
    end, or by a watchdog thread when a leg overruns its deadline (a collective that never completes, e.g. because
    one rank failed inside a step while the others wait for it) — the legs measured until then are not lost.
    The watchdog then ends the process with a non-zero status; the launcher takes the other ranks down."""

    def __init__(self, rank):
        import threading
        self.rank, self.line, self.printed = rank, None, False
        self.lock = threading.Lock()
        self.deadline, self.stage = None, None
        if rank == 0:
            threading.Thread(target=self._watch, daemon=True).start()

    def arm(self, stage, seconds):
        self.stage, self.deadline = stage, time.monotonic() + seconds

    def disarm(self):
        self.deadline = None

    def emit(self, extra=None):
        with self.lock:
            if self.printed or self.rank != 0 or self.line is None:
                return
            if extra:
                self.line.update(extra)
            print(json.dumps(self.line), flush=True)
            self.printed = True

    def _watch(self):
        while True:
            time.sleep(1.0)
            d = self.deadline
            if d is not None and time.monotonic() > d:
                self.emit({"aborted_in": self.stage, "abort_reason": "the leg did not finish within its deadline; the line holds what was measured before it"})
                sys.stdout.flush()
                os._exit(3)


def kernel_ms(ksum, steps):
    return {k: v[0] * v[1] / steps for k, v in ksum.items()}       # ms per step of each entry point (all its launches)


def build_line(args, world, storage, shape_name, workload, n_own, E_rank, E_total, ms_per_step, ksum, info, dist_info):
    """The JSON line of one measured configuration (rank 0)."""
    s_bytes = 4 if storage == "f32" else 2
    per_step = kernel_ms(ksum, args.steps)
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
        how = {"off": "", "pipeline": ", as pairwise steps on a communication stream (a peer's rows travel while the next peer's are packed / swept)",
               "two_phase": ", as pairwise steps on a communication stream, the forward sweep cut into owned-source and halo-source phases"}
        partitioning = ("row-partitioned x%d, RCCL exchange of halo Wh rows (forward) and of their gradients (backward) every "
                        "step%s, one all-reduce of the parameter gradients" % (world, how.get(dist_info["overlap"], "")))
    return {
        "metric": "GAT layer fwd+bwd edges/sec (8 heads) + % HBM roofline at 1/2/4/8 GPUs",
        "value": E_total / (ms_per_step * 1e-3), "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if storage == "f32" else "f32 arithmetic, bf16 gather tables", "data": "synthetic",
        "backend": dist_info["backend"] if dist_info else None,
        "config": {"workload": "%s: %d nodes, %d edge entries (self-loops included), %d-d features, %d heads x %d, ELU, "
                               "fwd+bwd of one layer" % (shape_name, n_own, E_rank, F_IN, HEADS, WIDTH),
                   "name": workload, "partitioning": partitioning},
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


SHAPES = {
    "products": (None, None, "R-MAT power-law graph at ogbn-products shape per GPU"),
    "rmat2b": (RMAT2B_NODES_PER_GPU, RMAT2B_EDGES_PER_GPU, "one eighth of the 100 M-node / 2 B-edge R-MAT graph per GPU"),
}


def run_shard_legs(args, leg_args, workload, n_own, e_own, rank, world, dev, backend, keeper, deadline_s):
    """Every leg of one workload on this rank's shard.  Returns (legs, shard facts): legs maps a leg's name to
    {"ms_per_step", "kernels", "comm"} or {"error"}.  A leg's outcome is agreed on by all ranks (one MAX
    all-reduce of a failure flag after it), so that every rank goes on to the same next collective."""
    import torch.distributed as dist
    from gat_amd import _lib, dist as gdist
    keeper.arm("set-up of the %s shard" % workload, deadline_s * 2)
    bench = gdist.BenchShard(rank, world, n_own, e_own, F_IN, HEADS, WIDTH, dev, args.chunk, args.storage)
    facts = {"E_rank": bench.E_rank, "info": bench.info(), "n_halo": bench.pg.n_halo, "n_send": bench.pg.exchange.n_send,
             "local_entry_fraction": round(bench.pg.local_entries / max(1, bench.E_rank), 4)}
    t = torch.tensor([float(bench.E_rank)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    facts["E_total"] = float(t[0])
    legs = {}

    def agreed_failure(local_error):
        flag = torch.tensor([1.0 if local_error else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return float(flag[0]) > 0

    def leg(name, halo, overlap, la):
        keeper.arm("%s / %s" % (workload, name), deadline_s)
        err = None
        try:
            step = bench.step_fn(halo, overlap=overlap)
            bench.pg.exchange.enable_timing()
            elapsed, ksum = timed_steps(step, la, world, dev, _lib)
            torch.cuda.synchronize()
            a2a_total_ms, sent, received, n_x = bench.pg.exchange.a2a_summary()
            per = la.steps + la.warmup               # the exchange's events cover the warm-up steps too
            res = {"ms_per_step": elapsed / la.steps * 1e3, "ksum": ksum, "steps": la.steps,
                   "comm": {"exchange_ms": round(a2a_total_ms / per, 4), "exchanges_per_step": n_x // per,
                            "bytes_sent_per_step": sent // per, "bytes_received_per_step": received // per}}
        except Exception as exc:                     # noqa: BLE001 - a failed leg must not cost the line the legs that ran
            err = "%s: %s" % (type(exc).__name__, exc)
        if agreed_failure(err):
            res = {"error": err or "failed on another rank"}
        legs[name] = res
        bench.release(halo)
        return "error" not in res

    want_exchange = args.halo in ("exchange", "both")
    want_replicate = args.halo in ("replicate", "both")
    if want_exchange:
        ov = args.overlap
        if ov in ("auto", "off"):
            leg("exchange_sequential", "exchange", False, leg_args)            # one all-to-all per direction: the plainest form first
        if ov in ("auto", "on", "pipeline"):
            leg("exchange_pipeline", "exchange", "pipeline", leg_args)
        if ov == "two_phase":
            leg("exchange_two_phase", "exchange", "two_phase", leg_args)
        elif ov == "auto" and "error" not in legs.get("exchange_pipeline", {"error": 1}):
            # the two-phase forward sweep parks every row's state: it pays only when the links are slow and the share
            # of owned sources is large, so it gets three trial steps and a full leg only if those are the fastest yet
            trial = argparse.Namespace(warmup=2, steps=3)
            if leg("exchange_two_phase_trial", "exchange", "two_phase", trial):
                best = min(v["ms_per_step"] for k, v in legs.items() if k.startswith("exchange") and "error" not in v and "trial" not in k)
                if legs["exchange_two_phase_trial"]["ms_per_step"] < best:
                    leg("exchange_two_phase", "exchange", "two_phase", leg_args)
    if want_replicate:
        leg("replicate", "replicate", False, leg_args)
    keeper.disarm()
    del bench
    torch.cuda.empty_cache()
    return legs, facts


def best_exchange(legs):
    ok = {k: v for k, v in legs.items() if k.startswith("exchange") and "trial" not in k and "error" not in v}
    return min(ok, key=lambda k: ok[k]["ms_per_step"]) if ok else None


def legs_summary(legs):
    out = {}
    for k, v in legs.items():
        out[k] = {"error": v["error"]} if "error" in v else dict(
            v["comm"], ms_per_step=round(v["ms_per_step"], 4), steps=v["steps"],
            kernels_ms={n: round(m, 4) for n, m in sorted(kernel_ms(v["ksum"], v["steps"]).items())})
    return out


def run_distributed(args, rank, world, local):
    import datetime
    import torch.distributed as dist
    # GATMI_DIST_BACKEND=gloo rehearses N > 1 on a box with fewer GPUs than ranks (ranks share devices,
    # buffers are staged through the host); the measured configuration is always nccl = RCCL.
    backend = os.environ.get("GATMI_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=args.deadline * 3))
    else:
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=args.deadline * 3))
    keeper = LineKeeper(rank)
    name_of = {"exchange_sequential": "off", "exchange_pipeline": "pipeline", "exchange_two_phase": "two_phase"}
    n_own, e_own, shape_name = SHAPES[args.workload]
    from gat_amd import synth
    n_own, e_own = args.nodes or n_own or synth.PRODUCTS_NODES, args.edges or e_own or synth.PRODUCTS_EDGES
    rc = 0
    legs, facts = run_shard_legs(args, args, args.workload, n_own, e_own, rank, world, dev, backend, keeper, args.deadline)
    head = best_exchange(legs) if args.halo != "replicate" else None
    if head is None and "error" not in legs.get("replicate", {"error": 1}):
        head = "replicate"
    if head is None:
        if rank == 0:
            print(json.dumps({"metric": "GAT layer fwd+bwd edges/sec (8 heads) + % HBM roofline at 1/2/4/8 GPUs", "value": None, "n_gpus": world,
                              "error": "no leg completed", "legs": legs_summary(legs)}), flush=True)
        dist.destroy_process_group()
        raise SystemExit(2)
    if rank == 0:
        r = legs[head]
        dist_info = dict(r["comm"], backend=("rccl" if backend == "nccl" else backend), world_size=world,
                         halo="replicate" if head == "replicate" else "exchange", overlap=name_of.get(head),
                         n_halo=facts["n_halo"], n_send=facts["n_send"], local_entry_fraction=facts["local_entry_fraction"],
                         legs=legs_summary(legs),
                         note="value = the fastest fully timed exchange-mode leg (every leg runs the same W warm-up + K timed steps)")
        failed = [k for k, v in legs.items() if "error" in v]
        if failed and args.halo != "replicate" and best_exchange(legs) is None:
            dist_info["exchange_error"] = "; ".join("%s: %s" % (k, legs[k]["error"]) for k in failed if k.startswith("exchange"))
        la = argparse.Namespace(steps=r["steps"], warmup=args.warmup)
        line = build_line(la, world, args.storage, shape_name, args.workload, n_own, facts["E_rank"], facts["E_total"],
                          r["ms_per_step"], r["ksum"], facts["info"], dist_info)
        line["dist"] = dist_info
        if "replicate" in legs and head != "replicate":
            rp = legs["replicate"]
            line["replicate_mode"] = {"error": rp["error"]} if "error" in rp else dict(
                legs_summary({"replicate": rp})["replicate"], value=facts["E_total"] / (rp["ms_per_step"] * 1e-3))
        keeper.line = line
    if any("error" in v for v in legs.values()) and (args.halo == "replicate" or best_exchange(legs) is None):
        rc = 4
    # configs[4]: every rank's share of the 100 M-node / 2 B-edge graph, in the same process, so that the first
    # multi-GPU line already carries the graph the scaling target is stated on
    if args.workload == "products" and args.configs4 and (not (args.nodes or args.edges) or args.c4_nodes):
        c4 = {"denominators_one_gpu": CONFIGS4_DENOMINATORS}
        try:
            la = argparse.Namespace(steps=min(args.steps, 10), warmup=min(args.warmup, 3))
            n4, e4, name4 = SHAPES["rmat2b"]
            n4, e4 = args.c4_nodes or n4, args.c4_edges or e4
            legs4, facts4 = run_shard_legs(args, la, "rmat2b", n4, e4, rank, world, dev, backend, keeper, args.deadline)
            c4.update(workload="%s: %d nodes, %d edge entries per rank" % (name4, n4, facts4["E_rank"]), world_size=world,
                      n_halo=facts4["n_halo"], n_send=facts4["n_send"], local_entry_fraction=facts4["local_entry_fraction"],
                      legs=legs_summary(legs4))
            b4 = best_exchange(legs4)
            for key, lname in (("exchange", b4), ("replicate", "replicate" if "error" not in legs4.get("replicate", {"error": 1}) else None)):
                if lname:
                    ms = legs4[lname]["ms_per_step"]
                    c4[key] = {"leg": lname, "ms_per_step": round(ms, 4), "value": facts4["E_total"] / (ms * 1e-3), "unit": "edges/s",
                               "vs_whole_graph_on_one_gpu": round(facts4["E_total"] / (ms * 1e-3) / CONFIGS4_DENOMINATORS["whole_graph_one_gpu_edges_per_s"], 3)
                               if world == 8 else None}
        except Exception as exc:                     # noqa: BLE001 - the products line stands whatever happens here
            c4["error"] = "%s: %s" % (type(exc).__name__, exc)
            rc = rc or 5
        if rank == 0 and keeper.line is not None:
            keeper.line["configs4"] = c4
    keeper.disarm()
    keeper.emit()
    try:
        dist.destroy_process_group()
    except Exception:                                # noqa: BLE001
        pass
    if rc:
        raise SystemExit(rc)


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
    if world > 1:
        return run_distributed(args, rank, world, local)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import gat_amd  # noqa: F401
    from gat_amd import _lib, synth, CSRGraph
    from gat_amd.layers import GATHeads

    n_own, e_own, shape_name = SHAPES[args.workload]
    n_own, e_own = args.nodes or n_own or synth.PRODUCTS_NODES, args.edges or e_own or synth.PRODUCTS_EDGES
    storage = args.storage
    dst, src = synth.rmat_graph(n_own, e_own, 4, dev)
    graph = CSRGraph.from_device_coo(dst, src, n_own, n_own, chunk=args.chunk)
    del dst, src
    E_rank = graph.nnz
    x = synth.normal((n_own, F_IN), 5, dev)
    dout = synth.normal((n_own, HEADS * WIDTH), 8, dev)
    heads = GATHeads(F_IN, HEADS, WIDTH, device=dev, storage=storage)
    heads.load(synth.glorot_layer(F_IN, HEADS, WIDTH, 7, dev))
    heads.train()
    params = list(heads.parameters())

    def layer_step():
        for p in params:
            p.grad = None
        out = heads(x, graph, "elu", "concat")
        out.backward(dout)
    elapsed, ksum = timed_steps(layer_step, args, 1, dev, _lib)
    line = build_line(args, 1, storage, shape_name, args.workload, n_own, E_rank, float(E_rank), elapsed / args.steps * 1e3, ksum,
                      {"graph": repr(graph)}, None)
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
