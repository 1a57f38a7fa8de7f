"""gatmi_gat_fwd_noalloc (the forward entry point that allocates nothing: split-row partial results
go to a fixed store inside the library instead of a caller workspace) against gatmi_gat_fwd, through
the C ABI.  Same kernels, same order of operations: every output must be bit-identical.  Two calls
in a row, the second on a smaller graph (the store keeps the first call's contents), then the size
limit and a graph without split rows."""
import ctypes

import numpy as np
import pytest
import torch

from tests.util import layer_params, random_graph

pytestmark = pytest.mark.gpu


def _inputs(n, F, H, D, seed, hubs, hub_deg, chunk, dev):
    from gat_amd import CSRGraph, _lib, ops
    from gat_amd._lib import ptr
    dst, src = random_graph(n, 4.0, seed=seed, hubs=hubs, hub_deg=hub_deg)
    graph = CSRGraph.from_coo(dst, src, n, dev, chunk=chunk)
    p = layer_params(F, H, D, seed + 1)
    x = torch.tensor(np.random.default_rng(seed + 2).normal(0, 1, (n, F)), dtype=torch.float32, device=dev)
    cfg = ops.make_cfg(H, D, "elu", "concat")
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in p.items()}
    _, _, _, _, Wh, f1, f2, _ = ops._project(x, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], cfg)
    bias = ops._pad_last(tp["bias"].reshape(H, D), D, cfg.dp).reshape(-1)
    return graph, cfg, Wh, f1, f2, bias


def _outputs(n, H, dp, dev, fill):
    mk = lambda *s: torch.full(s, fill, dtype=torch.float32, device=dev)
    return [mk(n, H * dp), mk(n, H * dp), mk(n, H * dp), mk(n, H), mk(n, H)]       # out, agg, pos, lse, q


def _call(name, graph, cfg, Wh, f1, f2, bias, outs, ws=None, n_slots=None):
    from gat_amd import _lib
    from gat_amd._lib import ptr
    H, dp = cfg.H, cfg.dp
    sp = ptr(graph.split_csr) if len(graph.split_csr) else None
    head = [graph.n_dst, H, dp, ptr(graph.items_csr), len(graph.items_csr), sp, len(graph.split_csr)]
    if name == "gatmi_gat_fwd_noalloc":
        head.append(graph.n_slots_csr if n_slots is None else n_slots)
    tail = [ptr(graph.col), ptr(Wh), _lib.DTYPE_F32, H * dp, ptr(f1), ptr(f2), None, None, ptr(bias), None, _lib.ACT_ELU, _lib.REDUCE_CONCAT,
            0.2, 1.0, 1.0, 0] + [ptr(o) for o in outs]
    if name == "gatmi_gat_fwd":
        tail.append(ptr(ws))
    tail.append(_lib.current_stream())
    return getattr(_lib.lib, name)(*(head + tail))


def _compare(n, F, H, D, seed, hubs, hub_deg, chunk, dev):
    from gat_amd import _lib
    graph, cfg, Wh, f1, f2, bias = _inputs(n, F, H, D, seed, hubs, hub_deg, chunk, dev)
    ws = None
    if graph.n_slots_csr:
        ws = torch.empty(_lib.lib.gatmi_gat_fwd_ws_floats(graph.n_slots_csr, H, cfg.dp), dtype=torch.float32, device=dev)
    want = _outputs(n, H, cfg.dp, dev, 7.0)
    got = _outputs(n, H, cfg.dp, dev, -7.0)
    assert _call("gatmi_gat_fwd", graph, cfg, Wh, f1, f2, bias, want, ws=ws) == 0, _lib.last_error()
    assert _call("gatmi_gat_fwd_noalloc", graph, cfg, Wh, f1, f2, bias, got) == 0, _lib.last_error()
    torch.cuda.synchronize()
    for a, b, name in zip(got, want, ("out", "agg", "pos", "lse", "q")):
        assert torch.equal(a, b), "%s differs between gat_fwd_noalloc and gat_fwd (n=%d, %d split rows)" % (name, n, len(graph.split_csr))
        assert torch.isfinite(a).all()
    return graph


def test_noalloc_forward_equals_workspace_forward_twice_in_a_row():
    dev = torch.device("cuda:0")
    big = _compare(900, 24, 8, 8, seed=5, hubs=4, hub_deg=700, chunk=32, dev=dev)
    assert len(big.split_csr) >= 4 and big.n_slots_csr > 40
    small = _compare(300, 12, 8, 8, seed=9, hubs=1, hub_deg=200, chunk=32, dev=dev)       # fewer slots: stale records stay behind it
    assert 0 < small.n_slots_csr < big.n_slots_csr
    _compare(257, 10, 1, 7, seed=3, hubs=2, hub_deg=150, chunk=16, dev=dev)                # one 7-wide head (padded to 8)
    flat = _compare(200, 8, 4, 16, seed=4, hubs=0, hub_deg=0, chunk=256, dev=dev)          # no split rows: the store is not touched
    assert flat.n_slots_csr == 0


def test_noalloc_forward_rejects_sizes_beyond_its_store():
    from gat_amd import _lib
    dev = torch.device("cuda:0")
    graph, cfg, Wh, f1, f2, bias = _inputs(300, 12, 8, 8, 9, 1, 200, 32, dev)
    outs = _outputs(300, cfg.H, cfg.dp, dev, 0.0)
    limit = (8 << 20) // 4 // (3 * cfg.H + 2 * cfg.H * cfg.dp)
    rc = _call("gatmi_gat_fwd_noalloc", graph, cfg, Wh, f1, f2, bias, outs, n_slots=limit + 1)
    assert rc == -5 and "built-in store" in _lib.last_error()                              # GATMI_ERANGE, nothing launched
    rc = _call("gatmi_gat_fwd_noalloc", graph, cfg, Wh, f1, f2, bias, outs, n_slots=0)     # split rows but "no" slots
    assert rc == -1
    torch.cuda.synchronize()
    assert all(bool((o == 0).all()) for o in outs)
    # a slot count smaller than the items' own: the pieces beyond it are skipped, never written past the store's claim
    rc = _call("gatmi_gat_fwd_noalloc", graph, cfg, Wh, f1, f2, bias, outs, n_slots=1)
    torch.cuda.synchronize()
    assert rc == 0


def test_noalloc_forward_takes_two_phase_items_whose_slots_have_no_split_record():
    """gatmi_work_items_phased parks every row that has entries in both phases in a slot; a row with one piece
    per phase gets a slot but no split record, so a shard can hold slots and no split rows at all."""
    from gat_amd import _lib
    from gat_amd._lib import ptr
    dev = torch.device("cuda:0")
    n, H = 240, 8
    graph, cfg, Wh, f1, f2, bias = _inputs(n, 12, H, 8, seed=21, hubs=0, hub_deg=0, chunk=256, dev=dev)
    rowptr = graph.rowptr_host
    cut = np.ascontiguousarray(rowptr[:-1] + np.diff(rowptr) // 2, dtype=np.int64)
    na, nb, ns, nl = (ctypes.c_int64() for _ in range(4))
    _lib.call("gatmi_work_items_phased_count", n, ptr(rowptr), ptr(cut), graph.chunk, ctypes.byref(na), ctypes.byref(nb),
              ctypes.byref(ns), ctypes.byref(nl))
    assert ns.value == 0 and nl.value > 0, "wanted slots without split records, got %d split rows / %d slots" % (ns.value, nl.value)
    ia, ib = np.empty((na.value, 4), dtype=np.int32), np.empty((nb.value, 4), dtype=np.int32)
    sp = np.empty((max(ns.value, 1), 4), dtype=np.int32)
    _lib.call("gatmi_work_items_phased", n, ptr(rowptr), ptr(cut), graph.chunk, ptr(ia), ptr(ib), ptr(sp))
    ia, ib = torch.from_numpy(ia).to(dev), torch.from_numpy(ib).to(dev)

    def run(name, outs, ws):
        for items in (ia, ib):
            head = [n, H, cfg.dp, ptr(items), len(items), None, 0]
            if name == "gatmi_gat_fwd_noalloc":
                head.append(nl.value)
            tail = [ptr(graph.col), ptr(Wh), _lib.DTYPE_F32, H * cfg.dp, ptr(f1), ptr(f2), None, None, ptr(bias), None, _lib.ACT_ELU,
                    _lib.REDUCE_CONCAT, 0.2, 1.0, 1.0, 0] + [ptr(o) for o in outs]
            if name == "gatmi_gat_fwd":
                tail.append(ptr(ws))
            tail.append(_lib.current_stream())
            assert getattr(_lib.lib, name)(*(head + tail)) == 0, _lib.last_error()

    ws = torch.empty(_lib.lib.gatmi_gat_fwd_ws_floats(nl.value, H, cfg.dp), dtype=torch.float32, device=dev)
    want, got = _outputs(n, H, cfg.dp, dev, 7.0), _outputs(n, H, cfg.dp, dev, -7.0)
    run("gatmi_gat_fwd", want, ws)
    run("gatmi_gat_fwd_noalloc", got, None)
    torch.cuda.synchronize()
    whole = _outputs(n, H, cfg.dp, dev, 3.0)
    assert _call("gatmi_gat_fwd", graph, cfg, Wh, f1, f2, bias, whole, ws=ws) == 0, _lib.last_error()
    torch.cuda.synchronize()
    for a, b, c, name in zip(got, want, whole, ("out", "agg", "pos", "lse", "q")):
        assert torch.equal(a, b), "%s differs between gat_fwd_noalloc and gat_fwd on two-phase items" % name
        assert torch.allclose(a, c, rtol=1e-5, atol=1e-6), "%s: two sweeps differ from one" % name
