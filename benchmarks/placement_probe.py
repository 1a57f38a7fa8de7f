"""Does the time of the per-source backward kernel depend on WHERE its tables lie?

gat_bwd_csc gathers from two tables per entry (g rows and the per-row statistics) and writes a third
(dWh).  This probe carves the three out of one slab at chosen distances from each other and times the
kernel for each placement on the benchmark graph.  Prints one JSON line per placement."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--edges", type=int, default=0)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    from gat_amd import CSRGraph, _lib, ops, synth
    from gat_amd._lib import ptr
    dev = torch.device("cuda:0")
    n = args.nodes or synth.PRODUCTS_NODES
    e = args.edges or synth.PRODUCTS_EDGES
    H, D, F = 8, 8, 100
    hd = H * D
    dst, src = synth.rmat_graph(n, e, 4, dev)
    g = CSRGraph.from_device_coo(dst, src, n, n)
    del dst, src
    p = synth.glorot_layer(F, H, D, 7, dev)
    x = synth.normal((n, F), 5, dev)
    cfg = ops.make_cfg(H, D, "elu", "concat")
    _, a1p, a2p, _, Wh, f1, f2, _ = ops._project(x, p["W"], p["a1"], p["a2"], p["b1"], p["b2"], cfg)
    b2c = p["b2"].reshape(H).contiguous()
    st = _lib.current_stream()
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    out, agg, pos, lse, q = new(n, hd), new(n, hd), new(n, hd), new(n, H), new(n, H)
    ws = new(max(1, _lib.lib.gatmi_gat_fwd_ws_floats(g.n_slots_csr, H, D)))
    _lib.call("gatmi_gat_fwd", n, H, D, ptr(g.items_csr), len(g.items_csr), ptr(g.split_csr), len(g.split_csr),
              ptr(g.col), ptr(Wh), _lib.DTYPE_F32, hd, ptr(f1), None, ptr(a2p), ptr(b2c), None, None,
              _lib.ACT_ELU, _lib.REDUCE_CONCAT, 0.2, 1.0, 1.0, 0, ptr(out), ptr(agg), ptr(pos), ptr(lse), ptr(q), ptr(ws), st)
    dout = synth.normal((n, hd), 8, dev)
    g0, stats0, df1, dbias = new(n, hd), new(n, 3 * H), new(n, H), new(hd)
    wsn = new(_lib.lib.gatmi_gat_bwd_node_ws_floats(H, D))
    _lib.call("gatmi_gat_bwd_node", n, H, D, ptr(dout), None, ptr(agg), ptr(pos), ptr(q), ptr(f1), ptr(lse), None,
              _lib.ACT_ELU, _lib.REDUCE_CONCAT, 0.2, ptr(g0), hd, None, 0, ptr(stats0), 3 * H, ptr(df1), ptr(dbias), ptr(wsn), st)
    del out, agg, pos, dout, x
    wsb = new(max(1, _lib.lib.gatmi_gat_bwd_ws_floats(g.n_slots_csc, H, D)))
    df2 = new(n, H)
    g_bytes, s_bytes = n * hd * 4, n * 3 * H * 4
    MB = 1 << 20
    slab = torch.empty(2 * g_bytes + s_bytes + 64 * MB, dtype=torch.uint8, device=dev)
    base = (slab.data_ptr() + 2 * MB - 1) // (2 * MB) * (2 * MB) - slab.data_ptr()          # 2 MiB aligned start

    def carve(offset, rows, width):
        t = slab[offset:offset + rows * width * 4].view(torch.float32).view(rows, width)
        assert t.data_ptr() % 256 == 0
        return t

    def up(v, a):
        return (v + a - 1) // a * a
    for gap_g_s, gap_s_d in [(0, 0), (256, 0), (4096, 0), (65536, 0), (MB, 0), (2 * MB + 4096, 0), (0, 4096), (4096, 4096),
                             (8192 + 256, 16384 + 512), (3 * MB + 768, 5 * MB + 1280)]:
        o_g = base
        o_s = up(o_g + g_bytes, 256) + gap_g_s
        o_d = up(o_s + s_bytes, 256) + gap_s_d
        gt, stt, dft = carve(o_g, n, hd), carve(o_s, n, 3 * H), carve(o_d, n, hd)
        if gap_g_s == 256:          # this slot of the list: one interleaved record per row instead of two tables
            rec = carve(o_g, n, hd + 3 * H)
            gt, stt = rec[:, :hd], rec[:, hd:]
            dft = carve(up(o_g + n * (hd + 3 * H) * 4, 256), n, hd)
        gt.copy_(g0)
        stt.copy_(stats0)
        evs = []
        for it in range(args.iters + 2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.call("gatmi_gat_bwd_csc", n, H, D, ptr(g.items_csc), len(g.items_csc), ptr(g.split_csc), len(g.split_csc),
                      ptr(g.rowidx), None, ptr(gt), _lib.DTYPE_F32, gt.stride(0), ptr(stt), stt.stride(0), ptr(Wh), None, ptr(a2p), ptr(b2c),
                      ptr(df1), ptr(a1p), ptr(a2p), 0.2, 1.0, 1.0, 0, ptr(dft), ptr(df2), ptr(wsb), st)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs[2:])
        print(json.dumps({"gap_g_to_stats": gap_g_s, "gap_stats_to_dfeat": gap_s_d, "g_mod_2MiB": (slab.data_ptr() + o_g) % (2 * MB),
                          "stats_mod_2MiB": (slab.data_ptr() + o_s) % (2 * MB), "dfeat_mod_2MiB": (slab.data_ptr() + o_d) % (2 * MB),
                          "median_ms": round(ms[len(ms) // 2], 4), "min_ms": round(ms[0], 4)}), flush=True)


if __name__ == "__main__":
    main()
