#!/usr/bin/env python3
"""HIP-graph replay going stale on ROCm 7.2 / PyTorch 2.10+rocm7.0, shown with plain torch operators (nothing of gat_amd is
imported): a chain of small layers is captured with torch.cuda.graph, ordinary launches are issued, then 20 replays on fresh
inputs are each compared with the same chain run eagerly.

    graph_replay_min.py 10 launches3000     0 of 20 replays differ
    graph_replay_min.py 10 launches6000     the last replays differ (about 6 800 launches since the capture)
    graph_replay_min.py 10 launches20000    20 of 20 differ: outputs keep what the last eager call left
    graph_replay_min.py 10 launches20000 ownstream      the same: a replay stream of its own does not help
    graph_replay_min.py 10 launches20000 envearly       0 of 20: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 set before the first HIP call
    graph_replay_min.py 10 launches20000 envlate        20 of 20: the runtime reads the switch once, at its first call

Single operators captured alone replay correctly, memset nodes are ordered against the kernels around them, and no ordinary
allocation overlaps the graph's private pool; HIP_FORCE_DEV_KERNARG=0 and DEBUG_HIP_GRAPH_KERNARG_OPT=0 change nothing.

flags: preN (N eager chains before the first replay), launchesN (N in-place launches, no allocation), ownstream, envearly,
envlate, mainwarm (warm up on the current stream), scalars (0-d outputs), nonorm, nosoft (shorter chains)."""
import os
import sys

import torch

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10
flags = set(sys.argv[2:])
if "envearly" in flags:                                      # after torch is loaded, before the first HIP call
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
dev = torch.device("cuda:0")
if "envlate" in flags:                                       # after HIP has been initialised
    torch.zeros(1, device=dev)
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
torch.manual_seed(0)
Ws = [torch.randn(64, 64, device=dev) / 8 for _ in range(L)]
x = torch.randn(50, 90, 64, device=dev)
a = torch.rand(50, 90, 90, device=dev)
outs = [torch.zeros((), device=dev) for _ in range(3)] if "scalars" in flags else list(torch.zeros(3, device=dev))


def chain():
    h = x
    for k, W in enumerate(Ws):
        h = torch.relu(h @ W)
        if k % 10 == 0 and "nosoft" not in flags:
            h = torch.softmax(a, dim=-1) @ h
        if k % 7 == 0 and "nonorm" not in flags:
            h = h / (h.norm(dim=1, keepdim=True) + 1e-6)
    r = torch.sigmoid(h @ h.transpose(1, 2))
    for o, v in zip(outs, (h.abs().max(), r.abs().max(), ((a > 0).float() - r).pow(2).mean())):
        o.copy_(v)


if "mainwarm" in flags:
    chain(); chain()
    torch.cuda.synchronize()
else:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(); chain()
    torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain()
gstream = torch.cuda.Stream()
pre = [f for f in flags if f.startswith("pre")]
for _ in range(int(pre[0][3:]) if pre else 0):               # eager runs between capture and the first replay
    chain()
tiny = torch.zeros(8, device=dev)
for f in flags:
    if f.startswith("launches"):                             # plain in-place launches, no allocation
        for _ in range(int(f[8:])):
            tiny.add_(1.0)
torch.cuda.synchronize()
bad = []
for i in range(20):
    gen = torch.Generator(device=dev).manual_seed(1000 + i)
    x.copy_(torch.randn(50, 90, 64, device=dev, generator=gen))
    a.copy_(torch.rand(50, 90, 90, device=dev, generator=gen))
    if "ownstream" in flags:                                 # the graph gets a stream that never sees an eager launch
        cur = torch.cuda.current_stream()
        gstream.wait_stream(cur)
        with torch.cuda.stream(gstream):
            g.replay()
        cur.wait_stream(gstream)
    else:
        g.replay()
    got = [float(o) for o in outs]
    chain()
    want = [float(o) for o in outs]
    if any(abs(p - q) > 1e-4 * max(1.0, abs(q)) for p, q in zip(got, want)):
        bad.append((i, [round(t, 4) for t in got], [round(t, 4) for t in want]))
print("layers %d %s: %d of 20 replays differ" % (L, sorted(flags), len(bad)), bad[:2])
