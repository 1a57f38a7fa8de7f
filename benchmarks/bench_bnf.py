#!/usr/bin/env python3
"""Secondary measurement: the brain-network attention step (utils/layers.py:174-179 with :229-237)
and the inner-product decoder (:307-313) at the reference's regime (BNF_AE.py:17-30: batch 50,
246 Brainnetome nodes, 6 heads, 64 features) and at a large batch, forward + backward, next to the
same math written with eager torch ops on the same GPU (what a line-by-line port would run)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def eager_bnf(x, A, m):
    """one torch op per reference line, all heads at once"""
    B, N, F = x.shape
    H, Dz = m.a1.shape
    z = (x @ m.W.reshape(F, H * Dz)).reshape(B, N, H, Dz)
    f1 = (z * m.a1).sum(-1) + m.b1
    f2 = (z * m.a2).sum(-1) + m.b2
    logits = f1.permute(0, 2, 1)[:, :, :, None] + f2.permute(0, 2, 1)[:, :, None, :]
    coefs = torch.softmax(torch.nn.functional.leaky_relu(logits, 0.2), dim=-1)
    att = m.m1[None, :, None, None] + coefs
    net1 = (A > 0).float()[:, None] * m.m2[None, :, None, None] + A[:, None]
    return torch.einsum("bhij,bjf->bihf", net1 * att, x)


def main():
    from gat_amd import graphs
    from gat_amd.layers import BNFAttention, InnerProductDecoder
    dev = torch.device("cuda:0")
    res = []
    for B, N, F, H in ((50, 246, 64, 6), (1024, 246, 64, 6)):
        g = torch.Generator(device=dev).manual_seed(0)
        x = torch.randn(B, N, F, device=dev, generator=g, requires_grad=True)
        A = torch.rand(B, N, N, device=dev, generator=g)
        A = ((A + A.transpose(1, 2)) / 2 * (torch.rand(B, N, N, device=dev, generator=g) < 0.3)).requires_grad_(True)
        m = BNFAttention(F, H, 64, device=dev)
        dout = torch.randn(B, N, H, F, device=dev, generator=g)

        def run(fn):
            def step():
                x.grad = A.grad = None
                for p in m.parameters():
                    p.grad = None
                fn().backward(dout)
            return step
        fused = timed(run(lambda: m(x, A)))
        eager = timed(run(lambda: eager_bnf(x, A, m)))
        # the same fused step replayed from a HIP graph (the library only enqueues, so it captures whole)
        replay_fn, _ = graphs.capture(run(lambda: m(x, A)))
        replay = timed(replay_fn)
        z = torch.randn(B, N, F, device=dev, generator=g, requires_grad=True)
        dflat = torch.randn(B * N * N, device=dev, generator=g)

        def dec(fn):
            def step():
                z.grad = None
                fn().backward(dflat)
            return step
        dfused = timed(dec(lambda: InnerProductDecoder(z)))
        deager = timed(dec(lambda: torch.sigmoid(z @ z.transpose(1, 2)).reshape(-1)))
        res.append({"B": B, "N": N, "F": F, "heads": H, "bnf_step_ms": round(fused, 3), "bnf_step_graph_replay_ms": round(replay, 3), "bnf_eager_torch_ms": round(eager, 3),
                    "decoder_ms": round(dfused, 3), "decoder_eager_torch_ms": round(deager, 3),
                    "decoder_GBps": round(B * N * N * 4 * 4 / (dfused * 1e-3) / 1e9, 1)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
