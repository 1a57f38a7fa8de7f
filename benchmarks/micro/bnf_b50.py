#!/usr/bin/env python3
"""The brain-network attention step at the reference's batch (BNF_AE.py:17-30: 50 graphs of 246 nodes, 6 heads, 64
features), forward + backward, for a kernel trace:  rocprofv3 --kernel-trace --stats -- python3 benchmarks/micro/bnf_b50.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gat_amd.layers import BNFAttention

dev = torch.device("cuda:0")
B, N, F, H = int(os.environ.get("BNF_B", "50")), 246, 64, 6
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, N, F, device=dev, generator=g, requires_grad=True)
A = torch.rand(B, N, N, device=dev, generator=g)
A = ((A + A.transpose(1, 2)) / 2 * (torch.rand(B, N, N, device=dev, generator=g) < 0.3)).requires_grad_(True)
m = BNFAttention(F, H, 64, device=dev)
dout = torch.randn(B, N, H, F, device=dev, generator=g)
for _ in range(30):
    x.grad = A.grad = None
    for p in m.parameters():
        p.grad = None
    m(x, A).backward(dout)
torch.cuda.synchronize()
