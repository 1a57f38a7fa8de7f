#!/usr/bin/env python3
"""Host-side profile of the eager brain-network autoencoder training step (where the Python time goes when the
step is not replayed from a HIP graph):  python benchmarks/micro/prof_ae_eager.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gat_amd.models import BaseGAttN, GAT_AE_BNF_V2  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N = 50, 90
model = GAT_AE_BNF_V2(1, N, 2, nb_CAM=256, hid_units=[64], n_heads=[6], device=dev)
opt, l2 = BaseGAttN.training(model.parameters(), lr=0.001, l2_coef=0.001, penalty_in_optimizer=True)
w = torch.rand(B, N, N, device=dev) * (torch.rand(B, N, N, device=dev) < 0.3)
w = (w + w.transpose(1, 2)) / 2
x = w.sum(-1, keepdim=True) / N
y = torch.nn.functional.one_hot(torch.randint(0, 2, (B,), device=dev), 2).float()


def step():
    opt.zero_grad(set_to_none=True)
    digits, rec_a, rec_b, _, _ = model(x, w, training=True)
    target = (w > 0).float()
    loss = BaseGAttN.AE_classify_loss(digits, y, target, rec_a, target, rec_b, 1.0) + l2()
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
