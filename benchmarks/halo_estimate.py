"""Unique remote source rows (halo) of one products-sized shard of a world-times larger R-MAT
graph: what each rank receives per forward exchange and returns per backward exchange."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gat_amd import synth
dev=torch.device('cuda:0')
n_own, e_own = synth.PRODUCTS_NODES, synth.PRODUCTS_EDGES
for world in (2,4,8):
    n=n_own*world
    d,s=synth.rmat_graph(n, e_own, 4, dev, row_range=(0,n_own))
    rem=s[(s>=n_own)]
    u=torch.unique(rem)
    print(world, "edges", d.numel(), "remote edge frac %.3f"%(rem.numel()/d.numel()), "n_halo", u.numel(), "halo MB fwd %.0f"%(u.numel()*288/1e6))
    del d,s,rem,u
