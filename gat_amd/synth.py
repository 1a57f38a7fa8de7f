"""Seeded synthetic inputs shaped like the benchmark configurations (no datasets ship with the
reference: utils/process.py:50 reads ../data/, which is absent; its own stand-in is
load_random_data, process.py:98-123).  All generators run on the device they are given.
"""
import math

import torch

RMAT_ABCD = (0.57, 0.19, 0.19, 0.05)

# ogbn-products shape used by the single-GPU benchmark
PRODUCTS_NODES = 2_449_029
PRODUCTS_EDGES = 61_859_140


def _rmat_batch(m, scale, gen, device, abcd):
    a, b, c, _ = abcd
    dst = torch.zeros(m, dtype=torch.int64, device=device)
    src = torch.zeros(m, dtype=torch.int64, device=device)
    for _ in range(scale):
        r = torch.rand(m, device=device, generator=gen)
        dbit = (r >= a + b).to(torch.int64)                       # quadrants c, d: lower half
        sbit = ((r >= a) & (r < a + b) | (r >= a + b + c)).to(torch.int64)   # quadrants b, d: right half
        dst = dst * 2 + dbit
        src = src * 2 + sbit
    return dst, src


def rmat_graph(n, n_edges, seed, device, abcd=RMAT_ABCD, row_range=None):
    """Exactly n_edges distinct directed (dst, src) pairs with dst != src, ids below n, R-MAT
    distributed, plus one self-loop per node (the reference adds A + I, process.py:181).
    row_range=(lo, hi) keeps only destinations in [lo, hi) (one partition's rows) and then
    n_edges counts the entries of that range.  Returns int64 device tensors (dst, src)."""
    scale = max(1, math.ceil(math.log2(max(n, 2))))
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    lo, hi = (0, n) if row_range is None else row_range
    mult = 1_000_003
    while math.gcd(mult, n) != 1:
        mult += 2
    keys = torch.empty(0, dtype=torch.int64, device=device)
    batch = max(1 << 16, int(n_edges * 0.35))
    while keys.numel() < n_edges:
        dst, src = _rmat_batch(batch, scale, gen, device, abcd)
        if row_range is not None:
            # one partition's rows: destinations are folded into the range; source ids go through a
            # fixed affine bijection of [0, n) so that the R-MAT hubs (the low ids) are spread over
            # all owners instead of piling up on the first partition (a random node labelling)
            dst = lo + dst % (hi - lo)
            src = torch.where(src < n, (src * mult + 12345) % n, src)
        ok = (dst < n) & (src < n) & (dst != src)
        k = dst[ok] * n + src[ok]
        keys = torch.unique(torch.cat([keys, k]))
        del dst, src, ok, k
    if keys.numel() > n_edges:
        pick = torch.randperm(keys.numel(), device=device, generator=gen)[:n_edges]
        keys = keys[pick]
    dst = torch.div(keys, n, rounding_mode="floor")
    src = keys - dst * n
    loops = torch.arange(lo, hi, dtype=torch.int64, device=device)
    return torch.cat([dst, loops]), torch.cat([src, loops])


def glorot_layer(f_in, heads, width, seed, device, zero_bias=True):
    """Weights of one layer: W ~ U(+-sqrt(6/(f_in+width))), a1, a2 ~ U(+-sqrt(6/(width+1)))
    (the Glorot-uniform default of tf.layers.conv1d); biases zero as TF initialises them, or
    small normal values when zero_bias is false."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    u = lambda shape, lim: (torch.rand(shape, device=device, generator=gen) * 2 - 1) * lim
    nrm = lambda shape: torch.randn(shape, device=device, generator=gen) * 0.1
    p = {
        "W": u((f_in, heads, width), math.sqrt(6.0 / (f_in + width))),
        "a1": u((heads, width), math.sqrt(6.0 / (width + 1))),
        "a2": u((heads, width), math.sqrt(6.0 / (width + 1))),
    }
    for name, shape in (("b1", (heads,)), ("b2", (heads,)), ("bias", (heads, width))):
        p[name] = torch.zeros(shape, device=device) if zero_bias else nrm(shape)
    return p


def normal(shape, seed, device):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    return torch.randn(shape, device=device, generator=gen)
