import cProfile, os, pstats, torch, numpy as np, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gat_amd import ops
from gat_amd.graph import MaskLists
dev = torch.device('cuda:0')
B, N, F, H, D = 128, 248, 64, 8, 8
rng = np.random.default_rng(0)
adj = (rng.random((B, N, N)) < 0.1) | np.eye(N, dtype=bool)[None]
bm = torch.tensor(np.where(adj, 0.0, -1e9).astype(np.float32), device=dev)
x = torch.randn(B, N, F, device=dev, requires_grad=True)
p = {k: torch.randn(*s, device=dev, requires_grad=True) for k, s in dict(W=(F, H, D), a1=(H, D), a2=(H, D), b1=(H,), b2=(H,), bias=(H, D)).items()}
cfg = ops.make_cfg(H, D, 'elu', mask_path='lists')
dout = torch.randn(B, N, H * D, device=dev)
def step():
    out = ops.dense_gat_layer(x, p['W'], p['a1'], p['a2'], p['b1'], p['b2'], p['bias'], bm, cfg)
    out.backward(dout)
for _ in range(20): step()
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print('ms/step', (time.perf_counter() - t) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
