"""Shared helpers of the parity tests: seeded inputs and the comparison rule."""
import numpy as np


def random_graph(n, avg_deg, seed, directed=True, hubs=0, hub_deg=0):
    """COO (dst, src) with distinct pairs and one self-loop per node (process.py:181 adds them)."""
    rng = np.random.default_rng(seed)
    m = int(n * avg_deg)
    dst = rng.integers(0, n, m)
    src = rng.integers(0, n, m)
    if hubs:
        hd = rng.integers(0, n, hubs)
        for h in hd:
            k = min(hub_deg, n)
            nb = rng.choice(n, size=k, replace=False)
            dst = np.concatenate([dst, np.full(k, h)])
            src = np.concatenate([src, nb])
            # and the transpose so that out-degrees are skewed too
            dst = np.concatenate([dst, nb])
            src = np.concatenate([src, np.full(k, h)])
    if not directed:
        dst, src = np.concatenate([dst, src]), np.concatenate([src, dst])
    dst = np.concatenate([dst, np.arange(n)])
    src = np.concatenate([src, np.arange(n)])
    key = np.unique(dst.astype(np.int64) * n + src)
    return (key // n).astype(np.int32), (key % n).astype(np.int32)


def layer_params(F, H, D, seed, res=False, dtype=np.float64):
    """Explicit weights (parity never depends on initialisers).  Biases are non-zero on purpose."""
    rng = np.random.default_rng(seed)
    lim_w = np.sqrt(6.0 / (F + D))
    lim_a = np.sqrt(6.0 / (D + 1))
    p = {
        "W": rng.uniform(-lim_w, lim_w, (F, H, D)),
        "a1": rng.uniform(-lim_a, lim_a, (H, D)),
        "a2": rng.uniform(-lim_a, lim_a, (H, D)),
        "b1": rng.normal(0, 0.1, H),
        "b2": rng.normal(0, 0.1, H),
        "bias": rng.normal(0, 0.1, (H, D)),
    }
    if res:
        p["W_res"] = rng.uniform(-lim_w, lim_w, (F, H, D))
        p["b_res"] = rng.normal(0, 0.1, (H, D))
    return {k: v.astype(dtype) for k, v in p.items()}


def assert_close(got, ref, rtol=1e-5, atol_scale=2e-6, what=""):
    """|got - ref| <= rtol*|ref| + atol_scale*max|ref| elementwise (fp32 kernels vs an fp64 oracle:
    the summation order differs, so an absolute term relative to the tensor's scale is needed)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (what, got.shape, ref.shape)
    scale = float(np.max(np.abs(ref))) if ref.size else 0.0
    err = np.abs(got - ref)
    tol = rtol * np.abs(ref) + atol_scale * scale + 1e-30
    bad = err > tol
    if bad.any():
        k = np.unravel_index(np.argmax(err / tol), err.shape)
        raise AssertionError("%s: %d/%d elements out of tolerance; worst at %s got %.9g ref %.9g (err %.3g, tol %.3g, scale %.3g)"
                             % (what, bad.sum(), bad.size, k, got[k], ref[k], err[k], tol[k], scale))


# ---- host twins of the in-kernel dropout masks ---------------------------------------------------
FEAT_SALT = 0x5DEECE66D
_M64 = (1 << 64) - 1


def scramble_seed(seed):
    """Host twin of scramble_seed in gat_amd/csrc/device_common.h: the splitmix64 finaliser every kernel applies
    to its (stepped) seed before drawing masks from it."""
    z = int(seed) & _M64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & _M64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def keep_mask_host(seed, idx, keep):
    """Host twin of keep_mask in gat_amd/csrc/device_common.h (32-bit mixer of the folded (seed, index) pair, top 24 bits)."""
    idx = np.asarray(idx).astype(np.uint64)
    seed = scramble_seed(seed)
    m32 = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        lo = (idx & m32) ^ np.uint64(seed & 0xFFFFFFFF)
        hi = (idx >> np.uint64(32)) ^ np.uint64(seed >> 32)
        x = (lo + hi * np.uint64(0x9E3779B9)) & m32
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x7FEB352D)) & m32
        x ^= x >> np.uint64(15)
        x = (x * np.uint64(0x846CA68B)) & m32
        x ^= x >> np.uint64(16)
    r = (x >> np.uint64(8)).astype(np.float32)
    return (r * np.float32(1.0 / 16777216.0)) < np.float32(keep)


def _mix32(x):
    m32 = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(16))
        x = (x * np.uint64(0x7FEB352D)) & m32
        x = x ^ (x >> np.uint64(15))
        x = (x * np.uint64(0x846CA68B)) & m32
        return x ^ (x >> np.uint64(16))


def feat_keep_host(seed, elem, keep, salt=0):
    """Host twin of drop_feat / drop_feat1 in gat_amd/csrc/device_common.h: element index e of the [rows, hd]
    table belongs to group e >> 2, 64 mixed bits per group, 16 bits per element against keep * 65536."""
    elem = np.asarray(elem).astype(np.uint64)
    s = (scramble_seed(seed) ^ salt ^ FEAT_SALT) & _M64
    m32 = np.uint64(0xFFFFFFFF)
    group = elem >> np.uint64(2)
    with np.errstate(over="ignore"):
        lo = _mix32((((group & m32) ^ np.uint64(s & 0xFFFFFFFF)) + ((group >> np.uint64(32)) ^ np.uint64(s >> 32)) * np.uint64(0x9E3779B9)) & m32)
        hi = _mix32((lo + np.uint64(0x9E3779B9)) & m32)
    z = (hi << np.uint64(32)) | lo
    field = (z >> (np.uint64(16) * (elem & np.uint64(3)))) & np.uint64(0xffff)
    return field < np.uint64(int(np.float32(keep) * np.float32(65536.0)))


IN_SALT = 0x9FB21C651E98DF25


def input_keep_host(seed, n, F, keep):
    """Host twin of drop_in4 in gat_amd/csrc/device_common.h (dropout of the layer's input inside the projection
    kernels): element (row, k) belongs to group row * (F / 4) + k / 4, field k & 3 -> bool [n, F]."""
    assert F % 4 == 0
    rows = np.arange(n, dtype=np.uint64)[:, None]
    k = np.arange(F, dtype=np.uint64)[None, :]
    elem = (rows * np.uint64(F // 4) + (k >> np.uint64(2))) * np.uint64(4) + (k & np.uint64(3))
    return feat_keep_host(seed, elem, keep, salt=IN_SALT)


def sparse_dropout_masks(seed, n_entries, n, H, D, dp, coef_drop, in_drop):
    """Masks the CSR kernels draw: coefficients keyed on (CSR position, head) -> [E,H]; projected
    features keyed on (source, padded column) -> [n,H,D].  None where the rate is 0."""
    coef = ft = None
    if coef_drop > 0:
        idx = np.arange(n_entries, dtype=np.uint64)[:, None] * np.uint64(H) + np.arange(H, dtype=np.uint64)[None, :]
        coef = keep_mask_host(seed, idx, 1.0 - coef_drop).astype(np.float64)
    if in_drop > 0:
        cols = (np.arange(H)[:, None] * dp + np.arange(D)[None, :]).astype(np.uint64)
        idx = np.arange(n, dtype=np.uint64)[:, None, None] * np.uint64(H * dp) + cols[None]
        ft = feat_keep_host(seed, idx, 1.0 - in_drop).astype(np.float64)
    return coef, ft


def dense_dropout_masks(seed, B, N, H, D, dp, coef_drop, in_drop):
    """Masks the dense kernels draw: coefficients keyed on (graph, i, j, head) -> [B,H,N,N];
    projected features keyed on (graph*N + node, padded column) -> [B,N,H,D]."""
    coef = ft = None
    if coef_drop > 0:
        b, i, j, h = np.meshgrid(np.arange(B), np.arange(N), np.arange(N), np.arange(H), indexing="ij")
        idx = (((b * N + i) * N + j) * H + h).astype(np.uint64)
        coef = keep_mask_host(seed, idx, 1.0 - coef_drop).astype(np.float64).transpose(0, 3, 1, 2)
    if in_drop > 0:
        cols = (np.arange(H)[:, None] * dp + np.arange(D)[None, :]).astype(np.uint64)
        idx = np.arange(B * N, dtype=np.uint64)[:, None, None] * np.uint64(H * dp) + cols[None]
        ft = feat_keep_host(seed, idx, 1.0 - in_drop).astype(np.float64).reshape(B, N, H, D)
    return coef, ft


# ---- multi-process tests -------------------------------------------------------------------------
def free_port():
    """A TCP port nobody listens on right now (bound to port 0 and released)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_ranks(worker, world, args=(), timeout=300):
    """Start `worker(rank, world, port, *args, q)` in `world` spawned processes and collect one
    (rank, message) per process from the queue.  Whatever happens (a worker that dies without
    reporting, a timeout), every process still alive afterwards is terminated and joined, so no rank
    stays behind blocked in a collective with the GPU or the port held."""
    import queue
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    try:
        for _ in procs:
            try:
                results.append(q.get(timeout=timeout))
            except queue.Empty:
                dead = [i for i, p in enumerate(procs) if not p.is_alive()]
                raise AssertionError("no report within %d s (%d of %d reported; exited ranks: %s, exit codes %s)"
                                     % (timeout, len(results), world, dead, [procs[i].exitcode for i in dead]))
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
                if p.is_alive():
                    p.kill()
                    p.join(timeout=10)
    for rank, msg in results:
        assert msg == "ok", "rank %s: %s" % (rank, msg)
    return results
