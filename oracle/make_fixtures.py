"""Generate tests/golden/preproc_*.npz by running the reference's own utils/process.py.

Run only in the build container (needs /root/reference); the GPU box never sees the reference.
utils/process.py imports unchanged once the single dead legacy import at process.py:5
(`scipy.sparse.linalg.eigen.arpack.eigsh`, removed from modern scipy, unused by the functions
called here) is satisfied by an empty stand-in module.

Each fixture stores the INPUT graph and the reference's OUTPUTS:
  adj_dense            float64 [N,N]    input adjacency (weights kept)
  pab_indices/_data/_shape             preprocess_adj_bias(adj)         (process.py:179-188)
  bias_n1, bias_n2     float64 [1,N,N]  adj_to_bias(adj, [size], 1|2)   (process.py:14-25)
  size                 int              the `sizes[0]` used (smaller than N in some cases)
  feat_dense, feat_norm                 preprocess_features(features)[0] (process.py:155-162)
"""
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

REF = os.environ.get("GAT_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference_process():
    for name in ("scipy.sparse.linalg.eigen", "scipy.sparse.linalg.eigen.arpack"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["scipy.sparse.linalg.eigen.arpack"].eigsh = None
    sys.path.insert(0, REF)
    from utils import process  # noqa: E402
    return process


def cases():
    rng = np.random.default_rng(20240611)

    def rand_adj(n, density, symmetric, weighted, self_loops=False, isolated=None):
        a = (rng.random((n, n)) < density).astype(np.float64)
        if weighted:
            a = a * rng.random((n, n))
        np.fill_diagonal(a, 0.0)
        if symmetric:
            a = np.maximum(a, a.T)
        if self_loops:
            idx = rng.choice(n, size=max(1, n // 4), replace=False)
            a[idx, idx] = 1.0 if not weighted else 0.5
        if isolated is not None:
            a[isolated, :] = 0.0
            a[:, isolated] = 0.0
        return a

    yield "sym16", rand_adj(16, 0.2, True, False), 16
    yield "dir24", rand_adj(24, 0.15, False, False), 24
    yield "selfloops20", rand_adj(20, 0.2, True, False, self_loops=True), 20
    yield "weighted32", rand_adj(32, 0.15, False, True), 32
    yield "isolated18", rand_adj(18, 0.25, True, False, isolated=5), 18
    yield "border40", rand_adj(40, 0.1, False, True), 29      # sizes < N: untouched border
    yield "single1", np.zeros((1, 1)), 1


def main():
    process = import_reference_process()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(7)
    for name, a, size in cases():
        n = a.shape[0]
        indices, data, shape = process.preprocess_adj_bias(sp.csr_matrix(a))
        b1 = process.adj_to_bias(a[None].copy(), [size], nhood=1)
        b2 = process.adj_to_bias(a[None].copy(), [size], nhood=2)
        feat = sp.random(n, 12, density=0.3, random_state=np.random.RandomState(n), format="csr")
        if n > 2:
            feat = feat.tolil()
            feat[1, :] = 0.0          # an all-zero row: inf -> 0 branch of process.py:159
            feat = feat.tocsr()
        fnorm, _ = process.preprocess_features(feat)
        np.savez_compressed(
            os.path.join(OUT, "preproc_%s.npz" % name),
            adj_dense=a, size=np.int64(size),
            pab_indices=indices, pab_data=data, pab_shape=np.asarray(shape, dtype=np.int64),
            bias_n1=b1, bias_n2=b2,
            feat_dense=np.asarray(feat.todense()), feat_norm=np.asarray(fnorm))
        print(name, n, indices.shape, indices.dtype, data.dtype, b1.dtype)

    # the reference's own synthetic-input generator (process.py:98-123), seeded through numpy's
    # global state, which is what sp.random falls back to
    np.random.seed(1234)
    adj, features, *_ = process.load_random_data(200)
    indices, data, shape = process.preprocess_adj_bias(adj.tocsr())
    fnorm, _ = process.preprocess_features(features.tocsr())
    np.savez_compressed(
        os.path.join(OUT, "preproc_random200.npz"),
        adj_row=adj.row, adj_col=adj.col, adj_data=adj.data,
        pab_indices=indices, pab_data=data, pab_shape=np.asarray(shape, dtype=np.int64),
        feat_row=features.row, feat_col=features.col, feat_data=features.data,
        feat_norm=np.asarray(fnorm))
    print("random200", indices.shape)
    del rng


if __name__ == "__main__":
    main()
