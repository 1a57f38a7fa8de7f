"""Host-side preprocessing with the reference's names and outputs (utils/process.py).

These stay Python/numpy like the reference's; they produce, bit for bit, what the reference's
functions produce (pinned by tests/golden/preproc_*.npz, which were written by the reference's
own code).  The structures the kernels walk (CSR, CSC, work items) are then built from these
outputs by the C++ builder in libgatmi (gat_amd.graph).
"""
import numpy as np
import scipy.sparse as sp


def adj_to_bias(adj, sizes, nhood=1):
    """utils/process.py:14-25.  adj [G,N,N] -> float64 [G,N,N] additive mask: 0 (stored as -0.0)
    where node j is within nhood hops of node i (self included), -1e9 elsewhere.

    Only the leading sizes[g] x sizes[g] block of graph g is binarised; entries outside it keep the
    raw value of (adj+I)^nhood, as in the reference, whose clamp loop (process.py:21-24) stops at
    sizes[g].  The reference's 7.3-million-iteration Python loop at Cora size is replaced by one
    vectorised comparison per graph."""
    adj = np.asarray(adj)
    nb_graphs, n = adj.shape[0], adj.shape[1]
    eye = np.eye(n)
    mt = np.empty(adj.shape)
    for g in range(nb_graphs):
        m = eye.copy()
        step = np.asarray(adj[g]) + eye
        for _ in range(nhood):
            m = np.matmul(m, step)
        s = int(sizes[g])
        block = m[:s, :s]
        block[block > 0.0] = 1.0
        mt[g] = m
    return -1e9 * (1.0 - mt)


def preprocess_adj_bias(adj):
    """utils/process.py:179-188.  scipy sparse adjacency -> (indices [E,2] int32, values [E]
    float32 ones, shape): self-loops added, positive entries binarised, COO order of scipy, and
    indices emitted as (col, row) exactly like the reference (process.py:186)."""
    num_nodes = adj.shape[0]
    a = adj + sp.eye(num_nodes)
    a[a > 0.0] = 1.0
    if not sp.isspmatrix_coo(a):
        a = a.tocoo()
    a = a.astype(np.float32)
    indices = np.vstack((a.col, a.row)).transpose()
    return indices, a.data, a.shape


def sparse_to_tuple(sparse_mx):
    """utils/process.py:125-141: (coords [nnz,2] (row,col), values, shape) of a scipy matrix."""
    def to_tuple(mx):
        if not sp.isspmatrix_coo(mx):
            mx = mx.tocoo()
        return np.vstack((mx.row, mx.col)).transpose(), mx.data, mx.shape
    if isinstance(sparse_mx, list):
        return [to_tuple(m) for m in sparse_mx]
    return to_tuple(sparse_mx)


def preprocess_features(features):
    """utils/process.py:155-162: row-normalise (D^-1 X, rows summing to 0 stay 0); returns the
    dense matrix and the sparse tuple."""
    rowsum = np.array(features.sum(1))
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    features = sp.diags(r_inv).dot(features)
    return features.todense(), sparse_to_tuple(features)


def load_random_data(size, rng=None):
    """utils/process.py:98-123: the reference's Cora-like random stand-in (adjacency density
    0.002, 1000 features at density 0.015, 7 classes, first half train / second half val+test).
    rng: a numpy RandomState for reproducibility (the reference uses the global state)."""
    rng = rng or np.random.mtrand._rand
    adj = sp.random(size, size, density=0.002, random_state=rng)
    features = sp.random(size, 1000, density=0.015, random_state=rng)
    int_labels = rng.randint(7, size=size)
    labels = np.zeros((size, 7))
    labels[np.arange(size), int_labels] = 1
    train_mask = np.zeros(size, dtype=bool)
    train_mask[: int(size / 2)] = True
    val_mask = ~train_mask
    test_mask = val_mask.copy()
    y_train, y_val, y_test = np.zeros(labels.shape), np.zeros(labels.shape), np.zeros(labels.shape)
    y_train[train_mask] = labels[train_mask]
    y_val[val_mask] = labels[val_mask]
    y_test[test_mask] = labels[test_mask]
    return adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask


def pad_graphs(adjs, feats, labels=None):
    """utils/process_ppi.py:183-215,255-269: stack graphs of different sizes into one padded batch.

    Graph g keeps its adjacency in the leading sizes[g] x sizes[g] block; the padding is the
    identity (process_ppi.py:194), padded feature and label rows are zero, and the node mask is 1
    for real nodes.  Returns (adj [G,N,N] float64, feat [G,N,F] float64, sizes [G] int64,
    node_mask [G,N] float64[, labels [G,N,C] float64]) with N = max size.  Feeding adj and sizes to
    adj_to_bias gives the additive mask under which padded nodes attend to themselves only (their
    diagonal entry is (I·(I+I)) = 2, hence a bias of +1e9, every other entry of their row -1e9)."""
    sizes = np.array([a.shape[0] for a in adjs], dtype=np.int64)
    n = int(sizes.max())
    adj = np.empty((len(adjs), n, n))
    feat = np.zeros((len(adjs), n, feats[0].shape[1]))
    mask = np.zeros((len(adjs), n))
    lab = None if labels is None else np.zeros((len(adjs), n, labels[0].shape[1]))
    for g, a in enumerate(adjs):
        s = int(sizes[g])
        a = a.toarray() if sp.issparse(a) else np.asarray(a)
        block = np.identity(n)
        block[:s, :s] = a
        adj[g] = block
        f = feats[g]
        feat[g, :s] = f.toarray() if sp.issparse(f) else np.asarray(f)
        mask[g, :s] = 1.0
        if lab is not None:
            lab[g, :s] = labels[g]
    out = (adj, feat, sizes, mask)
    return out if lab is None else out + (lab,)
