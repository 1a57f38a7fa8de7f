"""Sparse node features on the device: the CSR of X [n, F] for the projection Wh = X.W and the
CSR of X^T for the weight gradient dW = X^T.dWh.

The reference carries sparse features as the `(coords, values, shape)` tuple that
`preprocess_features` returns beside the dense matrix (utils/process.py:155-162, tuple layout from
`sparse_to_tuple` :125-141: coords [nnz,2] = (row, col)).  Cora's [2708,1433] feature matrix is
98.7 % zeros, so the first layer's projection is a few dozen multiply-adds per node instead of a
dense GEMM.  Index preparation happens once on the host with scipy, as the reference does.
"""
import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from ._lib import ptr


class SparseFeatures:
    """rowptr int64 [n+1], col int32 [nnz], val fp32 [nnz] (rows sorted by column) and the
    transpose: t_rowptr int64 [F+1], t_row int32 [nnz], t_perm int64 [nnz] (position of each
    transposed entry in `val`), all on `device`."""

    def __init__(self, rowptr, col, val, t_rowptr, t_row, t_perm, shape):
        self.rowptr, self.col, self.val = rowptr, col, val
        self.t_rowptr, self.t_row, self.t_perm = t_rowptr, t_row, t_perm
        self.shape = (int(shape[0]), int(shape[1]))

    # tensor-like surface used by the layer wrappers
    @property
    def device(self):
        return self.val.device

    @property
    def is_cuda(self):
        return self.val.is_cuda

    def dim(self):
        return 2

    @property
    def nnz(self):
        return int(self.val.numel())

    @classmethod
    def from_scipy(cls, mx, device):
        csr = sp.csr_matrix(mx, dtype=np.float32)
        csr.sum_duplicates()
        csr.sort_indices()
        n, f = csr.shape
        if csr.nnz >= 2 ** 31 or n >= 2 ** 31 or f >= 2 ** 31:
            raise ValueError("SparseFeatures: sizes must fit int32 ids")
        # transpose by a stable sort on the column id: entries of one feature stay in node order
        order = np.argsort(csr.indices, kind="stable").astype(np.int64)
        rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(csr.indptr))
        t_rowptr = np.zeros(f + 1, dtype=np.int64)
        np.cumsum(np.bincount(csr.indices, minlength=f), out=t_rowptr[1:])
        dev = torch.device(device)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev)
        return cls(t(csr.indptr, torch.int64), t(csr.indices, torch.int32), t(csr.data, torch.float32),
                   t(t_rowptr, torch.int64), t(rows[order], torch.int32), t(order, torch.int64), (n, f))

    @classmethod
    def from_tuple(cls, tup, device):
        """The reference's sparse tuple (coords [nnz,2] (row,col), values [nnz], shape)."""
        coords, values, shape = tup
        coords = np.asarray(coords)
        mx = sp.coo_matrix((np.asarray(values, dtype=np.float32), (coords[:, 0], coords[:, 1])), shape=tuple(shape))
        return cls.from_scipy(mx, device)

    @classmethod
    def from_dense(cls, x, device=None):
        x = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        return cls.from_scipy(sp.csr_matrix(x.reshape(-1, x.shape[-1])), device or "cuda:0")

    def with_values(self, val):
        """Same sparsity pattern, other stored values (e.g. after dropout)."""
        return SparseFeatures(self.rowptr, self.col, val.contiguous(), self.t_rowptr, self.t_row, self.t_perm, self.shape)

    @property
    def t_val(self):
        return self.val.index_select(0, self.t_perm)

    def row_normalize(self):
        """preprocess_features (utils/process.py:155-162) on the device: every row divided by its
        sum (rows that sum to 0 stay 0), computed by one kernel thread per row in stored order."""
        out = torch.empty_like(self.val)
        _lib.call("gatmi_csr_row_normalize", self.shape[0], ptr(self.rowptr), ptr(self.val), ptr(out), _lib.current_stream())
        return self.with_values(out)

    def dropout(self, rate):
        """tf.nn.dropout on the features (utils/layers.py:8-9) restricted to the stored entries:
        dropping or rescaling a zero leaves it zero."""
        return self.with_values(torch.nn.functional.dropout(self.val, rate, True))

    def to_dense(self):
        n, f = self.shape
        rows = torch.repeat_interleave(torch.arange(n, device=self.device), self.rowptr[1:] - self.rowptr[:-1])
        out = torch.zeros((n, f), dtype=torch.float32, device=self.device)
        out[rows, self.col.long()] = self.val
        return out
