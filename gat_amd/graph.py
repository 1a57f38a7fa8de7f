"""Graph handle for the sparse attention kernels: destination-major CSR, its CSC transpose and the
length-sorted work-item lists both row kernels walk.

Orientation follows the reference: entry (i, j) lets destination i attend to source j, exactly
like bias_mat[i][j] == 0 in utils/layers.py:16-17.  The reference's sparse triple
(utils/process.py:186-188) stores indices[:, 0] = SparseTensor row = i and indices[:, 1] = j.
"""
import ctypes

import numpy as np
import torch

from . import _lib

DEFAULT_CHUNK = 0          # 0 = choose from the entry count (auto_chunk)
MIN_CHUNK, MAX_CHUNK = 256, 4096
RESIDENT_ROW_GROUPS = 32768   # 256 CUs x 32 waves x 4 row groups of 16 lanes
SHARED_BIAS_LIMIT = 16.0      # from_bias_mat: largest |bias| shared by a row's kept entries (fp32 spacing there: 2e-6)


def auto_chunk(nnz):
    """Longest piece a row group may own: about one resident group's share of the entries, so that
    no single piece outlasts the rest of the launch (measured on a 64 M-entry R-MAT graph: 2048-4096
    is fastest, 8192 already leaves a tail), rounded to a power of two inside [256, 4096]."""
    c = MIN_CHUNK
    while c < MAX_CHUNK and c * RESIDENT_ROW_GROUPS < nnz:
        c *= 2
    return c


def _work_items(ptr_host, chunk, order="length"):
    """gatmi_work_items on a host int64 pointer array -> (items [n_items,4] int32, split [n_split,4] int32, n_slots).
    order="rows" keeps the rows in their own order instead of longest first (possible when no row is
    longer than the chunk): neighbouring row groups then read neighbouring sources, which is what
    a batch of small graphs wants; a power-law graph wants the length sort."""
    n = len(ptr_host) - 1
    ptr_host = np.ascontiguousarray(ptr_host, dtype=np.int64)
    if n >= 2 ** 31 or (n > 0 and int(ptr_host[-1]) >= 2 ** 31):
        raise _lib.GatmiError("work items: %d rows / %d entries do not fit the kernels' 32-bit offsets "
                              "(one GPU's shard must stay below 2^31 of each)" % (n, int(ptr_host[-1]) if n > 0 else 0))
    if order == "rows" and n > 0 and int(np.diff(ptr_host).max()) <= chunk:
        items = np.empty((n, 4), dtype=np.int32)
        items[:, 0] = np.arange(n, dtype=np.int32)
        items[:, 1] = ptr_host[:-1]
        items[:, 2] = np.diff(ptr_host)
        items[:, 3] = -1
        return items, np.empty((0, 4), dtype=np.int32), 0
    n_items, n_split, n_slots = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    _lib.call("gatmi_work_items_count", n, _lib.ptr(ptr_host), chunk,
              ctypes.byref(n_items), ctypes.byref(n_split), ctypes.byref(n_slots))
    items = np.empty((n_items.value, 4), dtype=np.int32)
    split = np.empty((n_split.value, 4), dtype=np.int32)
    _lib.call("gatmi_work_items", n, _lib.ptr(ptr_host), chunk, _lib.ptr(items), _lib.ptr(split))
    return items, split, n_slots.value


def csr_from_coo_host(n, dst, src, add_self_loops=False, dedup=True):
    """Host COO -> CSR through the C ABI.  Returns (rowptr int64 [n+1], col int32 [E], perm int64 [E])."""
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    src = np.ascontiguousarray(src, dtype=np.int32)
    nnz = len(dst)
    cap = nnz + (n if add_self_loops else 0)
    rowptr = np.empty(n + 1, dtype=np.int64)
    col = np.empty(max(cap, 1), dtype=np.int32)
    perm = np.empty(max(cap, 1), dtype=np.int64)
    out = ctypes.c_int64()
    _lib.call("gatmi_csr_from_coo", n, nnz, _lib.ptr(dst), _lib.ptr(src), int(add_self_loops), int(dedup), cap,
              _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(perm), ctypes.byref(out))
    return rowptr, col[:out.value].copy(), perm[:out.value].copy()


def csc_from_csr_host(n_rows, n_cols, rowptr, col):
    """Host CSR -> CSC through the C ABI.  Returns (colptr int64, rowidx int32, perm int32)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    nnz = int(rowptr[-1])
    colptr = np.empty(n_cols + 1, dtype=np.int64)
    rowidx = np.empty(max(nnz, 1), dtype=np.int32)
    perm = np.empty(max(nnz, 1), dtype=np.int32)
    _lib.call("gatmi_csc_from_csr", n_rows, n_cols, _lib.ptr(rowptr), _lib.ptr(col),
              _lib.ptr(colptr), _lib.ptr(rowidx), _lib.ptr(perm))
    return colptr, rowidx[:nnz].copy(), perm[:nnz].copy()


class CSRGraph:
    """Device-resident adjacency of one (shard of a) graph.

    n_dst rows (destinations) x n_src columns (sources).  On a single GPU n_dst == n_src; under
    row partitioning n_src = owned + halo rows.
    """

    def __init__(self, n_dst, n_src, rowptr, col, colptr, rowidx, perm, device, chunk=DEFAULT_CHUNK, order="length"):
        self.n_dst, self.n_src = int(n_dst), int(n_src)
        self.device = torch.device(device)
        self.rowptr_host = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.colptr_host = np.ascontiguousarray(colptr, dtype=np.int64)
        self.nnz = int(self.rowptr_host[-1])
        self.chunk = int(chunk) if chunk else auto_chunk(self.nnz)
        to = lambda a, dt: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(self.device, dt).contiguous()
        self.col = to(col, torch.int32)
        self.rowidx = to(rowidx, torch.int32)
        self.perm = to(perm, torch.int32)
        it, sp, self.n_slots_csr = _work_items(self.rowptr_host, self.chunk, order)
        self.items_csr, self.split_csr = to(it, torch.int32), to(sp, torch.int32)
        it, sp, self.n_slots_csc = _work_items(self.colptr_host, self.chunk, order)
        self.items_csc, self.split_csc = to(it, torch.int32), to(sp, torch.int32)

    # ---- constructors -------------------------------------------------------------------------
    @classmethod
    def from_coo(cls, dst, src, n, device, add_self_loops=False, dedup=True, chunk=DEFAULT_CHUNK):
        """Host path: integer work done by the C++ builder of libgatmi (bit-exact, deterministic)."""
        rowptr, col, _ = csr_from_coo_host(n, dst, src, add_self_loops, dedup)
        colptr, rowidx, perm = csc_from_csr_host(n, n, rowptr, col)
        return cls(n, n, rowptr, col, colptr, rowidx, perm, device, chunk)

    @classmethod
    def from_reference_triple(cls, adj_mat, device, chunk=DEFAULT_CHUNK):
        """adj_mat = (indices [E,2], values [E], shape) as returned by preprocess_adj_bias
        (utils/process.py:188) and fed to tf.sparse_placeholder (execute_cora_sparse.py:71,115)."""
        indices, _values, shape = adj_mat
        indices = np.asarray(indices)
        assert shape[0] == shape[1], "adjacency must be square"
        return cls.from_coo(indices[:, 0], indices[:, 1], int(shape[0]), device, False, True, chunk)

    @classmethod
    def from_csr(cls, rowptr, col, n_src, device, chunk=DEFAULT_CHUNK):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        n_dst = len(rowptr) - 1
        colptr, rowidx, perm = csc_from_csr_host(n_dst, n_src, rowptr, col)
        return cls(n_dst, n_src, rowptr, col, colptr, rowidx, perm, device, chunk)

    @classmethod
    def from_device_coo(cls, dst, src, n_dst, n_src, chunk=DEFAULT_CHUNK, item_order="length"):
        """Large graphs already on the GPU: dst/src int64 device tensors with DISTINCT (dst, src)
        pairs.  Sorting runs on the device; only the two pointer arrays visit the host (for the
        work-item builder).  Produces the same arrays as the host path."""
        dev = dst.device
        if max(n_dst, n_src) >= 2 ** 31 or dst.numel() >= 2 ** 31:
            raise _lib.GatmiError("from_device_coo: %d x %d nodes / %d entries do not fit the kernels' 32-bit ids and "
                                  "offsets" % (n_dst, n_src, dst.numel()))
        key = dst * n_src + src
        key, _ = torch.sort(key)
        d = torch.div(key, n_src, rounding_mode="floor")
        col = (key - d * n_src).to(torch.int32)
        rowptr = torch.zeros(n_dst + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(d, minlength=n_dst), 0)
        # CSC: stable sort of the CSR entries by source keeps destinations ascending inside a column
        _, order = torch.sort(col.to(torch.int64), stable=True)
        rowidx = d[order].to(torch.int32)
        perm = order.to(torch.int32)
        colptr = torch.zeros(n_src + 1, dtype=torch.int64, device=dev)
        colptr[1:] = torch.cumsum(torch.bincount(col.to(torch.int64), minlength=n_src), 0)
        del key, d, order
        return cls(n_dst, n_src, rowptr.cpu().numpy(), col, colptr.cpu().numpy(), rowidx, perm, dev, chunk, item_order)

    @classmethod
    def from_bias_mat(cls, bias_mat, cutoff=1.0e6):
        """A batch of dense additive masks [B,N,N] (adj_to_bias, utils/process.py:14-25) as ONE
        block-diagonal graph over B*N nodes: entry (b,i,j) is kept iff it lies within `cutoff` of its
        row's maximum (anything lower has a softmax coefficient of exactly 0 in fp32).  Build it
        once for a mask that is used step after step and pass it where bias_mat went: the layer then
        runs the CSR kernels and never reads the N x N floats again.  Only for hard masks: the kept
        entries of a row must all carry the same bias (0 in an adj_to_bias mask, +1e9 on the diagonal
        of a padded node: a constant shift of a row's logits cancels in its softmax); biases that
        differ inside a row would have to be added to the logits, which the CSR kernels do not do --
        ValueError then.  ValueError too for a row that keeps several entries under one LARGE bias (a
        row masked everywhere): in fp32 the bias swallows the logits and the row attends uniformly,
        which the CSR kernels would not reproduce."""
        if bias_mat.dim() != 3 or bias_mat.shape[1] != bias_mat.shape[2] or not bias_mat.is_cuda:
            raise ValueError("from_bias_mat: a [B,N,N] device tensor is expected")
        B, N, _ = bias_mat.shape
        rowmax = bias_mat.amax(-1, keepdim=True)
        keep = bias_mat > rowmax - cutoff
        if bool(((bias_mat != rowmax) & keep).any()):
            raise ValueError("from_bias_mat: the mask carries different biases on the entries of a row that "
                             "contribute; keep it dense (attn_head adds them to the logits)")
        # A shared bias cancels in the softmax only while fp32 can still resolve the logits next to it:
        # LeakyReLU(z) + bias is rounded to the spacing of fp32 at |bias| (utils/layers.py:17), and a row
        # that is masked everywhere (-1e9 in every column) swallows its logits and attends uniformly.
        # Rows that keep a single entry (a padded node's +1e9 diagonal) have coefficient 1 either way.
        if bool(((rowmax.abs() > SHARED_BIAS_LIMIT) & (keep.sum(-1, keepdim=True) > 1)).any()):
            raise ValueError("from_bias_mat: a row keeps several entries under a shared bias beyond %g in magnitude "
                             "(e.g. a row masked everywhere): fp32 rounds its logits away, which only the dense "
                             "kernels reproduce; keep it dense" % SHARED_BIAS_LIMIT)
        b, i, j = keep.nonzero(as_tuple=True)
        del keep
        base = b * N
        return cls.from_device_coo(base + i, base + j, B * N, B * N, max(MIN_CHUNK, 1 << (N - 1).bit_length()), item_order="rows")

    # ---- helpers ------------------------------------------------------------------------------
    def in_degrees(self):
        return np.diff(self.rowptr_host)

    def out_degrees(self):
        return np.diff(self.colptr_host)

    def __repr__(self):
        return "CSRGraph(n_dst=%d, n_src=%d, nnz=%d, chunk=%d, split rows csr/csc=%d/%d)" % (
            self.n_dst, self.n_src, self.nnz, self.chunk, len(self.split_csr), len(self.split_csc))


class MaskLists:
    """Prebuilt lists of a batch of dense additive masks [B,N,N] (adj_to_bias, utils/process.py:14-25) for the
    graph-resident kernels: per graph, the columns every row visits and the rows that visit every column, in the
    order in which the mask-reading kernels compact them, plus the graph's rows / columns ordered by list length.
    Built once (a few launches and two prefix sums) for a mask that is fed unchanged step after step — the reference
    feeds one bias_mat every epoch, execute_cora.py:100-107 — and passed where bias_mat went: the layer then neither
    reads the N x N floats nor compacts anything, and its results are bit-identical to the mask-reading kernels'.
    Hard masks only (every visited entry of a row carries the row's maximum, utils/layers.py:17 adds a constant
    that cancels in the softmax): ValueError otherwise, as for CSRGraph.from_bias_mat."""

    def __init__(self, bias_mat):
        if bias_mat.dim() != 3 or bias_mat.shape[1] != bias_mat.shape[2] or not bias_mat.is_cuda or bias_mat.dtype != torch.float32:
            raise ValueError("MaskLists: a [B,N,N] fp32 device tensor is expected")
        bias_mat = bias_mat.contiguous()
        B, N, _ = bias_mat.shape
        dev = bias_mat.device
        if B * N >= 2 ** 31 - 1 or not (4 <= N <= 320):
            raise ValueError("MaskLists: graphs of 4..320 nodes are covered, got N = %d" % N)
        st = _lib.current_stream()
        W = _lib.lib.gatmi_dense_mask_words(N)
        bits = torch.empty((B * N, W), dtype=torch.int32, device=dev)
        _lib.call("gatmi_dense_mask_bits", B, N, _lib.ptr(bias_mat), SHARED_BIAS_LIMIT, _lib.ptr(bits), st)
        if bool(bits[:, W - 1].any()):
            raise ValueError("MaskLists: some row's visited entries carry biases that matter to its softmax (they differ "
                             "inside the row, or a large bias is shared by several entries); keep the mask dense")
        row_cnt = torch.empty(B * N, dtype=torch.int32, device=dev)
        col_cnt = torch.empty(B * N, dtype=torch.int32, device=dev)
        _lib.call("gatmi_dense_lists_count", B, N, _lib.ptr(bits), _lib.ptr(row_cnt), _lib.ptr(col_cnt), st)

        def prefix(cnt):
            p = torch.zeros(B * N + 1, dtype=torch.int64, device=dev)
            p[1:] = torch.cumsum(cnt, 0, dtype=torch.int64)
            if int(p[-1]) >= 2 ** 31 - 1:
                raise ValueError("MaskLists: more than 2^31 list entries")
            per_graph = p[N::N] - p[:-1:N]
            order = torch.argsort(cnt.view(B, N), dim=1, descending=True, stable=True).to(torch.int16)
            return p.to(torch.int32), int(per_graph.max()), order.reshape(-1).contiguous()
        self.lptr, self.max_fwd, self.lord = prefix(row_cnt)
        self.cptr, self.max_bwd, self.cord = prefix(col_cnt)
        self.lcol = torch.empty(max(int(self.lptr[-1]), 4), dtype=torch.int16, device=dev)
        self.crow = torch.empty(max(int(self.cptr[-1]), 4), dtype=torch.int16, device=dev)
        _lib.call("gatmi_dense_lists_fill", B, N, _lib.ptr(bits), _lib.ptr(self.lptr), _lib.ptr(self.cptr), _lib.ptr(self.lcol),
                  _lib.ptr(self.crow), st)
        self.B, self.N, self.device = B, N, dev
        self.entries = int(row_cnt.sum())

    def fits(self, H, dp):
        """Whether the graph-resident kernels can hold every graph's lists beside its rows for heads of this shape."""
        cap = _lib.lib.gatmi_dense_lists_capacity
        return (self.B >= 256 and 0 < self.max_fwd <= cap(self.N, H, dp, 0) and 0 < self.max_bwd <= cap(self.N, H, dp, 1))

    def __repr__(self):
        return "MaskLists(B=%d, N=%d, list entries=%d, largest graph fwd/bwd=%d/%d)" % (self.B, self.N, self.entries, self.max_fwd, self.max_bwd)
