"""CPU tests of the C ABI: the library loads, exports every symbol include/gatmi.h declares, and
its host integer routines (COO->CSR->CSC, work items, partitions, halo plans) are exact."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ref_numpy
from tests.util import random_graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gat_amd import _lib
    header = open(os.path.join(ROOT, "include", "gatmi.h")).read()
    declared = set(re.findall(r"\b(gatmi_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "libgatmi.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), "ctypes table and header disagree: %s" % (declared ^ set(_lib.EXPORTS))
    assert _lib.lib.gatmi_version() >= 100


def test_error_reporting():
    from gat_amd import _lib
    n = 4
    dst = np.array([0, 9], dtype=np.int32)     # 9 is out of range
    src = np.array([0, 1], dtype=np.int32)
    rowptr = np.empty(n + 1, dtype=np.int64)
    col = np.empty(8, dtype=np.int32)
    out = ctypes.c_int64()
    rc = _lib.lib.gatmi_csr_from_coo(n, 2, _lib.ptr(dst), _lib.ptr(src), 0, 0, 8, _lib.ptr(rowptr), _lib.ptr(col), None, ctypes.byref(out))
    assert rc == -5 and "outside" in _lib.last_error()
    with pytest.raises(_lib.GatmiError):
        _lib.call("gatmi_csr_from_coo", n, 2, _lib.ptr(dst), _lib.ptr(src), 0, 0, 1, _lib.ptr(rowptr), _lib.ptr(col), None, ctypes.byref(out))
    # unsupported head shape is reported, not crashed on (no GPU is touched before validation)
    rc = _lib.lib.gatmi_gat_fwd(1, 3, 8, None, 1, None, 0, None, None, 0, 24, None, None, None, None, None, None, 0, 0, 0.2, 1.0, 1.0, 0,
                                None, None, None, None, None, None, None)
    assert rc == -3 and "power of two" in _lib.last_error()


@pytest.mark.parametrize("seed,n,deg,directed", [(0, 1, 0.0, True), (1, 50, 3.0, True), (2, 200, 5.0, False), (3, 64, 0.3, True)])
def test_csr_csc_against_scipy(seed, n, deg, directed):
    from gat_amd import graph
    dst, src = random_graph(n, deg, seed, directed=directed)
    rowptr, col, perm = graph.csr_from_coo_host(n, dst, src)
    r2, c2 = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    assert (rowptr == r2).all() and (col == c2).all()
    assert (dst[perm] == np.repeat(np.arange(n), np.diff(rowptr))).all() and (src[perm] == col).all()
    m = sp.csr_matrix((np.ones(len(col)), col, rowptr), shape=(n, n))
    csc = m.tocsc()
    csc.sort_indices()
    colptr, rowidx, p2 = graph.csc_from_csr_host(n, n, rowptr, col)
    assert (colptr == csc.indptr).all() and (rowidx == csc.indices).all()
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    assert (rows[p2] == rowidx).all() and (col[p2] == np.repeat(np.arange(n), np.diff(colptr))).all()


def test_csr_self_loops_and_dedup():
    from gat_amd import graph
    dst = np.array([2, 0, 2, 2, 1, 0], dtype=np.int32)
    src = np.array([1, 2, 1, 0, 1, 2], dtype=np.int32)   # (2,1) and (0,2) twice, (1,1) is a self-loop
    rowptr, col, perm = graph.csr_from_coo_host(3, dst, src, add_self_loops=True, dedup=True)
    assert rowptr.tolist() == [0, 2, 3, 6]
    assert col.tolist() == [0, 2, 1, 0, 1, 2]
    assert perm.tolist() == [-1, 1, 4, 3, 0, -1]
    rowptr, col, _ = graph.csr_from_coo_host(3, dst, src, add_self_loops=False, dedup=False)
    assert col.tolist() == [2, 2, 1, 0, 1, 1]


def test_reference_triple_round_trip():
    """CSR built from the reference's (col,row) indices equals the CSR of (A+I > 0) transposed the
    way the reference feeds it to tf.SparseTensor (indices[:,0] is the row = destination)."""
    from gat_amd import graph, preprocess
    rng = np.random.default_rng(5)
    a = (rng.random((30, 30)) < 0.1).astype(np.float64)
    idx, vals, shape = preprocess.preprocess_adj_bias(sp.csr_matrix(a))
    rowptr, col, _ = graph.csr_from_coo_host(30, idx[:, 0], idx[:, 1])
    want = sp.csr_matrix(((a + np.eye(30)) > 0).astype(np.float64).T)
    want.sort_indices()
    assert (rowptr == want.indptr).all() and (col == want.indices).all()


def test_work_items_cover_every_entry_once():
    from gat_amd import graph
    dst, src = random_graph(300, 2.0, 4, hubs=3, hub_deg=250)
    rowptr, col, _ = graph.csr_from_coo_host(300, dst, src)
    for chunk in (1, 7, 64, 1000):
        items, split, n_slots = graph._work_items(rowptr, chunk)
        lens = items[:, 2]
        assert (lens[:-1] >= lens[1:]).all() and lens.max() <= chunk
        covered = np.zeros(rowptr[-1], dtype=np.int64)
        for row, beg, ln, slot in items:
            assert rowptr[row] <= beg and beg + ln <= rowptr[row + 1]
            covered[beg:beg + ln] += 1
        assert (covered == 1).all()
        whole = items[items[:, 3] < 0]
        assert (np.diff(rowptr)[whole[:, 0]] == whole[:, 2]).all()
        pieces = items[items[:, 3] >= 0]
        assert sorted(pieces[:, 3].tolist()) == list(range(n_slots))
        for row, s0, ns, _ in split:
            assert ns == -(-(rowptr[row + 1] - rowptr[row]) // chunk) and ns > 1
            mine = pieces[(pieces[:, 3] >= s0) & (pieces[:, 3] < s0 + ns)]
            assert (mine[:, 0] == row).all() and mine[:, 2].sum() == rowptr[row + 1] - rowptr[row]
        assert len(items) == 300 - len(split) + n_slots
        # rows with no entries still get an item (their output is the bias)
        assert set(items[:, 0].tolist()) == set(range(300))


def test_partition_and_halo_plan():
    from gat_amd import _lib, graph
    n, P = 103, 4
    dst, src = random_graph(n, 4.0, 6)
    rowptr, col, _ = graph.csr_from_coo_host(n, dst, src)
    part = np.empty(P + 1, dtype=np.int64)
    _lib.call("gatmi_partition_rows", n, P, _lib.ptr(rowptr), 0, _lib.ptr(part))
    assert part.tolist() == [(n * p) // P for p in range(P + 1)]
    part_e = np.empty(P + 1, dtype=np.int64)
    _lib.call("gatmi_partition_rows", n, P, _lib.ptr(rowptr), 1, _lib.ptr(part_e))
    assert part_e[0] == 0 and part_e[-1] == n and (np.diff(part_e) >= 0).all()
    for p in range(1, P):
        assert rowptr[part_e[p]] * P >= rowptr[-1] * p and (part_e[p] == 0 or rowptr[part_e[p] - 1] * P < rowptr[-1] * p)
    seen = np.zeros(n, dtype=bool)
    for r in range(P):
        lo, hi = part[r], part[r + 1]
        cg = col[rowptr[lo]:rowptr[hi]].astype(np.int64)
        cl = np.empty(len(cg), dtype=np.int32)
        halo = np.empty(len(cg), dtype=np.int64)
        counts = np.empty(P, dtype=np.int64)
        nh = ctypes.c_int64()
        _lib.call("gatmi_halo_plan", P, r, _lib.ptr(part), len(cg), _lib.ptr(cg), _lib.ptr(cl), _lib.ptr(halo),
                  _lib.ptr(counts), ctypes.byref(nh))
        halo = halo[:nh.value]
        want = np.unique(cg[(cg < lo) | (cg >= hi)])
        assert (halo == want).all()
        owner = np.searchsorted(part, halo, side="right") - 1
        assert (counts == np.bincount(owner, minlength=P)).all() and counts[r] == 0
        table = np.concatenate([np.arange(lo, hi), halo])
        assert (table[cl] == cg).all()
        seen[lo:hi] = True
    assert seen.all()


def test_sparse_features_index_preparation():
    """CSR and transposed CSR of a feature matrix built from the reference's sparse tuple
    (utils/process.py:125-141): same entries as scipy's, transposed entries in node order."""
    import scipy.sparse as sp
    import torch
    from gat_amd import preprocess
    from gat_amd.features import SparseFeatures
    rng = np.random.RandomState(3)
    x = sp.random(60, 35, density=0.1, random_state=rng, format="coo", dtype=np.float32)
    f = SparseFeatures.from_tuple(preprocess.sparse_to_tuple(x), "cpu")
    csr = x.tocsr()
    csr.sort_indices()
    assert (f.rowptr.numpy() == csr.indptr).all() and (f.col.numpy() == csr.indices).all()
    assert (f.val.numpy() == csr.data).all()
    t = x.T.tocsr()
    t.sort_indices()
    assert (f.t_rowptr.numpy() == t.indptr).all() and (f.t_row.numpy() == t.indices).all()
    assert (f.t_val.numpy() == t.data).all()
    assert f.t_rowptr.dtype == torch.int64 and f.t_row.dtype == torch.int32
    assert (f.to_dense().numpy() == x.toarray()).all()


@pytest.mark.parametrize("seed,chunk", [(0, 8), (1, 64), (2, 3)])
def test_phased_work_items_cover_each_part_once_and_pair_their_slots(seed, chunk):
    """gatmi_work_items_phased: rows cut at `cut` into a first part (phase A) and a second part
    (phase B), each chunked; the last field of an item is -1 (finish), a slot (park), or -(slot+2)
    (take up that slot, then finish); rows with more than one piece per phase get a split record."""
    from gat_amd import _lib
    from gat_amd._lib import ptr
    UNSET = -2 ** 31
    rng = np.random.default_rng(seed)
    n = 300
    lens = rng.integers(0, 40, n)
    lens[rng.integers(0, n, 5)] = 0                         # empty rows
    lens[rng.integers(0, n, 3)] = 500                       # rows far longer than the chunk
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(lens)
    frac = rng.random(n)
    frac[rng.integers(0, n, 30)] = 0.0                      # nothing in the first part
    frac[rng.integers(0, n, 30)] = 1.0                      # nothing in the second part
    cut = rowptr[:-1] + np.floor(frac * lens).astype(np.int64)
    na, nb, ns, nl = (ctypes.c_int64() for _ in range(4))
    _lib.call("gatmi_work_items_phased_count", n, ptr(rowptr), ptr(cut), chunk, ctypes.byref(na), ctypes.byref(nb),
              ctypes.byref(ns), ctypes.byref(nl))
    ia, ib = np.full((na.value, 4), UNSET, np.int32), np.full((nb.value, 4), UNSET, np.int32)
    sp_ = np.full((ns.value, 4), UNSET, np.int32)
    _lib.call("gatmi_work_items_phased", n, ptr(rowptr), ptr(cut), chunk, ptr(ia), ptr(ib), ptr(sp_))
    assert (ia != UNSET).all() and (ib != UNSET).all() and (sp_ != UNSET).all()
    for items in (ia, ib):
        assert (np.diff(items[:, 2]) <= 0).all() and (items[:, 2] <= chunk).all()           # longest first
    covered = np.zeros(int(rowptr[-1]), dtype=np.int32)
    for items, phase in ((ia, 0), (ib, 1)):
        for row, first, ln, w in items:
            lo, hi = (rowptr[row], cut[row]) if phase == 0 else (cut[row], rowptr[row + 1])
            assert lo <= first and first + ln <= hi, "piece outside its part"
            covered[first:first + ln] += 1
    assert (covered == 1).all()
    slots_parked, slots_resumed, slot_row = set(), set(), {}
    pieces = {}
    for items in (ia, ib):
        for row, first, ln, w in items:
            pieces.setdefault(int(row), []).append(int(w))
            if w >= 0:
                assert w not in slots_parked
                slots_parked.add(int(w)); slot_row[int(w)] = int(row)
            elif w <= -2:
                slots_resumed.add(int(-w - 2))
    assert set(pieces) == set(range(n))                     # every row has an item, empty rows included
    assert slots_parked == set(range(nl.value))
    split_rows = {int(r[0]): (int(r[1]), int(r[2])) for r in sp_}
    for row, ws in pieces.items():
        if len(ws) == 1:
            assert ws == [-1]
        elif row in split_rows:
            first, cnt = split_rows[row]
            assert sorted(ws) == list(range(first, first + cnt))
        else:
            assert len(ws) == 2 and ws[0] >= 0 and ws[1] == -(ws[0] + 2) and slot_row[ws[0]] == row
    assert all(s in slots_parked for s in slots_resumed)
    bad = cut.copy()
    bad[7] = rowptr[8] + 1
    rc = _lib.lib.gatmi_work_items_phased_count(n, ptr(rowptr), ptr(bad), chunk, ctypes.byref(na), ctypes.byref(nb), ctypes.byref(ns), ctypes.byref(nl))
    assert rc == -1 and "cut" in _lib.last_error()


def test_stepped_seed_is_the_kernels_seed_arithmetic():
    """gat_amd.graphs.stepped_seed restates on the host what a kernel does with a registered step word:
    seed + step * multiplier modulo 2^64, with an odd multiplier (so that steps never repeat a seed)."""
    from gat_amd import _lib, graphs
    mul = int(_lib.lib.gatmi_dropout_step_multiplier())
    assert mul % 2 == 1 and mul > 2 ** 63
    assert graphs.stepped_seed(123, 0) == 123
    assert graphs.stepped_seed(123, 1) == (123 + mul) % 2 ** 64
    assert graphs.stepped_seed(2 ** 64 - 1, 2 ** 64 - 1) == (2 ** 64 - 1 + (2 ** 64 - 1) * mul) % 2 ** 64
    assert len({graphs.stepped_seed(7 + h, s) for h in range(4) for s in range(1000)}) == 4000
