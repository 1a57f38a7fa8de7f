"""The ctypes-only binding sketched for the reference's maintainers (one `sp_attn_head` call through
libgatmi.so with nothing but ctypes; torch only lends device memory and the stream) must work as
written: graph built with gatmi_csr_from_coo / gatmi_work_items, forward through gatmi_proj_fwd and
gatmi_gat_fwd, compared with the float64 oracle."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ref_numpy
from tests.util import assert_close, random_graph

pytestmark = pytest.mark.gpu


def test_ctypes_only_sp_attn_head():
    from gat_amd import _lib                      # only for the path of the shared object
    g = ctypes.CDLL(_lib.LIB_PATH)
    g.gatmi_last_error.restype = ctypes.c_char_p
    g.gatmi_gat_fwd_ws_floats.restype = ctypes.c_size_t

    def ok(rc, what):
        if rc:
            raise RuntimeError("%s: %s" % (what, g.gatmi_last_error().decode()))

    dev = torch.device("cuda:0")
    keep = []                                     # device buffers stay alive until the end

    def upload(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        keep.append(t)
        return ctypes.c_void_p(t.data_ptr())

    def alloc(nfloats):
        t = torch.empty(max(int(nfloats), 1), dtype=torch.float32, device=dev)
        keep.append(t)
        return t

    i64, p = ctypes.c_int64, ctypes.c_void_p
    n, f_in, out_sz, chunk = 300, 10, 8, 256
    dst, src = random_graph(n, 5.0, seed=3, hubs=1, hub_deg=280)
    indices = np.stack([dst, src], 1).astype(np.int32)                      # preprocess_adj_bias layout
    d = np.ascontiguousarray(indices[:, 0]); s = np.ascontiguousarray(indices[:, 1])
    rowptr = np.empty(n + 1, np.int64); col = np.empty(len(d), np.int32); nnz = i64()
    ok(g.gatmi_csr_from_coo(i64(n), i64(len(d)), p(d.ctypes.data), p(s.ctypes.data), 0, 1, i64(len(d)),
                            p(rowptr.ctypes.data), p(col.ctypes.data), None, ctypes.byref(nnz)), "csr_from_coo")
    ni, ns, nl = i64(), i64(), i64()
    ok(g.gatmi_work_items_count(i64(n), p(rowptr.ctypes.data), chunk, ctypes.byref(ni), ctypes.byref(ns), ctypes.byref(nl)), "items_count")
    items = np.empty((ni.value, 4), np.int32); split = np.empty((max(ns.value, 1), 4), np.int32)
    ok(g.gatmi_work_items(i64(n), p(rowptr.ctypes.data), chunk, p(items.ctypes.data), p(split.ctypes.data)), "items")
    assert ns.value >= 1                                                    # the hub row is split

    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, (n, f_in)).astype(np.float32)
    params = {"W": rng.normal(0, 0.4, (f_in, 1, out_sz)), "a1": rng.normal(0, 0.4, (1, out_sz)), "a2": rng.normal(0, 0.4, (1, out_sz)),
              "b1": rng.normal(0, 0.1, 1), "b2": rng.normal(0, 0.1, 1), "bias": rng.normal(0, 0.1, (1, out_sz))}
    dp_ = {k: upload(v.astype(np.float32)) for k, v in params.items()}
    stream = p(torch.cuda.current_stream().cuda_stream)
    H, dp = 1, out_sz
    wh, f1, f2, out = alloc(n * dp), alloc(n), alloc(n), alloc(n * dp)
    ptr = lambda t: p(t.data_ptr())
    ok(g.gatmi_proj_fwd(upload(x), i64(n), f_in, i64(f_in), dp_["W"], dp_["a1"], dp_["a2"], dp_["b1"], dp_["b2"], H, dp,
                        ptr(wh), None, i64(0), ptr(f1), ptr(f2), None, None, None, ctypes.c_float(1.0), ctypes.c_uint64(0), stream), "proj_fwd")
    ws = alloc(g.gatmi_gat_fwd_ws_floats(i64(nl.value), H, dp))
    ok(g.gatmi_gat_fwd(i64(n), H, dp, upload(items), i64(ni.value), upload(split[:ns.value]), i64(ns.value), upload(col[:nnz.value]),
                       ptr(wh), 0, i64(dp), ptr(f1),       # fp32 rows, dp floats apart
                       None, dp_["a2"], dp_["b2"],   # f2 formed from the gathered rows (or: f2, None, None to gather the table)
                       dp_["bias"], None,
                       1,      # GATMI_ACT_ELU
                       0,      # GATMI_REDUCE_CONCAT
                       ctypes.c_float(0.2), ctypes.c_float(1.0), ctypes.c_float(1.0), ctypes.c_uint64(0),
                       ptr(out), None, None, None, None, ptr(ws), stream), "gat_fwd")
    torch.cuda.synchronize()
    rp, rc = ref_numpy.coo_to_csr(indices, n)
    ref = ref_numpy.gat_layer_csr(x.astype(np.float64), params, rp, rc, activation=ref_numpy.elu, reduce="concat")
    assert_close(out.reshape(n, dp).cpu().numpy(), ref, what="ctypes-only sp_attn_head")
