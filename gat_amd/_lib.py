"""ctypes binding of libgatmi.so (the C ABI declared in include/gatmi.h).

The library is built in-tree by `make -C gat_amd/csrc` (see __graft_entry__.build).  There is no
fallback: if the shared object is missing, importing this module raises, and every wrapper turns
a non-zero return code into RuntimeError carrying gatmi_last_error().
"""
import ctypes
import os

# PyTorch-ROCm ships its own HIP runtime (soname libamdhip64.so.7).  It must be in the process
# before libgatmi.so is opened, so that the kernels are launched through the same runtime instance
# that owns torch's device memory and streams; loading libgatmi first would map a second runtime.
import torch  # noqa: F401  (isort: keep above the CDLL call)

from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgatmi.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "gat_amd: %s not found. Build it with `make -C %s` (needs hipcc, --offload-arch=gfx950); "
        "there is no CPU or PyTorch fallback for the graph-attention kernels." % (LIB_PATH, os.path.join(_HERE, "csrc")))

lib = ctypes.CDLL(LIB_PATH)

P = c_void_p

_SIGNATURES = {
    "gatmi_version": (c_int, []),
    "gatmi_last_error": (c_char_p, []),
    "gatmi_csr_from_coo": (c_int, [c_int64, c_int64, P, P, c_int, c_int, c_int64, P, P, P, P]),
    "gatmi_csc_from_csr": (c_int, [c_int64, c_int64, P, P, P, P, P]),
    "gatmi_work_items_count": (c_int, [c_int64, P, c_int32, P, P, P]),
    "gatmi_work_items": (c_int, [c_int64, P, c_int32, P, P]),
    "gatmi_work_items_phased_count": (c_int, [c_int64, P, P, c_int32, P, P, P, P]),
    "gatmi_work_items_phased": (c_int, [c_int64, P, P, c_int32, P, P, P]),
    "gatmi_partition_rows": (c_int, [c_int64, c_int32, P, c_int32, P]),
    "gatmi_halo_plan": (c_int, [c_int32, c_int32, P, c_int64, P, P, P, P, P]),
    "gatmi_proj_fwd": (c_int, [P, c_int64, c_int32, c_int64, P, P, P, P, P, c_int32, c_int32, P, P, c_int64, P, P, P, P, P, c_float, c_uint64, P]),
    "gatmi_proj_bwd_ws_floats": (c_size_t, [c_int32, c_int32, c_int32]),
    "gatmi_proj_bwd_params": (c_int, [P, c_int64, c_int32, c_int64, P, P, P, P, c_int32, c_int32, P, P, P, P, P, P, c_float, c_uint64, P]),
    "gatmi_proj_bwd_input": (c_int, [P, c_int64, c_int32, P, c_int32, c_int32, P, c_int64, c_int, c_float, c_uint64, P]),
    "gatmi_bias_act_fwd": (c_int, [P, c_int64, c_int32, c_int64, P, c_int32, P, P]),
    "gatmi_bias_act_bwd": (c_int, [P, P, c_int64, c_int32, c_int64, c_int32, P, P, P, P]),
    "gatmi_column_sums_ws_floats": (c_size_t, [c_int64, c_int32]),
    "gatmi_column_sums": (c_int, [P, c_int64, c_int32, c_int64, P, P, P]),
    "gatmi_linear_bwd_ws_floats": (c_size_t, [c_int64, c_int32, c_int32]),
    "gatmi_linear_bwd_params": (c_int, [P, c_int64, c_int32, c_int64, P, c_int32, P, P, P, P]),
    "gatmi_proj_fwd_csr": (c_int, [P, P, P, c_int64, c_int32, P, P, P, P, P, c_int32, c_int32, P, P, P, P]),
    "gatmi_csr_row_normalize": (c_int, [c_int64, P, P, P, P]),
    "gatmi_proj_bwd_params_csr": (c_int, [P, P, P, c_int64, c_int32, P, P, P, P, c_int32, c_int32, P, P, P, P, P, P, P]),
    "gatmi_gat_fwd_ws_floats": (c_size_t, [c_int64, c_int32, c_int32]),
    "gatmi_to_bf16": (c_int, [P, c_int64, P, P]),
    "gatmi_to_bf16_records": (c_int, [P, P, c_int64, c_int32, c_int32, c_int64, P, P]),
    "gatmi_gat_fwd": (c_int, [c_int64, c_int32, c_int32, P, c_int64, P, c_int64, P, P, c_int32, c_int64, P, P, P, P, P, P, c_int32, c_int32,
                              c_float, c_float, c_float, c_uint64, P, P, P, P, P, P, P]),
    "gatmi_gat_fwd_noalloc": (c_int, [c_int64, c_int32, c_int32, P, c_int64, P, c_int64, c_int64, P, P, c_int32, c_int64, P, P, P, P, P, P,
                                      c_int32, c_int32, c_float, c_float, c_float, c_uint64, P, P, P, P, P, P]),
    "gatmi_dropout_step_source": (c_int, [P]),
    "gatmi_dropout_step_source_current": (P, []),
    "gatmi_dropout_step_multiplier": (c_uint64, []),
    "gatmi_gat_bwd_node_ws_floats": (c_size_t, [c_int32, c_int32]),
    "gatmi_gat_bwd_node": (c_int, [c_int64, c_int32, c_int32, P, P, P, P, P, P, P, P, c_int32, c_int32, c_float,
                                   P, c_int64, P, c_int64, P, c_int64, P, P, P, P]),
    "gatmi_gat_bwd_ws_floats": (c_size_t, [c_int64, c_int32, c_int32]),
    "gatmi_gat_bwd_csc": (c_int, [c_int64, c_int32, c_int32, P, c_int64, P, c_int64, P, P, P, c_int32, c_int64, P, c_int64, P, P, P, P, P, P, P,
                                  c_float, c_float, c_float, c_uint64, P, P, P, P]),
    "gatmi_gat_bwd_finalize": (c_int, [c_int64, c_int32, c_int32, P, P, P, P, P, P]),
    "gatmi_gat_bwd_finalize_halo": (c_int, [c_int64, c_int32, c_int32, P, P, P, P, P, P, P, P, P, P]),
    "gatmi_dense_attn_fwd": (c_int, [c_int64, c_int32, c_int32, c_int32, P, P, P, P, P, P, c_int32, c_float,
                                     c_float, c_float, c_uint64, P, P, P, P, P, P]),
    "gatmi_dense_attn_bwd": (c_int, [c_int64, c_int32, c_int32, c_int32, P, P, P, P, P, P, P, P, c_float,
                                     c_float, c_float, c_uint64, P, P, P]),
    "gatmi_dense_mask_words": (c_size_t, [c_int32]),
    "gatmi_dense_attn_fwd_masked": (c_int, [c_int64, c_int32, c_int32, c_int32, P, P, P, P, P, P, c_int32, c_float,
                                            c_float, c_float, c_uint64, P, P, P, P, P, P, P]),
    "gatmi_dense_attn_bwd_masked": (c_int, [c_int64, c_int32, c_int32, c_int32, P, P, P, P, P, P, P, P, P, c_float,
                                            c_float, c_float, c_uint64, P, P, P]),
    "gatmi_dense_attn_bwd_fused_ok": (c_int, [c_int64, c_int32, c_int32, c_int32, c_float, c_float]),
    "gatmi_dense_attn_bwd_fused_ws_floats": (c_size_t, [c_int64, c_int32, c_int32]),
    "gatmi_dense_attn_bwd_fused": (c_int, [c_int64, c_int32, c_int32, c_int32, P, P, P, P, P, P, P, P, c_int32, P, P, P, P,
                                           P, P, P, c_int64, P, P, c_float, c_float, c_float, c_uint64, P, P, P, P, P, P, P, P, P, P, P]),
    "gatmi_dense_attn_bwd_fused_param_grads": (c_int, [c_int32, c_int32, c_int32, c_int32]),
    "gatmi_dense_mask_bits": (c_int, [c_int64, c_int32, P, c_float, P, P]),
    "gatmi_dense_lists_count": (c_int, [c_int64, c_int32, P, P, P, P]),
    "gatmi_dense_lists_fill": (c_int, [c_int64, c_int32, P, P, P, P, P, P]),
    "gatmi_dense_lists_capacity": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "gatmi_dense_attn_fwd_lists": (c_int, [c_int64, c_int32, c_int32, c_int32, P, P, P, P, P, P, c_int64, P, P, c_int32, c_float,
                                           c_float, c_float, c_uint64, P, P, P, P, P, P]),
    "gatmi_masked_ce_ws_floats": (c_size_t, []),
    "gatmi_masked_ce": (c_int, [c_int64, c_int32, P, c_int64, P, c_int64, P, P, P, P, P]),
    "gatmi_scale_by_device_scalar": (c_int, [P, c_int64, P, P]),
    "gatmi_bnf_attn_fwd_ws_floats": (c_size_t, [c_int64, c_int32, c_int32, c_int32]),
    "gatmi_bnf_attn_fwd": (c_int, [c_int64, c_int32, c_int32, c_int32, P, P, P, P, P, P, c_float, P, P, P, P, P, c_int32, P]),
    "gatmi_bnf_attn_bwd_ws_floats": (c_size_t, [c_int64, c_int32, c_int32, c_int32, c_int32]),
    "gatmi_bnf_attn_bwd": (c_int, [c_int64, c_int32, c_int32, c_int32, P, P, P, P, P, P, P, P, P, P, c_float, P, P, c_int, P, P, P, P, c_int32, P]),
    "gatmi_ipd_fwd": (c_int, [c_int64, c_int32, c_int32, P, c_int32, c_float, c_uint64, P, P]),
    "gatmi_ipd_bwd": (c_int, [c_int64, c_int32, c_int32, P, P, P, c_int32, c_float, c_uint64, P, P]),
    "gatmi_halo_pack": (c_int, [P, P, c_int64, c_int32, P, P]),
    "gatmi_halo_unpack_add": (c_int, [P, P, c_int64, c_int32, P, P]),
    "gatmi_halo_unpack_sum": (c_int, [P, P, P, P, c_int64, c_int32, P, P]),
}

EXPORTS = tuple(_SIGNATURES)

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = the .so does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args

ACT_IDENTITY, ACT_ELU, ACT_SIGMOID, ACT_RELU = 0, 1, 2, 3
DTYPE_F32, DTYPE_BF16 = 0, 1
REDUCE_CONCAT, REDUCE_MEAN = 0, 1


class GatmiError(RuntimeError):
    pass


def last_error():
    return lib.gatmi_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise GatmiError("%s failed (code %d): %s" % (what, rc, last_error()))


def call(name, *args):
    """Call an int-returning entry point and raise on failure."""
    check(getattr(lib, name)(*args), name)


def ptr(t):
    """Device or host pointer of a torch tensor / numpy array, None -> NULL."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream():
    """Handle of the HIP stream torch currently records on (the hipStream_t the kernels are enqueued
    on).  The raw accessor avoids building a torch.cuda.Stream object per call (about 10 us each, a
    visible share of a launch-bound step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
