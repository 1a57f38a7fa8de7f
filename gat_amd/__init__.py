"""gat_amd — an MI355X (gfx950) native graph-attention layer.

Drop-in for the attention heads of the TensorFlow-1 reference (utils/layers.py `attn_head`,
`sp_attn_head`; models `GAT` / `SpGAT` `.inference()`), computed by hand-written HIP kernels in
libgatmi.so behind a plain C ABI (include/gatmi.h).
"""
from . import _hip_env  # noqa: F401  (first: it has to run before the process's first HIP call)
from . import _lib  # noqa: F401  (raises ImportError when libgatmi.so has not been built)
from .graph import CSRGraph  # noqa: F401

__version__ = "0.1.0"
