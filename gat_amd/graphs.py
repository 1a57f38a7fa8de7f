"""Replay of a whole training step from a HIP graph.

libgatmi only enqueues kernels on the caller's stream (no allocation, no synchronisation, no host
read-back), so a step built from its operators captures as it is.  At Cora's size a two-layer
forward + backward is about 25 launches of a few microseconds each; eager execution is bound by
the host issuing them, the replay is not.
"""
import contextlib
import warnings

import torch

from . import _hip_env, _lib

_MASK64 = (1 << 64) - 1


def stepped_seed(seed, step):
    """The seed a kernel uses when it reads `step` from a registered step word
    (gatmi_dropout_step_source): seed + step * multiplier, mod 2^64."""
    return (int(seed) + int(step) * int(_lib.lib.gatmi_dropout_step_multiplier())) & _MASK64


@contextlib.contextmanager
def dropout_step_word(word):
    """While the block runs, every dropout launch issued on `word`'s device — from any thread or
    stream: the registration is device-wide — carries the address of `word` (a one-element int64 tensor
    on that device) and reads it when the kernel runs.  On exit the word that was registered before
    (an enclosing scope's, or none) is put back."""
    if word.dtype != torch.int64 or word.numel() != 1 or not word.is_cuda:
        raise ValueError("the step word is a one-element int64 tensor on the GPU")
    with torch.cuda.device(word.device):
        before = _lib.lib.gatmi_dropout_step_source_current()
        _lib.call("gatmi_dropout_step_source", word.data_ptr())
        try:
            yield word
        finally:
            with torch.cuda.device(word.device):
                _lib.call("gatmi_dropout_step_source", before)


def capture(step, warmup=3, fresh_dropout=True):
    """Run `step()` `warmup` times on a side stream (allocator warm-up, lazy initialisations, the
    one-off mask statistics of the dense layer), capture one more call into a HIP graph and return
    (replay, result): `replay()` re-runs the captured kernels on the tensors the capture used — update
    inputs in place (`x.copy_(...)`) before replaying and read outputs from the same tensors after —
    and `result` is whatever the captured call returned.

    Dropout: the kernels receive the dropout seed BY VALUE (drawn on the host when the step is
    issued), and a graph freezes kernel arguments.  With `fresh_dropout` (default) the captured
    launches therefore also carry the address of a step word on the device, the captured region starts
    by adding one to it, and the kernels mix it into their seed when they run: every replay draws new
    coefficient and projected-feature masks (utils/layers.py:20,22), the forward and backward launches
    of one replay agree, and `replay.step_word` holds the count (gat_amd.graphs.stepped_seed gives the
    seed a replay used).  torch's own dropout of the layer input (:9) advances its generator under
    replay by itself.  `fresh_dropout=False` replays the masks of the captured call unchanged.

    A graph whose packets the HIP runtime built once goes stale after several thousand eager launches
    between capture and replay (gat_amd/_hip_env.py); importing this package before the process first
    uses the GPU turns that mode off, and a capture in a process where it could not warns."""
    if not _hip_env.replay_is_safe():
        warnings.warn("%s was not 0 when this process first used the GPU: on this HIP runtime the replay of a graph "
                      "can silently run with stale kernel arguments once a few thousand eager launches have been "
                      "issued since its capture; export %s=0, or import gat_amd before the first GPU call"
                      % (_hip_env.SWITCH, _hip_env.SWITCH), RuntimeWarning, stacklevel=2)
    device = torch.device("cuda", torch.cuda.current_device())
    word = torch.zeros(1, dtype=torch.int64, device=device) if fresh_dropout else None
    scope = dropout_step_word(word) if fresh_dropout else contextlib.nullcontext()

    def one():
        if word is not None:
            word.add_(1)
        return step()

    with scope:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                one()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            result = one()

    def replay():
        graph.replay()

    replay.graph = graph
    replay.step_word = word          # kept alive with the graph: the captured kernels read it
    return replay, result
