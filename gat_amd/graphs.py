"""Replay of a whole training step from a HIP graph.

libgatmi only enqueues kernels on the caller's stream (no allocation, no synchronisation, no host
read-back), so a step built from its operators captures as it is.  At Cora's size a two-layer
forward + backward is about 25 launches of a few microseconds each; eager execution is bound by
the host issuing them, the replay is not.
"""
import torch


def capture(step, warmup=3):
    """Run `step()` `warmup` times on a side stream (allocator warm-up, lazy initialisations, the
    one-off mask statistics of the dense layer), capture one more call into a HIP graph and return
    (replay, result): `replay()` re-runs the captured kernels on the tensors the capture used — update
    inputs in place (`x.copy_(...)`) before replaying and read outputs from the same tensors after —
    and `result` is whatever the captured call returned.

    Dropout: the kernels receive the dropout seed BY VALUE (drawn on the host when the step is
    issued), so a captured step replays the SAME coefficient and projected-feature masks
    (utils/layers.py:20,22) on every replay; only torch's own dropout of the layer input (:9)
    advances its generator under replay.  Capture steps whose dropout rates are 0 (evaluation, the
    benchmarks) or accept fixed masks; for fresh masks every step run eagerly."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        result = step()
    return graph.replay, result
