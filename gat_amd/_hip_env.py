"""One switch of the HIP runtime that replay of a captured step depends on.

On ROCm 7.2 with PyTorch 2.10+rocm7.0, a HIP graph instantiated with the runtime's default "packet capture" (the AQL
packets and kernel arguments of the graph's nodes are built once, at instantiation) goes stale once about six thousand
ordinary kernel launches have been issued by the process between the capture and a replay: from then on some of the
graph's kernels run with arguments of unrelated eager launches, outputs keep the values an eager call left, and the
replay reports no error.  It takes no code of this package to see it (benchmarks/micro/graph_replay_min.py shows it with
plain torch operators; a replay stream of its own does not help).  With DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 the runtime
builds the packets on every replay and the results stay right (the same reproducer, 20 000 launches in between); a
replayed training step of a 50 x 90-node autoencoder cost 4.4 ms that way against 4.1 ms, and 12.3 ms eager,
when the switch was introduced.

The runtime reads the switch once, at the process's first HIP call, so it is set here, when the package is imported,
unless the environment already holds a value (set it to 1 to keep the runtime's default).  A process that has already
used the GPU before importing the package is too late for it: gat_amd.graphs.capture warns in that case.
"""
import os

import torch

SWITCH = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"

_before = os.environ.get(SWITCH)
_hip_was_up = torch.cuda.is_initialized()
if _before is None and not _hip_was_up:
    os.environ[SWITCH] = "0"


def replay_is_safe():
    """True when graphs of this process are rebuilt on every replay: the switch was 0 in the environment the process
    started with, or this module set it before torch had initialised HIP."""
    if _before is not None:
        return _before == "0"
    return not _hip_was_up
