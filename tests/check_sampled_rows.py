#!/usr/bin/env python3
"""Checker for the rows benchmarks/bench_c5_single.py --dump writes: an output row of the layer depends
only on the row's own neighbour list, so the CPU oracle evaluated on the sub-graph induced by the
sampled rows (complete neighbour lists, everything else empty) must reproduce them.  Not collected by
pytest (the graph behind the dump takes minutes to build); run it by hand after the benchmark:

    python tests/check_sampled_rows.py /tmp/c5_rows.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_numpy                     # noqa: E402
from tests.util import assert_close              # noqa: E402


def main(path):
    d = np.load(path)
    layer = {k[2:]: d[k].astype(np.float64) for k in d.files if k.startswith("p_")}
    ref = ref_numpy.gat_layer_csr(d["x"].astype(np.float64), layer, d["rowptr"], d["col"])
    assert_close(d["got"], ref[d["rows"]], what="sampled rows of the forward pass")
    print("sampled rows agree with the oracle: %d rows, %d nodes in the sub-graph, %d entries"
          % (len(d["rows"]), len(d["x"]), len(d["col"])))


if __name__ == "__main__":
    main(sys.argv[1])
