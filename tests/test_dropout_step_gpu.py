"""Dropout masks that change under HIP-graph replay (gatmi_dropout_step_source).

A launch with dropout on carries the address of a step word when one is registered; its kernels use
seed + word * multiplier.  Checked here: (1) for every kernel family, a run that reads step k from the
word is bit-identical — forward and every gradient — to a run without a word whose seed was stepped
on the host, so the host-replayed masks of tests/test_dropout_gpu.py cover these runs too; (2) a
captured training step draws new masks on every replay, and each replay equals the eager step at
that replay's step count; (3) with no word registered nothing changes."""
import numpy as np
import pytest
import torch

from tests.util import layer_params, random_graph

pytestmark = pytest.mark.gpu

SEED = 0x1234_5678_9ABC


def _leaves(p, dev):
    return {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}


def _sparse_runner(storage):
    from gat_amd import CSRGraph, ops
    dev = torch.device("cuda:0")
    n, F, H, D = 400, 24, 8, 8
    dst, src = random_graph(n, 7.0, seed=11, hubs=2, hub_deg=250)
    graph = CSRGraph.from_coo(dst, src, n, dev, chunk=64)
    tp = _leaves(layer_params(F, H, D, seed=5), dev)
    x = torch.randn(n, F, device=dev, generator=torch.Generator(device=dev).manual_seed(1), requires_grad=True)
    dout = torch.randn(n, H * D, device=dev, generator=torch.Generator(device=dev).manual_seed(2))

    def run(seed):
        cfg = ops.make_cfg(H, D, "elu", "concat", coef_drop=0.5, in_drop=0.4, training=True, seed=seed, storage=storage)
        for t in (x, *tp.values()):
            t.grad = None
        out = ops.sparse_gat_layer(x, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph, cfg)
        out.backward(dout)
        return [out.detach().clone(), x.grad.clone()] + [tp[k].grad.clone() for k in ("W", "a1", "a2", "bias")]
    return run


def _dense_runner(path):
    from gat_amd import ops
    dev = torch.device("cuda:0")
    B, N, F, H, D = 3, 72, 12, 8, 8
    rng = np.random.default_rng(7)
    adj = (rng.random((B, N, N)) < 0.15) | np.eye(N, dtype=bool)[None]
    bm = torch.tensor(np.where(adj, 0.0, -1e9), dtype=torch.float32, device=dev)
    tp = _leaves(layer_params(F, H, D, seed=6), dev)
    x = torch.tensor(rng.normal(0, 1, (B, N, F)), dtype=torch.float32, device=dev, requires_grad=True)
    dout = torch.tensor(rng.normal(0, 1, (B, N, H * D)), dtype=torch.float32, device=dev)

    def run(seed):
        cfg = ops.make_cfg(H, D, "elu", "concat", coef_drop=0.5, in_drop=0.4, training=True, seed=seed, mask_path=path)
        for t in (x, *tp.values()):
            t.grad = None
        out = ops.dense_gat_layer(x, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], bm, cfg)
        out.backward(dout)
        return [out.detach().clone(), x.grad.clone()] + [tp[k].grad.clone() for k in ("W", "a1", "a2", "bias")]
    return run


def _decoder_runner():
    from gat_amd import ops
    dev = torch.device("cuda:0")
    z = torch.randn(5, 40, 16, device=dev, generator=torch.Generator(device=dev).manual_seed(3), requires_grad=True)
    dflat = torch.randn(5 * 40 * 40, device=dev, generator=torch.Generator(device=dev).manual_seed(4))

    def run(seed):
        z.grad = None
        out = ops.inner_product_decoder(z, "sigmoid", dropout=0.3, seed=seed)
        out.backward(dflat.reshape(out.shape))
        return [out.detach().clone(), z.grad.clone()]
    return run


RUNNERS = {
    "csr": lambda: _sparse_runner("f32"),
    "csr_bf16_tables": lambda: _sparse_runner("bf16"),
    "dense_matrix_core": lambda: _dense_runner("dense"),
    "dense_lists": lambda: _dense_runner("lists"),
    "decoder": _decoder_runner,
}


@pytest.mark.parametrize("family", list(RUNNERS))
def test_step_word_is_the_seed_stepped_on_the_host(family):
    from gat_amd import graphs
    run = RUNNERS[family]()
    plain = run(SEED)
    word = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    with graphs.dropout_step_word(word):
        at0 = run(SEED)                                   # word = 0: the seed as passed
        word.fill_(3)
        at3 = run(SEED)
        word.fill_(-1)                                    # 2^64 - 1 as the kernels read it: wraps like the host twin
        wrapped = run(SEED)
    after = run(SEED)                                     # registration ended: the word is ignored again
    by_seed3 = run(graphs.stepped_seed(SEED, 3))
    by_seed_wrapped = run(graphs.stepped_seed(SEED, (1 << 64) - 1))
    torch.cuda.synchronize()
    for a, b, c in zip(plain, at0, after):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert not torch.equal(plain[0], at3[0]), "step 3 drew the masks of step 0"
    for a, b in zip(at3, by_seed3):
        assert torch.equal(a, b), "%s: a run at step 3 differs from the run with the stepped seed" % family
    for a, b in zip(wrapped, by_seed_wrapped):
        assert torch.equal(a, b)


def test_captured_step_draws_new_masks_on_every_replay():
    from gat_amd import graphs
    run = _sparse_runner("f32")
    kept = {}

    def step():
        kept["r"] = run(SEED)

    replay, _ = graphs.capture(step, warmup=2)
    seen = []
    for _ in range(3):
        replay()
        torch.cuda.synchronize()
        seen.append((int(replay.step_word.item()), [t.clone() for t in kept["r"]]))
    assert [k for k, _ in seen] == [3, 4, 5]              # two warm-ups (capturing runs nothing), then one per replay
    assert not torch.equal(seen[0][1][0], seen[1][1][0]) and not torch.equal(seen[1][1][0], seen[2][1][0])
    for k, got in seen:
        want = run(graphs.stepped_seed(SEED, k))          # eager, no word registered any more
        for a, b in zip(got, want):
            assert torch.equal(a, b), "replay at step %d differs from the eager step with that step's seed" % k


def test_capture_without_fresh_dropout_replays_one_set_of_masks():
    from gat_amd import graphs
    run = _sparse_runner("f32")
    kept = {}

    def step():
        kept["r"] = run(SEED)

    replay, _ = graphs.capture(step, warmup=1, fresh_dropout=False)
    assert replay.step_word is None
    replay()
    torch.cuda.synchronize()
    first = [t.clone() for t in kept["r"]]
    replay()
    torch.cuda.synchronize()
    for a, b, c in zip(first, kept["r"], run(SEED)):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_nested_scopes_put_back_the_outer_word():
    from gat_amd import _lib, graphs
    outer = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    inner = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    cur = _lib.lib.gatmi_dropout_step_source_current
    assert cur() is None
    with graphs.dropout_step_word(outer):
        assert cur() == outer.data_ptr()
        with graphs.dropout_step_word(inner):
            assert cur() == inner.data_ptr()
        assert cur() == outer.data_ptr(), "leaving the inner scope dropped the outer scope's word"
        run = _sparse_runner("f32")
        outer.fill_(2)
        at2 = run(SEED)
    assert cur() is None
    by_seed2 = run(graphs.stepped_seed(SEED, 2))
    torch.cuda.synchronize()
    for a, b in zip(at2, by_seed2):
        assert torch.equal(a, b)


def test_step_source_rejects_a_misaligned_word():
    from gat_amd import _lib
    word = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    assert _lib.lib.gatmi_dropout_step_source(word.data_ptr() + 4) == -2          # GATMI_EALIGN
    assert "aligned" in _lib.last_error()
    assert _lib.lib.gatmi_dropout_step_source(None) == 0
