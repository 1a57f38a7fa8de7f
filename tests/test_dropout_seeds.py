"""Independence of the dropout masks across neighbouring seeds (host twins of the in-kernel hashes;
tests/test_dropout_gpu.py checks that the kernels draw exactly these masks).

The layer hands its head groups seed + h0 and callers pass seeds such as the epoch number, so masks of
seeds that differ in a few low bits must not be index permutations of each other: with the seed folded
into the index by XOR alone, mask(seed + 8, i) == mask(seed, i ^ 8), and every edge would lose the same
number of heads at every step."""
import numpy as np
import pytest

from tests.util import feat_keep_host, keep_mask_host, scramble_seed

H = 8
E = 4096


def _coef(seed, keep=0.4):
    idx = np.arange(E, dtype=np.uint64)[:, None] * np.uint64(H) + np.arange(H, dtype=np.uint64)[None, :]
    return keep_mask_host(seed, idx, keep)


@pytest.mark.parametrize("base", [0, 7, 0x1234_5678_9ABC, (1 << 63) + 5])
def test_neighbouring_seeds_are_not_index_permutations(base):
    ref = _coef(base)
    ref_counts = ref.sum(1)
    flat = ref.reshape(-1)
    for d in range(1, 9):
        other = _coef(base + d)
        # (a) not the same mask under any XOR of the flat index with a small constant
        for x in range(1, 64):
            perm = np.arange(E * H) ^ x
            assert not np.array_equal(other.reshape(-1), flat[perm]), "seed + %d is seed with index ^ %d" % (d, x)
        # (b) dropped heads per edge differ between the seeds on a large share of the edges
        assert (other.sum(1) != ref_counts).mean() > 0.5
        # (c) and the two masks agree about as often as independent draws would (keep 0.4: 0.52)
        agree = (other == ref).mean()
        assert 0.48 < agree < 0.56, agree


def test_feature_masks_of_neighbouring_seeds_are_unrelated():
    elem = np.arange(64 * 1024, dtype=np.uint64)
    ref = feat_keep_host(11, elem, 0.4)
    for d in range(1, 9):
        other = feat_keep_host(11 + d, elem, 0.4)
        assert 0.49 < (other == ref).mean() < 0.55
        for x in (4, 8, 16, 32):          # a group is four elements: XOR of the group index
            assert not np.array_equal(other, ref[np.arange(elem.size) ^ x])


def test_keep_rate_and_scrambler_known_values():
    assert abs(_coef(3, 0.4).mean() - 0.4) < 0.01
    assert abs(feat_keep_host(3, np.arange(1 << 16, dtype=np.uint64), 0.4).mean() - 0.4) < 0.01
    # splitmix64 finaliser: the published first outputs of the generator seeded with 0 are
    # mix(0x9E3779B97F4A7C15), mix(2 * 0x9E3779B97F4A7C15)
    assert scramble_seed(0) == 0
    assert scramble_seed(0x9E3779B97F4A7C15) == 0xE220A8397B1DCDAF
    assert scramble_seed((2 * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)) == 0x6E789E6AA1B965F4
