"""The faiss-faithful rotation recipe (reference utils/nearest_neighbors.py:26 -> faiss.IndexLSH(d, nbits,
rotate_data=True) -> RandomRotationMatrix(d, nbits).init(5)).  faiss is absent, so nothing here compares with faiss: the tests
hold the restatement to the PUBLISHED recipe's structure -- std::mt19937 raw words, the per-block Marsaglia polar stream
against an independent scalar restatement, orthonormal rows (nbits <= d) / tight frame (nbits > d).  Parity unpinned."""
import math

import numpy as np
import pytest


def _scalar_float_randn(n, seed):
    """float_randn written as the C loop reads, one value at a time (slow; small n only)"""
    def gen(sd):
        bg = np.random.MT19937()
        bg._legacy_seeding(int(sd) & 0xFFFFFFFF)
        return bg
    g0 = gen(seed)
    a0, b0 = (int(v) & 0x7FFFFFFF for v in g0.random_raw(2))
    nblock = 1 if n < 1024 else 1024
    x = np.empty(n, dtype=np.float32)
    for j in range(nblock):
        rng = gen(a0 + j * b0)
        nxt = lambda: float(rng.random_raw(1)[0]) / 4294967295.0
        a = b = s = 0.0
        state = 0
        for i in range(j * n // nblock, (j + 1) * n // nblock):
            if state == 0:
                while True:
                    a = 2.0 * nxt() - 1
                    b = 2.0 * nxt() - 1
                    s = a * a + b * b
                    if s < 1.0:
                        break
                x[i] = a * math.sqrt(-2.0 * math.log(s) / s)
            else:
                x[i] = b * math.sqrt(-2.0 * math.log(s) / s)
            state = 1 - state
    return x


def test_raw_words_are_the_std_mt19937_stream():
    # std::mt19937(seed) and numpy's legacy seeding both run init_genrand(seed); known answer: mt19937(5489)'s 10000th output
    bg = np.random.MT19937()
    bg._legacy_seeding(5489)
    assert int(bg.random_raw(10000)[-1]) == 4123659995            # the C++ standard's check value for std::mt19937
    bg._legacy_seeding(5)
    rs = np.random.RandomState(5)
    assert np.array_equal(bg.random_raw(64), rs.randint(0, 2 ** 32, 64, dtype=np.uint32))


@pytest.mark.parametrize("n", [7, 1023, 1024, 5000])
def test_float_randn_matches_the_scalar_loop(n):
    from utils.nearest_neighbors import _faiss_float_randn
    got = _faiss_float_randn(n, 5)
    assert np.array_equal(got, _scalar_float_randn(n, 5))
    assert got.dtype == np.float32 and np.isfinite(got).all()


def test_faiss_recipe_structure():
    from utils.nearest_neighbors import lsh_rotation_matrix
    A = lsh_rotation_matrix(64, 32, recipe="faiss")                # nbits <= d: orthonormal rows
    assert A.shape == (32, 64) and A.dtype == np.float32
    np.testing.assert_allclose(A @ A.T, np.eye(32), atol=2e-6)
    B = lsh_rotation_matrix(32, 64, recipe="faiss")                # nbits > d: tight frame, A^T A = I
    assert B.shape == (64, 32)
    np.testing.assert_allclose(B.T @ B, np.eye(32), atol=2e-6)
    assert np.array_equal(B, lsh_rotation_matrix(32, 64, recipe="faiss"))          # deterministic (seed 5)
    assert not np.array_equal(B, lsh_rotation_matrix(32, 64, seed=6, recipe="faiss"))
    # the default recipe is unchanged and still what the oracle restates
    from oracle import pinsage_oracle as orc
    assert np.array_equal(lsh_rotation_matrix(32, 64), orc.lsh_rotation_matrix(32, 64))
    with pytest.raises(ValueError):
        lsh_rotation_matrix(8, 8, recipe="other")
