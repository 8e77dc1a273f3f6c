"""MT19937 jump-ahead polynomials (host GF(2) arithmetic) against numpy's own generator.  CPU only."""
import numpy as np


def _raw_sequence(key, n):
    x = [int(v) for v in key]
    while len(x) < n:
        k = len(x) - 624
        y = (x[k] & 0x80000000) | (x[k + 1] & 0x7FFFFFFF)
        x.append(x[k + 397] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0))
    return np.array(x, dtype=np.uint32)


def test_characteristic_polynomial_and_jumps():
    from pinsage_hip import mtjump
    phi = mtjump.characteristic_polynomial()
    assert phi.bit_length() - 1 == 19937 and bin(phi).count("1") == 135      # Matsumoto & Nishimura: 135 terms
    P = mtjump.jump_polynomials()
    assert P.shape == (mtjump.JUMP_LEVELS, 624) and P.dtype == np.uint32
    assert P[0, 0] == 2 and P[0, 1:].sum() == 0                               # t^1
    key = np.random.RandomState(2024).get_state()[1]
    seq = _raw_sequence(key, 624 + (1 << 17) + 1400)
    for level in (3, 12, 15, 16, 17):
        w = mtjump.apply_jump_reference(seq[700:700 + 624], level)
        assert np.array_equal(w, seq[700 + (1 << level):700 + (1 << level) + 624]), level


def test_raw_sequence_is_numpys_stream():
    """The recurrence used by the jump code is numpy's generator: tempered raw words == random_sample bits."""
    rs = np.random.RandomState(99)
    key = rs.get_state()[1]
    seq = _raw_sequence(key, 624 + 2000)[624:]                                # first twisted block onwards
    y = seq.astype(np.uint64)
    y ^= y >> np.uint64(11)
    y ^= (y << np.uint64(7)) & np.uint64(0x9D2C5680)
    y ^= (y << np.uint64(15)) & np.uint64(0xEFC60000)
    y ^= y >> np.uint64(18)
    y &= np.uint64(0xFFFFFFFF)
    a, b = y[0:2000:2] >> np.uint64(5), y[1:2000:2] >> np.uint64(6)
    ref = rs.random_sample(1000)
    assert np.array_equal((a.astype(np.float64) * 67108864.0 + b.astype(np.float64)) / 9007199254740992.0, ref)


def test_radix_polynomials_agree_with_the_binary_table():
    """entry (i, j-1) = t^(j * 2^(c + 5i)): the multipliers 1, 2, 4, 8, 16 of a level are rows of the binary table, and
    every other entry is a product of two smaller ones (checked through the recurrence on one window)."""
    from pinsage_hip import mtjump
    c = 17
    R, B = mtjump.radix_polynomials(c), mtjump.jump_polynomials()
    assert R.shape == (mtjump.RADIX_LEVELS, 31, 624) and R.dtype == np.uint32
    for i in range(mtjump.RADIX_LEVELS):
        for b, j in enumerate((1, 2, 4, 8, 16)):
            if c + 5 * i + b < mtjump.JUMP_LEVELS:
                assert np.array_equal(R[i, j - 1], B[c + 5 * i + b])
    # t^(3u) = t^(u) * t^(2u): applying the jumps u then 2u to a window equals the single jump 3u
    phi = mtjump.characteristic_polynomial()
    as_int = lambda row: int.from_bytes(row.astype("<u4").tobytes(), "little")
    for i, j in ((0, 3), (0, 7), (1, 5), (2, 15), (0, 31), (1, 22)):
        a, b = (j & -j), j - (j & -j)                              # split j into two smaller multipliers
        prod = mtjump._reduce(mtjump._mul(as_int(R[i, a - 1]), as_int(R[i, b - 1])), phi)
        assert prod == as_int(R[i, j - 1])


def test_window_polynomials_are_the_single_jumps():
    """row j-1 = t^(j * 2^c + shift): t^shift times the radix tables' entries for j = 1..31 and 32 j, every row the product of an
    earlier row with an unshifted multiplier, and the sparse reduction agrees with the bit-serial one."""
    from pinsage_hip import mtjump
    c, L = 17, mtjump.WINDOW_SHIFT
    W, R = mtjump.window_polynomials(c), mtjump.radix_polynomials(c)
    assert W.shape == (mtjump.WINDOW_POLYS, 624) and W.dtype == np.uint32
    phi = mtjump.characteristic_polynomial()
    as_int = lambda row: int.from_bytes(row.astype("<u4").tobytes(), "little")
    shifted = lambda row: mtjump._reduce(as_int(row) << L, phi)
    assert all(as_int(W[j - 1]) == shifted(R[0, j - 1]) for j in (1, 2, 7, 31))
    assert all(as_int(W[32 * j - 1]) == shifted(R[1, j - 1]) for j in (1, 5, 15))
    for j in (33, 100, 361, 511):                                  # t^(j u + L) = t^(a u + L) * t^((j - a) u), a u from the radix tables
        a = 32 * (j // 32)
        assert mtjump._reduce(mtjump._mul(as_int(W[j - a - 1]), as_int(R[1, a // 32 - 1])), phi) == as_int(W[j - 1])
    taps = [i for i in range(mtjump.DEG) if (phi >> i) & 1]
    p = mtjump._mul(as_int(W[76]), as_int(W[12]))
    assert mtjump._reduce_sparse(p, phi, taps) == mtjump._reduce(p, phi)
    # unshifted table: the plain single jumps
    W0 = mtjump.window_polynomials(c, count=40, shift=0)
    assert all(np.array_equal(W0[j - 1], R[0, j - 1]) for j in range(1, 32)) and np.array_equal(W0[31], R[1, 0])
