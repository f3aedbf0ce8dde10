"""Pin the oracle's numerics spec: Philox4x32-10 against the Random123 known-answer vectors and
the elementary functions against libm."""
import ctypes as C

import numpy as np

from oracle import orc_loader


def _philox(ctr, key):
    lib = orc_loader.load()
    c = (C.c_uint32 * 4)(*ctr)
    lib.orc_test_philox(c, C.c_uint32(key[0]), C.c_uint32(key[1]))
    return [int(v) for v in c]


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 with 10 rounds
    assert _philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_elementary_functions_accuracy():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 1, 100000), 10 ** rng.uniform(-16, 3, 100000)])
    x = x[x > 0]
    np.testing.assert_allclose(orc_loader.math_probe(0, x), np.log(x), rtol=1e-15, atol=1e-14)
    x = rng.uniform(-50, 50, 100000)
    np.testing.assert_allclose(orc_loader.math_probe(1, x), np.exp(x), rtol=1e-15)
    u = rng.uniform(0, 1, 100000)
    sc = orc_loader.math_probe(2, u).reshape(-1, 2)
    np.testing.assert_allclose(sc[:, 0], np.sin(2 * np.pi * u), atol=2e-15)
    np.testing.assert_allclose(sc[:, 1], np.cos(2 * np.pi * u), atol=2e-15)
    x = np.concatenate([rng.uniform(-5, 5, 100000), 10 ** rng.uniform(-12, 6, 50000), [0.0, np.inf, -np.inf]])
    np.testing.assert_allclose(orc_loader.math_probe(3, x), np.arctan(x), rtol=2e-15, atol=1e-300)
    x = rng.uniform(-10, 10, 100000)
    sc = orc_loader.math_probe(4, x).reshape(-1, 2)
    np.testing.assert_allclose(sc[:, 0], np.sin(x), atol=5e-15)
    x = rng.uniform(0, 30, 50000)
    np.testing.assert_allclose(orc_loader.math_probe(5, x), np.tanh(x), atol=1e-15)


def test_gaussian_deviates_are_standard_normal():
    g = orc_loader.gauss_probe(seed=7, obj=3, n=500000, slot=2)
    assert abs(g.mean()) < 5e-3
    assert abs(g.std() - 1.0) < 5e-3
    assert abs((g ** 4).mean() - 3.0) < 0.05
    # the two members of a pair are uncorrelated
    assert abs(np.mean(g[0::2] * g[1::2])) < 5e-3


def test_word_deviate_functions_accuracy():
    """Spec v6: functions of 32-bit words, their series cut where the truncation error is below 2^-36 of the result --
    sin / cos of 2 pi (w + 1/2) / 2^32 within 1.5e-11, log((w + 1/2) / 2^32) within 2.5e-12 relative; the quadrant
    reduction on the integer word is exact (the four quarter turns and their neighbours are among the test words)."""
    rng = np.random.default_rng(3)
    edges = np.array([0, 1, 2 ** 29 - 1, 2 ** 29, 2 ** 29 + 1, 2 ** 30 - 1, 2 ** 30, 2 ** 30 + 1, 2 ** 31 - 1, 2 ** 31, 3 * 2 ** 29 - 1,
                      3 * 2 ** 29, 3 * 2 ** 30, 5 * 2 ** 29, 7 * 2 ** 29 - 1, 7 * 2 ** 29, 2 ** 32 - 2, 2 ** 32 - 1], dtype=np.float64)
    w = np.concatenate([edges, rng.integers(0, 2 ** 32, 1000000).astype(np.float64)])
    u = (w + 0.5) / 2.0 ** 32
    sc = orc_loader.math_probe(10, w).reshape(-1, 2)
    # reference angle reduced exactly before the libm call (2 pi u itself loses 1e-16 x 2 pi of phase: irrelevant here)
    assert np.abs(sc[:, 0] - np.sin(2.0 * np.pi * u)).max() < 1.5e-11
    assert np.abs(sc[:, 1] - np.cos(2.0 * np.pi * u)).max() < 1.5e-11
    assert np.abs(sc[:, 0] ** 2 + sc[:, 1] ** 2 - 1.0).max() < 3e-11
    full = orc_loader.math_probe(2, u).reshape(-1, 2)                 # the full-length series on the same arguments
    assert np.abs(sc - full).max() < 1.5e-11
    lg = orc_loader.math_probe(11, w)
    assert np.abs(lg / np.log(u) - 1.0).max() < 2.5e-12
    # the single deviate is the cosine half of the pair made from the same two words
    pairs = rng.integers(0, 2 ** 32, 200000).astype(np.float64)
    one = orc_loader.math_probe(12, pairs)
    w0, w1 = pairs[0::2], pairs[1::2]
    r = np.sqrt(-2.0 * np.log((w0 + 0.5) / 2.0 ** 32))
    assert np.abs(one - r * np.cos(2.0 * np.pi * (w1 + 0.5) / 2.0 ** 32)).max() < 2e-10
    assert abs(one.mean()) < 0.01 and abs(one.std() - 1.0) < 0.01
