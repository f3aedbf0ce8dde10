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
