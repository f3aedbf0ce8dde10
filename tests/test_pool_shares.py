"""Renderer._pool_shares (the overlapped pool shoot of photon-pooling mode, IMS_POOL_OVERLAP=1): which objects are shot by share and which
whole, from the batches' own descriptions -- host logic, no GPU."""
import numpy as np

from imsim_amd import photon_pooling, stamp
from imsim_amd.engine import Renderer


def _batches(n_phot, nbatch, seed=5):
    modes = stamp.classify(n_phot.astype(float), 100.0)
    shares, _ = photon_pooling.batch_shares(n_phot, modes, nbatch, seed)
    return [(np.asarray(index), np.asarray(first), np.asarray(count), 0) for index, first, count in shares]


def test_objects_shot_by_share_are_covered_exactly_once_and_the_rest_whole():
    rng = np.random.default_rng(3)
    n_phot = np.concatenate([rng.integers(1, 90, 400), rng.integers(700, 50_000, 60), [0, 0, 5_000_000]]).astype(np.int64)
    nb = 10
    batches = _batches(n_phot, nb)
    rows_whole, shares = Renderer._pool_shares(None, batches, n_phot, 64)
    big = (n_phot // nb) > 64
    assert set(rows_whole) == set(np.flatnonzero(~big & (n_phot > 0)))           # photon-less rows are shot by nobody
    covered = np.zeros(len(n_phot), dtype=np.int64)
    seen = [set() for _ in n_phot]
    for rows, first, count in shares:
        assert np.all(big[rows]) and np.all(count > 64) and np.all(np.diff(rows) > 0)      # full wavefronts, shoot-table order
        for r, f, c in zip(rows, first, count):
            rng_ = set(range(int(f), int(f + c))) if c < 2000 else None
            if rng_ is not None:
                assert not (seen[r] & rng_)
                seen[r] |= rng_
        np.add.at(covered, rows, count)
    assert np.array_equal(covered[big], n_phot[big]) and not covered[~big].any()


def test_parts_form_gives_the_same_shares_as_the_host_form():
    rng = np.random.default_rng(4)
    n_phot = rng.integers(700, 30_000, 50).astype(np.int64)
    nb = 5
    rows = np.arange(len(n_phot), dtype=np.int64)
    parts = [("parts", [(rows, ("share", n_phot, i, nb), None, False)], 0) for i in range(nb)]
    host = [(rows, (n_phot * i) // nb, (n_phot * (i + 1)) // nb - (n_phot * i) // nb, 0) for i in range(nb)]
    a = Renderer._pool_shares(None, parts, n_phot, 64)
    b = Renderer._pool_shares(None, host, n_phot, 64)
    assert np.array_equal(a[0], b[0])
    for (ra, fa, ca), (rb, fb, cb) in zip(a[1], b[1]):
        assert np.array_equal(ra, rb) and np.array_equal(fa, fb) and np.array_equal(ca, cb)


def test_batches_that_do_not_cover_an_object_turn_the_overlap_off():
    n_phot = np.array([10_000, 20_000], dtype=np.int64)
    rows = np.arange(2, dtype=np.int64)
    nb = 4
    good = [(rows, (n_phot * i) // nb, (n_phot * (i + 1)) // nb - (n_phot * i) // nb, 0) for i in range(nb)]
    assert Renderer._pool_shares(None, good, n_phot, 64) is not None
    assert Renderer._pool_shares(None, good[:-1], n_phot, 64) is None            # a batch is missing: nothing overlaps, one launch shoots all
    assert Renderer._pool_shares(None, good, np.array([10, 20], dtype=np.int64), 64) is None   # nobody fills wavefronts per batch
