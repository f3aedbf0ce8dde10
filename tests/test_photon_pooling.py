"""Batching / partition logic of the photon-pooling image builder -- the expectations of the
reference's tests/test_photon_pooling.py (:129-267), run against imsim_amd.photon_pooling."""
from collections import Counter
from dataclasses import replace
from random import shuffle

import numpy as np
import pytest

from imsim_amd import photon_pooling as pp
from imsim_amd.stamp import ObjectInfo, ProcessingMode


def fft_list(n, flux=10 ** 6, start=0):
    return [ObjectInfo(i + start, flux, ProcessingMode.FFT) for i in range(n)]


def phot_list(n, flux=10 ** 5, start=0):
    return [ObjectInfo(i + start, flux, ProcessingMode.PHOT) for i in range(n)]


def faint_list(n, flux=100, start=0):
    return [ObjectInfo(i + start, flux, ProcessingMode.FAINT) for i in range(n)]


@pytest.mark.parametrize("maker,mode", [(fft_list, ProcessingMode.FFT), (phot_list, ProcessingMode.PHOT),
                                        (faint_list, ProcessingMode.FAINT)])
def test_partition_all_same_type(maker, mode):
    objs = maker(20)
    parts = dict(zip((ProcessingMode.FFT, ProcessingMode.PHOT, ProcessingMode.FAINT), pp.partition_objects(objs, 10)))
    for m, lst in parts.items():
        assert len(lst) == (20 if m == mode else 0)
    assert Counter(o.index for o in parts[mode]) == Counter(range(20))


def test_partition_mixed():
    base = fft_list(10) + phot_list(9) + faint_list(1)
    shuffle(base)
    base = [replace(o, index=i) for i, o in enumerate(base)]
    fft, phot, faint = pp.partition_objects(base, 10)
    assert (len(fft), len(phot), len(faint)) == (10, 9, 1)
    assert Counter(o.index for o in fft + phot + faint) == Counter(range(20))


def test_partition_low_flux_phot_is_demoted():
    objs = phot_list(10) + faint_list(5, start=10) + phot_list(5, flux=50, start=15)
    fft, phot, faint = pp.partition_objects(objs, 100)
    assert (len(fft), len(phot), len(faint)) == (0, 10, 10)
    assert all(o.phot_flux >= 100 for o in phot)


def test_make_batches():
    objs = fft_list(20)
    for b, batch in enumerate(pp.make_batches(objs, 10)):
        assert [o.index for o in batch] == [2 * b, 2 * b + 1]
    assert [len(b) for b in pp.make_batches(objs, 6)] == [4, 4, 3, 3, 3, 3]
    assert [len(b) for b in pp.make_batches(objs, 9)] == [3, 3, 2, 2, 2, 2, 2, 2, 2]
    assert [o.index for b in pp.make_batches(objs, 9) for o in b] == list(range(20))


def test_make_photon_batches_conserves_flux():
    phot, faint = phot_list(15), faint_list(5, start=15)
    nbatch = 11
    batches = pp.make_photon_batches(phot, faint, nbatch)
    count = Counter(o.index for b in batches for o in b)
    total = np.zeros(20)
    for b in batches:
        for o in b:
            total[o.index] += o.phot_flux
    assert all(count[o.index] == nbatch for o in phot)
    assert all(count[o.index] == 1 for o in faint)
    np.testing.assert_array_equal(total, [o.phot_flux for o in phot + faint])


def test_make_photon_subbatches():
    batch = phot_list(90) + faint_list(10, start=90)
    for nsub, expect in ((10, 10 * [10]), (8, 4 * [13] + 4 * [12]), (3, [34, 33, 33])):
        subs = pp.make_photon_subbatches(batch, nsub)
        assert [len(s) for s in subs] == expect
        assert [o for s in subs for o in s] == batch


def test_pooling_requires_lsst_photons_stamp():
    with pytest.raises(pp.GalSimConfigValueError):
        pp.check_stamp_type("LSST_Silicon")
    pp.check_stamp_type("LSST_Photons")
