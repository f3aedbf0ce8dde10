"""imsim_amd/tuning.py: the one module that reads (and, for rocFFT's kernel file, sets) the process environment.  No GPU."""
import os

import pytest

from imsim_amd import tuning


def test_unknown_switch_is_an_error():
    with pytest.raises(KeyError):
        tuning.env("IMS_NO_SUCH_SWITCH")
    with pytest.raises(KeyError):
        tuning.setdefault("IMS_NO_SUCH_SWITCH", 1)


def test_flag_and_number_follow_the_environment(monkeypatch):
    monkeypatch.delenv("IMS_FOCAL_JOINT", raising=False)
    assert tuning.number("IMS_FOCAL_JOINT") == int(tuning.KNOWN["IMS_FOCAL_JOINT"][0]) > 1 and tuning.flag("IMS_FOCAL_JOINT")
    monkeypatch.setenv("IMS_FOCAL_JOINT", "0")
    assert tuning.number("IMS_FOCAL_JOINT") == 0 and not tuning.flag("IMS_FOCAL_JOINT")
    monkeypatch.setenv("IMS_FOCAL_TOUCH", "")                    # set but empty: no touch, not the default order
    assert tuning.env("IMS_FOCAL_TOUCH") == ""
    monkeypatch.delenv("IMS_FOCAL_TOUCH")
    assert sorted(tuning.env("IMS_FOCAL_TOUCH").split(",")) == ["bulk", "mid", "pre", "top0"]


def test_fft_kernel_file_is_seeded_once_and_an_explicit_path_wins(tmp_path, monkeypatch):
    """fft_kernel_cache: unset -> <cache dir>/imsim_amd/rocfft_kernels.db, a copy of the seed when new; a file that is already
    there is kept (it holds what this machine compiled since); ROCFFT_RTC_CACHE_PATH given -> untouched."""
    seed = tmp_path / "seed.db"
    seed.write_bytes(b"kernels of the seed")
    monkeypatch.delenv("ROCFFT_RTC_CACHE_PATH", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))
    path = tuning.fft_kernel_cache(str(seed))
    assert path == str(tmp_path / "cache" / "imsim_amd" / "rocfft_kernels.db")
    assert os.environ["ROCFFT_RTC_CACHE_PATH"] == path
    assert open(path, "rb").read() == b"kernels of the seed"
    # the next process: its own file is kept
    with open(path, "wb") as f:
        f.write(b"grown")
    monkeypatch.delenv("ROCFFT_RTC_CACHE_PATH")
    assert tuning.fft_kernel_cache(str(seed)) == path and open(path, "rb").read() == b"grown"
    # an explicit choice is left alone
    monkeypatch.setenv("ROCFFT_RTC_CACHE_PATH", str(tmp_path / "mine.db"))
    assert tuning.fft_kernel_cache(str(seed)) == str(tmp_path / "mine.db")
    assert not (tmp_path / "mine.db").exists()


def test_fft_kernel_file_falls_back_to_the_temporary_directory(tmp_path, monkeypatch):
    """a cache directory that cannot be made (a batch user without a home): the temporary directory takes the file"""
    blocker = tmp_path / "file_not_dir"
    blocker.write_bytes(b"")
    monkeypatch.delenv("ROCFFT_RTC_CACHE_PATH", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(blocker))
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    monkeypatch.setattr(tempfile, "tempdir", None)               # re-read TMPDIR
    path = tuning.fft_kernel_cache(None)
    assert path is not None and path.startswith(str(tmp_path)) and path.endswith("rocfft_kernels.db")
    monkeypatch.setattr(tempfile, "tempdir", None)


def test_scoped_defaults_sit_between_the_environment_and_the_table(monkeypatch):
    """tuning.scoped: a caller's defaults for the duration of a block -- the environment still wins, an empty value leaves a
    switch alone, nesting restores what was there"""
    monkeypatch.delenv("IMS_PHOTON_LDS", raising=False)
    assert tuning.env("IMS_PHOTON_LDS") is None
    with tuning.scoped(IMS_PHOTON_LDS="41984"):
        assert tuning.number("IMS_PHOTON_LDS") == 41984 and tuning.library_tuning().photon_lds == 41984
        with tuning.scoped(IMS_PHOTON_LDS="1024"):
            assert tuning.number("IMS_PHOTON_LDS") == 1024
        assert tuning.number("IMS_PHOTON_LDS") == 41984
        with tuning.scoped(IMS_PHOTON_LDS=""):                   # empty: as elsewhere
            assert tuning.number("IMS_PHOTON_LDS") == 41984
        monkeypatch.setenv("IMS_PHOTON_LDS", "0")
        assert tuning.number("IMS_PHOTON_LDS") == 0              # an explicit choice of the user
        monkeypatch.delenv("IMS_PHOTON_LDS")
    assert tuning.env("IMS_PHOTON_LDS") is None and tuning.library_tuning().photon_lds == -1
    with pytest.raises(KeyError):
        tuning.scoped(IMS_NO_SUCH_SWITCH="1")
