"""Per-object spectral energy distributions: flux through the bandpass and the wavelength distribution of the photons.

Host-side restatement of what InstCatalog.getSED / getObj do with galsim.SED (imsim/instcat.py:380-431, :563-573):
SED file (nm, f_lambda) -> photons/nm/cm^2/s, normalised to the magnorm = 0 flux density at 500 nm (:395-398), redshifted
(:405), multiplied by the Milky-Way extinction curve for (Av, Rv) (:407-419; internal extinction is NOT applied by the
reference either, :402-403), then `obj.withFlux(fAt) * sed` (:573) whose flux through the bandpass is the object's
nominal_flux (imsim/stamp.py:184) and whose product with the throughput is what the WavelengthSampler draws from.

Everything is vectorised over the objects that share an SED file.  Extinction: the reference uses
dust_extinction.F19, whose spline tables are external data absent here; the closed-form curve of Cardelli, Clayton &
Mathis 1989 (with O'Donnell's 1994 optical coefficients, the `CCM` model the catalogs name) stands in -- the two differ
by a few per cent of A(lambda) in the optical.
"""
import gzip
import os

import numpy as np

from .instcat import FLUX_DENSITY_500


def read_sed_file(path):
    """(wavelength [nm], f_lambda) columns of an SED library file (plain or .gz, `#` comments)."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        data = np.loadtxt(f, comments="#", usecols=(0, 1), dtype=np.float64)
    w, f = data[:, 0], data[:, 1]
    order = np.argsort(w, kind="stable")
    return w[order], f[order]


class SED:
    """Tabulated SED in photons / nm / cm^2 / s against REST wavelength [nm], linear interpolation, zero outside."""

    def __init__(self, wave, fphotons, redshift=0.0):
        self.wave = np.asarray(wave, dtype=np.float64)
        self.fphotons = np.asarray(fphotons, dtype=np.float64)
        self.redshift = float(redshift)

    @classmethod
    def from_file(cls, path):
        """galsim.SED(file, wave_type='nm', flux_type='flambda'): photons ~ f_lambda * lambda (the constant 1 / hc drops
        out in the normalisation that always follows)."""
        w, flam = read_sed_file(path)
        return cls(w, flam * w)

    def __call__(self, wave):
        """photons / nm / cm^2 / s at OBSERVED wavelengths (galsim.SED.__call__: the spectrum at wave / (1 + z))"""
        return np.interp(np.asarray(wave, dtype=np.float64) / (1.0 + self.redshift), self.wave, self.fphotons, left=0.0, right=0.0)

    def with_flux_density(self, target=FLUX_DENSITY_500, wave=500.0):
        cur = float(self(wave))
        if not cur > 0.0:
            raise ValueError("SED has no flux at the normalisation wavelength")
        return SED(self.wave, self.fphotons * (target / cur), self.redshift)

    def at_redshift(self, z):
        return SED(self.wave, self.fphotons, z)


def ccm89(wave_nm, rv=3.1):
    """A(lambda) / A(V) of Cardelli, Clayton & Mathis (1989) with O'Donnell (1994) in the optical; x = 1 / lambda [1/um]
    clipped to the curve's range 0.3 .. 10."""
    x = np.clip(1.0e3 / np.asarray(wave_nm, dtype=np.float64), 0.3, 10.0)
    a = np.zeros_like(x)
    b = np.zeros_like(x)
    ir = x < 1.1
    a[ir] = 0.574 * x[ir] ** 1.61
    b[ir] = -0.527 * x[ir] ** 1.61
    op = (x >= 1.1) & (x < 3.3)
    y = x[op] - 1.82
    a[op] = 1.0 + y * (0.104 + y * (-0.609 + y * (0.701 + y * (1.137 + y * (-1.718 + y * (-0.827 + y * (1.647 - 0.505 * y)))))))
    b[op] = y * (1.952 + y * (2.908 + y * (-3.989 + y * (-7.985 + y * (11.102 + y * (5.491 + y * (-10.805 + 3.347 * y)))))))
    uv = (x >= 3.3) & (x < 8.0)
    xu = x[uv]
    fa = np.where(xu >= 5.9, -0.04473 * (xu - 5.9) ** 2 - 0.009779 * (xu - 5.9) ** 3, 0.0)
    fb = np.where(xu >= 5.9, 0.2130 * (xu - 5.9) ** 2 + 0.1207 * (xu - 5.9) ** 3, 0.0)
    a[uv] = 1.752 - 0.316 * xu - 0.104 / ((xu - 4.67) ** 2 + 0.341) + fa
    b[uv] = -3.090 + 1.825 * xu + 1.206 / ((xu - 4.62) ** 2 + 0.263) + fb
    fuv = x >= 8.0
    yf = x[fuv] - 8.0
    a[fuv] = -1.073 + yf * (-0.628 + yf * (0.137 - 0.070 * yf))
    b[fuv] = 13.670 + yf * (4.257 + yf * (-0.420 + 0.374 * yf))
    return a + b / rv


def extinction_factor(wave_nm, av, rv):
    """10^(-0.4 A(lambda)) for arrays of objects: wave [n_w], av / rv [n_obj] -> [n_obj][n_w]"""
    av = np.atleast_1d(np.asarray(av, dtype=np.float64))
    rv = np.atleast_1d(np.asarray(rv, dtype=np.float64))
    out = np.ones((len(av), len(wave_nm)))
    for r in np.unique(rv):
        sel = rv == r
        out[sel] = 10.0 ** (-0.4 * av[sel, None] * ccm89(wave_nm, r)[None, :])
    return out


class SedLibrary:
    """Cache of normalised SEDs by file name, searched in sed_dir and then beside the catalog (instcat.py:388-393)."""

    def __init__(self, sed_dir=None, inst_dir=None):
        self.dirs = [d for d in (sed_dir, inst_dir) if d]
        self._cache = {}

    def find(self, name):
        for d in self.dirs:
            full = os.path.join(d, name)
            if os.path.isfile(full):
                return full
        return None

    def get(self, name):
        if name not in self._cache:
            full = self.find(name)
            self._cache[name] = SED.from_file(full).with_flux_density() if full else None
        return self._cache[name]


def object_spectra(names, redshift, mw_av, mw_rv, bandpass_wave, bandpass_thr, library, n_pts=257, step=0.5):
    """Per-object flux through the bandpass and inverse CDF of the photon wavelengths.

    names [n] SED file names, redshift / mw_av / mw_rv [n]; bandpass on its own grid.  Returns
    (flux [n]: photons / cm^2 / s for magnorm = 0 -- multiply by 10^(-0.4 magnorm) area exptime,
     tables [n][n_pts]: wavelength at u = 0 .. 1 of the cumulative sed(lambda) T(lambda),
     missing: sorted names of SED files that were not found -- their objects get flux = -1 and the caller's fallback)."""
    names = np.asarray(names, dtype=object)
    n = len(names)
    lo, hi = float(bandpass_wave[0]), float(bandpass_wave[-1])
    grid = np.arange(lo, hi + 0.5 * step, step)
    grid[-1] = min(grid[-1], hi)
    thr = np.interp(grid, bandpass_wave, bandpass_thr, left=0.0, right=0.0)
    flux = np.full(n, -1.0)
    tables = np.zeros((n, n_pts))
    u = np.linspace(0.0, 1.0, n_pts)
    dw = np.diff(grid)
    missing = set()
    redshift = np.asarray(redshift, dtype=np.float64)
    for name in sorted(set(names.tolist())):
        sed = library.get(name)
        sel = np.flatnonzero(names == name)
        if sed is None:
            missing.add(name)
            continue
        for a in range(0, len(sel), 4096):                            # bounded temporaries
            ii = sel[a:a + 4096]
            rest = grid[None, :] / (1.0 + redshift[ii, None])
            spec = np.interp(rest, sed.wave, sed.fphotons, left=0.0, right=0.0)
            dens = spec * extinction_factor(grid, mw_av[ii], mw_rv[ii]) * thr[None, :]
            seg = 0.5 * (dens[:, 1:] + dens[:, :-1]) * dw[None, :]
            cdf = np.concatenate([np.zeros((len(ii), 1)), np.cumsum(seg, axis=1)], axis=1)
            flux[ii] = cdf[:, -1]
            for k, i in enumerate(ii):
                c = cdf[k]
                if not c[-1] > 0.0:
                    tables[i] = np.linspace(lo, hi, n_pts)
                    continue
                c = c / c[-1]
                keep = np.concatenate([[True], np.diff(c) > 0])
                tables[i] = np.interp(u, c[keep], grid[keep])
    return flux, tables, sorted(missing)
