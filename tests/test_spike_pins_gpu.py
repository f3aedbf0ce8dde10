"""Pins against numbers the REFERENCE produced: tests/data/fft-diffraction/raytrace_diffraction_values_{0,300}_exptime.npz
hold the spike statistics of imSim's own ray-traced (batoid + GalSim) rendering of one bright star through
RubinDiffractionOptics (generate_reference_data_from_raytracing, tests/test_diffraction_fft.py:275-291): centre,
folded spike angle and its spread, log-log slope and intercept of the radial brightness.  The same star is rendered
here by the HIP photon path and by the FFT spike path, the statistics are computed with the reference's estimators
(tests/spike_stats.py, restated from tests/test_diffraction_fft.py:515-607) and compared with the stored values under the
reference's own tolerances (:363-420).

THE ONE DISAGREEMENT WITH REFERENCE-HELD NUMBERS IS IN THIS FILE, and it is an `xfailed` in the GPU suite, not a green: with the
pupil the reference's config samples (PupilAnnulusSampler 2.55 .. 4.18, :85) the spread of the folded angle and the intercept are
NOT met (15.9 deg against the stored 2.73, -1.59 against -2.69 at exptime 0; profiles/round6_spike_sweep.log has every number).
The sweep bounds it: the photon count does not matter (1e6 .. 2e7), the wavelength mix does not (flat r band, 577.6 nm, Vega-like),
sampling inside ONE rim does not (inner rim: nothing; outer rim by >= 10 mm: slope and intercept, not the spread), sampling >= 10 mm
inside BOTH rims meets all five, and at 20 mm the stored numbers are reproduced to 0.2 deg / 0.02 / 0.07 for both exposure times."""
import math

import numpy as np
import pytest

import spike_stats as ss
from imsim_amd import _abi

pytestmark = pytest.mark.gpu
ROTTELPOS, ALT, AZ = ss.ROTTELPOS, ss.ALT, ss.AZ
_scene, _star = ss.scene, ss.star
center_of_brightness, folded_spike_angle, radial_brightness_asymptotics = (ss.center_of_brightness, ss.folded_spike_angle,
                                                                           ss.radial_brightness_asymptotics)
GOLD = ss.GOLD
_CACHE = {}


def _stats(exptime, r_outer=ss.R_OUTER, r_inner=ss.R_INNER):
    key = (exptime, r_outer, r_inner)
    if key not in _CACHE:
        ref = ss.stored(exptime)
        img = ss.render(ss.scene(exptime, r_outer, r_inner), ref)
        assert img.sum() > 4.0e6
        _CACHE[key] = (ss.image_stats(img), ss.stored_stats(ref))
    return _CACHE[key]


@pytest.mark.parametrize("exptime", [0.0, 300.0])
def test_nominal_pupil_meets_the_stored_centre_and_angle_and_the_inverse_square_law(exptime):
    """With the pupil of the reference's config: the stored centre (2 px, :383-386), the stored spike angle (1 deg; at exptime 300 s
    it carries the field rotation) and at exptime 0 the expected 45 deg - rotTelPos (:388-396), brightness ~ 1 / r^2 (:314: slope
    -2 +- 0.2; the reference holds its stored slope to -2 +- 0.6, :408-409) with bounded standard errors (:419-420)."""
    got, want = _stats(exptime)
    np.testing.assert_allclose(got["c"], want["c"], atol=2.0, rtol=0.0)
    if exptime == 0.0:
        np.testing.assert_allclose(got["angle_deg"], 45.0 - ROTTELPOS, atol=1.0)
    np.testing.assert_allclose(got["angle_deg"], want["angle_deg"], atol=1.0)
    np.testing.assert_allclose(got["slope"], -2.0, atol=0.2)
    np.testing.assert_allclose(want["slope"], -2.0, atol=0.6)
    assert got["slope_stderr"] < 0.2 and got["intercept_stderr"] < 0.8


@pytest.mark.xfail(strict=True, reason="KNOWN DISAGREEMENT with reference-held numbers: the rim zones of the pupil put an isotropic 1 / r^2 "
                                       "halo into this build's image that the reference's stored statistics do not show "
                                       "(profiles/round6_spike_sweep.log; README, 'Open question')")
@pytest.mark.parametrize("stat,tol", [("angle_stddev_deg", 2.0), ("intercept", 0.5)])
@pytest.mark.parametrize("exptime", [0.0, 300.0])
def test_nominal_pupil_against_the_stored_spread_and_intercept(exptime, stat, tol):
    """The two stored statistics that are NOT met with the nominal pupil, under the reference's tolerances (:400-405 spread of the
    folded angle 2 deg, :417-418 intercept 0.5).  strict: the day this passes, the disagreement is gone and the marker must go."""
    got, want = _stats(exptime)
    np.testing.assert_allclose(got[stat], want[stat], atol=tol, rtol=0.0)


@pytest.mark.parametrize("exptime", [0.0, 300.0])
def test_pupil_sampled_two_centimetres_inside_both_rims_meets_all_five_stored_statistics(exptime):
    """What the stored numbers ARE consistent with: no light from within ~2 cm of either pupil rim.  Sampling 2.58 .. 4.16 all five
    are met under the reference's tolerances (and the slope to 0.1) -- which pins the strut geometry, the kick law
    phi* = atan(lambda / 4 pi delta) through the ray trace, the plate scale and the field-rotation rate.  This is an EMULATION of
    the reference's image, not the configured pupil; what removes the rim light in the reference is not established (by the
    public design values no surface behind M1 clips the on-axis beam of the rims; LSST_r.yaml is external data; the stored files
    are written once and kept, tests/test_diffraction_fft.py:363-365, so they may also predate the present geometry)."""
    got, want = _stats(exptime, 4.16, 2.58)
    ok = ss.within(got, want)
    assert all(ok.values()), (ok, got, want)
    assert got["slope_stderr"] < 0.2 and got["intercept_stderr"] < 0.8


def test_fft_spikes_match_the_references_ray_tracing(tmp_path):
    """tests/test_diffraction_fft.py:353-420: the FFT spike stencil against the stored ray-tracing statistics."""
    from imsim_amd import fft_draw
    from imsim_amd.diffraction_fft import DiffractionFFT
    from imsim_amd.engine import Renderer
    ref = ss.stored(0.0)
    scene = _scene(0.0)
    scene.ops = []
    objects = _star(scene, ref["c"], 6_000_000)
    r = Renderer(scene)
    rows, _ = fft_draw.build_fft_objects(objects, np.array([6.0e6]), np.array([-1]))
    wl_eff = 622.2
    d = DiffractionFFT(exptime=0.0, azimuth=math.radians(AZ), altitude=math.radians(ALT), rotTelPos=math.radians(ROTTELPOS),
                       brightness_threshold=1.0e5)
    fft_draw.FftDrawer(r, [(_abi.IMS_KPSF_GAUSSIAN, 0, 0.3 / 2.3548200450309493)], add_noise=False, diffraction_fft=d,
                       wavelength=wl_eff).draw(rows)
    r.synchronize()
    img = r.image.cpu().numpy()
    # the estimators take "image > 0" as "light": numerical residue of the transform (here below 1e-9 electrons per
    # pixel, far under what GalSim's float32 stamps resolve next to a 2e6-electron peak) is not light
    img = np.where(img > 1.0e-6, img, 0.0)
    c = center_of_brightness(img)
    np.testing.assert_allclose(c, ref["c"], atol=2.0, rtol=0.0)
    angle, angle_std = folded_spike_angle(img, c[0], c[1], r_min=10.0)
    np.testing.assert_allclose(np.rad2deg(angle), 45.0 - ROTTELPOS, atol=1.0)
    np.testing.assert_allclose(np.rad2deg(angle_std), np.rad2deg(ref["angle_stddev"]), atol=2.0)
    slope, intercept, slope_err, intercept_err = radial_brightness_asymptotics(img, c[0], c[1])
    np.testing.assert_allclose(slope, -2.0, atol=0.6)                                   # :408
    s = 577.6 / wl_eff                                                                   # diffraction_fft.WAVELENGTH / effective wavelength
    np.testing.assert_allclose(intercept, ref["intercept"] - math.log(s), atol=0.5)     # :411-418
