"""GPU parity of the image front-end (sos_camera_parse / sos_undistort_*): parsed model, rectified K and remap table equal
to the oracle's for every camera file the reference ships and for synthetic FOV / KannalaBrandt / explicit-output / none
files; per frame the undistorted irradiance image and the whole pyramid behind it bit-identical to the oracle chain
(8- and 16-bit input, with and without photometric calibration)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.test_oracle_undistort import CAL, _text, read_pcalib, read_vignette

pytestmark = pytest.mark.gpu

FILES = [n for n in sorted(os.listdir(CAL)) if n.endswith(".txt") and "camera" in n]
SYNTH = {
    "fov": "FOV 0.535719308086809 0.669566858850269 0.493248545285398 0.500408664348414 0.897966326944875\n640 480\ncrop\n640 480\n",
    "kb": "KannalaBrandt 380.8 380.7 320.3 240.9 0.0034 0.0007 -0.002 0.0002\n640 480\ncrop\n512 384\n",
    "given": "RadTan 458.654 457.296 367.215 248.375 -0.28340811 0.07395907 0.00019359 1.76187114e-05\n752 480\n0.5 0.8 0.5 0.5 0\n640 400\n",
    "none": "Pinhole 400 400 319.5 239.5 0\n640 480\nnone\n640 480\n",
    "bare8": "458.654 457.296 367.215 248.375 -0.28340811 0.07395907 0.00019359 1.76187114e-05\n752 480\ncrop\n640 480\n",
    "bare5": "0.535719308086809 0.669566858850269 0.493248545285398 0.500408664348414 0\n640 480\ncrop\n640 480\n",
}


def _same_model(a, b):
    return (a.model, a.rect, a.wOrg, a.hOrg, a.w, a.h) == (b.model, b.rect, b.wOrg, b.hOrg, b.w, b.h) and list(a.pars) == list(b.pars) and \
        list(a.outCal) == list(b.outCal)


@pytest.mark.parametrize("text", [_text(n) for n in FILES] + list(SYNTH.values()), ids=FILES + list(SYNTH))
def test_setup_and_frames_match_oracle(text):
    from sos_slam_amd import lib
    o = orc.Undistorter(text)
    cam = lib.camera_parse(text)
    assert _same_model(cam, o.cam)
    ctx = lib.Context(cam.w, cam.h)
    u = lib.Undistorter(ctx, cam)
    K, rx, ry, pt = u.get()
    assert np.array_equal(K, o.K.astype(np.float32)) and pt == o.passthrough
    assert np.array_equal(rx, o.remapX) and np.array_equal(ry, o.remapY)
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:cam.hOrg, 0:cam.wOrg]
    base = 120 + 70 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + rng.normal(0, 6, (cam.hOrg, cam.wOrg))
    raw8 = np.clip(base, 0, 255).astype(np.uint8)
    raw16 = np.clip(base * 200, 0, 65535).astype(np.uint16)
    for raw, factor in ((raw8, 1.0), (raw16, 1.0 / 256)):
        img_g = u.frame(raw, exposure=0.0, slot=0, factor=factor)
        img_o = o.frame(raw, exposure=0.0, factor=factor)
        assert np.array_equal(img_g, img_o, equal_nan=True)
        dI_o, ab_o = orc.make_images(img_o)
        for lvl in range(len(dI_o)):
            dI_g, ab_g = ctx.download_level(0, lvl)
            assert np.array_equal(dI_g, dI_o[lvl], equal_nan=True) and np.array_equal(ab_g, ab_o[lvl], equal_nan=True), lvl
    u.close()
    ctx.close()


def test_photometric_modes_tumvi():
    from sos_slam_amd import lib
    text, G, V = _text("TUMVI_camera0.txt"), read_pcalib(), read_vignette()
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, (512, 512)).astype(np.uint8)
    cam = lib.camera_parse(text)
    ctx = lib.Context(cam.w, cam.h)
    for mode in (2, 1, 0):
        o = orc.Undistorter(text, G=G, vignette=V, photometric_mode=mode)
        u = lib.Undistorter(ctx, cam, G=G, vignette=V, photometric_mode=mode)
        for exposure in (0.0125, 0.0):
            a, b = u.frame(raw, exposure, slot=1), o.frame(raw, exposure)
            assert np.array_equal(a, b, equal_nan=True), (mode, exposure)
        u.close()
    # a response that is not strictly increasing invalidates the calibration, as in the reference
    Gbad = G.copy()
    Gbad[100] = Gbad[99]
    o = orc.Undistorter(text, G=Gbad, vignette=V)
    u = lib.Undistorter(ctx, cam, G=Gbad, vignette=V)
    assert o.valid == 0 and np.array_equal(u.frame(raw, 0.01, slot=1), o.frame(raw, 0.01), equal_nan=True)
    u.close()
    ctx.close()


def test_rejected_files():
    from sos_slam_amd import lib
    for bad in ("Pinhole 400 400 319.5 239.5 0\n640 480\nfull\n640 480\n", "garbage\n640 480\ncrop\n640 480\n", "Pinhole 1 2 3\n1 1\ncrop\n1 1\n"):
        with pytest.raises(RuntimeError):
            lib.camera_parse(bad)
