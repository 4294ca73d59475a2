"""CPU tests of the image front-end restatement (oracle/orc_undistort.c) on the calibration files the reference ships
(tests/golden/calib/: copies of the reference's tests/*/camera*.txt, TUM-VI pcalib0.txt and vignette0.png)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd.records import CAM_EQUIDISTANT, CAM_PINHOLE, CAM_RADTAN, RECT_CROP

CAL = os.path.join(os.path.dirname(__file__), "golden", "calib")


def _text(name):
    with open(os.path.join(CAL, name)) as f:
        return f.read()


def read_pcalib():
    return np.array(_text("pcalib0.txt").split("\n")[0].split(), dtype=np.float32)


def read_vignette():
    from PIL import Image
    im = Image.open(os.path.join(CAL, "vignette0.png"))
    a = np.array(im)
    assert a.dtype == np.uint16 or a.max() > 255
    return a.astype(np.float32)


@pytest.mark.parametrize("name,model,wh", [("EuRoC_camera0.txt", CAM_RADTAN, (752, 480)), ("TUMVI_camera0.txt", CAM_EQUIDISTANT, (512, 512)),
                                           ("KITTI_0_2_camera0.txt", CAM_PINHOLE, (1232, 368)), ("Malaga_camera0.txt", CAM_PINHOLE, None),
                                           ("RobotCar_camera0.txt", CAM_PINHOLE, None)])
def test_reference_camera_files(name, model, wh):
    u = orc.Undistorter(_text(name))
    m = u.cam
    assert m.model == model and m.rect == RECT_CROP
    if wh:
        assert (m.w, m.h) == wh
    fx, fy, cx, cy = u.K
    # crop rectification: every output pixel maps inside the input image, and the border touches it somewhere
    assert np.all(u.remapX > 0) and np.all(u.remapY > 0) and np.all(u.remapX < m.wOrg - 1) and np.all(u.remapY < m.hOrg - 1)
    edge = min(u.remapX[:, 0].min(), (m.wOrg - 1 - u.remapX[:, -1]).min(), u.remapY[0].min(), (m.hOrg - 1 - u.remapY[-1]).min())
    assert edge < 0.02 * max(m.wOrg, m.hOrg), edge
    # the table is the distortion model applied to K^-1 (x, y): check the centre ray and the model's symmetry
    xc, yc = int(round(cx)), int(round(cy))
    assert abs(u.remapX[yc, xc] - (m.pars[2] + (xc - cx) / fx * m.pars[0])) < 0.6
    assert abs(u.remapY[yc, xc] - (m.pars[3] + (yc - cy) / fy * m.pars[1])) < 0.6
    assert 0.1 * m.w < fx < 3 * m.w and 0.2 * m.w < cx < 0.8 * m.w   # (the 195-degree TUM-VI lens crops to fx = 60)
    if model == CAM_PINHOLE:   # no distortion: the table is an affine map
        xs = np.arange(m.w, dtype=np.float64)
        assert np.allclose(u.remapX[5], m.pars[0] * (xs - cx) / fx + m.pars[2], atol=2e-3)


def test_relative_format_and_explicit_output():
    # EuRoC numbers in the relative format are rescaled by the image size and shifted by half a pixel
    u = orc.Undistorter(_text("EuRoC_camera0.txt"))
    assert abs(u.cam.pars[0] - 0.609912234 * 752) < 1e-9 and abs(u.cam.pars[2] - (0.488982713 * 752 - 0.5)) < 1e-9
    txt = "RadTan 458.654 457.296 367.215 248.375 -0.28340811 0.07395907 0.00019359 1.76187114e-05\n752 480\n0.5 0.8 0.5 0.5 0\n640 400\n"
    v = orc.Undistorter(txt)
    assert np.allclose(v.K, [0.5 * 640, 0.8 * 400, 0.5 * 640 - 0.5, 0.5 * 400 - 0.5])
    w = orc.Undistorter("Pinhole 400 400 319.5 239.5 0\n640 480\nnone\n640 480\n")
    assert w.passthrough and np.allclose(w.K, [400, 400, 319.5, 239.5])
    with pytest.raises(ValueError):
        orc.Undistorter("Pinhole 400 400 319.5 239.5 0\n640 480\nfull\n640 480\n")
    with pytest.raises(ValueError):
        orc.Undistorter("garbage\n640 480\ncrop\n640 480\n")


def test_undistortion_inverts_a_rendered_distortion():
    """Physical check: render a smooth scene through the RadTan model, undistort, compare with the pinhole rendering."""
    u = orc.Undistorter(_text("EuRoC_camera0.txt"))
    m = u.cam
    fx, fy, cx, cy = u.K

    def scene(a, b):   # irradiance as a function of the normalised ray
        return 128 + 60 * np.sin(7 * a) * np.cos(5 * b) + 30 * np.sin(11 * (a + b))

    # distorted input image: invert the model numerically per input pixel (fixed-point iteration on the RadTan equations)
    k1, k2, r1, r2 = [m.pars[i] for i in range(4, 8)]
    yy, xx = np.mgrid[0:m.hOrg, 0:m.wOrg].astype(np.float64)
    xd, yd = (xx - m.pars[2]) / m.pars[0], (yy - m.pars[3]) / m.pars[1]
    a, b = xd.copy(), yd.copy()
    for _ in range(30):
        rho2 = a * a + b * b
        rad = k1 * rho2 + k2 * rho2 * rho2
        a = (xd - 2 * r1 * a * b - r2 * (rho2 + 2 * a * a)) / (1 + rad)
        b = (yd - 2 * r2 * a * b - r1 * (rho2 + 2 * b * b)) / (1 + rad)
    raw = np.clip(np.rint(scene(a, b)), 0, 255).astype(np.uint8)
    out = u.frame(raw, exposure=0.0)          # no photometric calibration: factor * raw
    yo, xo = np.mgrid[0:m.h, 0:m.w].astype(np.float64)
    ideal = scene((xo - cx) / fx, (yo - cy) / fy)
    err = np.abs(out - ideal)[8:-8, 8:-8]
    assert np.median(err) < 0.6 and np.percentile(err, 99) < 3.0, (np.median(err), np.percentile(err, 99))


def test_photometric_calibration_tumvi():
    G, V = read_pcalib(), read_vignette()
    assert len(G) == 256 and V.shape == (512, 512)
    u = orc.Undistorter(_text("TUMVI_camera0.txt"), G=G, vignette=V, photometric_mode=2)
    assert u.valid == 1
    assert u.G[0] == 0 and u.G[255] == 255 and np.all(np.diff(u.G) > 0)
    lit = V.reshape(-1) > 0          # the fisheye circle: outside it the shipped vignette is 0 and 1/V = inf, as in the reference
    assert lit.mean() > 0.6 and np.all(np.isfinite(u.vinv[lit])) and u.vinv[lit].min() >= 1.0 and np.all(np.isinf(u.vinv[~lit]))
    rng = np.random.default_rng(1)
    raw = rng.integers(0, 256, (512, 512)).astype(np.uint8)
    full = u.frame(raw, exposure=0.01)
    plain = u.frame(raw, exposure=0.0)       # unknown exposure -> factor * raw (processFrame :203-209)
    no_v = orc.Undistorter(_text("TUMVI_camera0.txt"), G=G, vignette=V, photometric_mode=1).frame(raw, exposure=0.01)
    ok = np.isfinite(full)
    assert ok.mean() > 0.5 and np.all(full[ok] >= no_v[ok] - 1e-4) and (full[ok] > no_v[ok] + 1e-3).mean() > 0.5   # 1 / vignette >= 1
    assert np.allclose(no_v, plain, atol=1e-3)                                      # the shipped response is the identity
