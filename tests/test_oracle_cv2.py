"""CPU: the OpenCV-operator oracle (oracle/cv2_ref.py, a tap-by-tap numpy restatement) against INDEPENDENT
implementations of the same documented rules - torch's bicubic kernel, scipy.ndimage, scipy.signal - and the product's
host mirror (artstyletransfer_amd/host_image.py) against that oracle.  This is the pin of SURVEY rows f-1 / f-2: OpenCV
itself is absent offline, so the reference cannot produce fixtures for them."""
import numpy as np
import pytest
import scipy.ndimage as ndi
import scipy.signal
import torch
import torch.nn.functional as F

from oracle import cv2_ref


def _torch_bicubic(img, nh, nw):
    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float64)).permute(2, 0, 1).unsqueeze(0)
    out = F.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False, antialias=False)
    return out.squeeze(0).permute(1, 2, 0).numpy()


@pytest.mark.parametrize("h,w,nh,nw", [(20, 30, 40, 60), (64, 96, 32, 48), (37, 53, 256, 367), (256, 383, 9, 13), (9, 13, 256, 384),
                                       (50, 50, 50, 50), (31, 17, 30, 18), (5, 7, 64, 3)])
def test_resize_cubic_vs_torch_bicubic(h, w, nh, nw):
    """Arbitrary up- and down-scales, not only the 1/2 of the pyramid: torch's upsample_bicubic2d (A = -0.75, half-pixel
    centres, clamped indices, no antialias) is an independent implementation of the rule cv2.INTER_CUBIC documents."""
    img = np.random.RandomState(h * 31 + w).rand(h, w, 3)
    ours = cv2_ref.resize_cubic(img, nh, nw)
    np.testing.assert_allclose(ours, _torch_bicubic(img, nh, nw), rtol=0, atol=1e-12)


def test_cubic_taps_known_answers():
    # t = 0: the pixel itself; t = 0.5: the fixed half-sample filter [-3, 19, 19, -3] / 32
    assert cv2_ref.cubic_taps(0.0) == pytest.approx((0.0, 1.0, 0.0, 0.0), abs=1e-15)
    assert cv2_ref.cubic_taps(0.5) == pytest.approx((-0.09375, 0.59375, 0.59375, -0.09375), abs=1e-15)
    for t in np.linspace(0, 0.99, 12):
        assert sum(cv2_ref.cubic_taps(float(t))) == pytest.approx(1.0, abs=1e-14)
    img = np.random.RandomState(0).rand(12, 9, 3)
    np.testing.assert_allclose(cv2_ref.resize_cubic(img, 12, 9), img, atol=1e-14)         # identity scale
    assert cv2_ref.resize_cubic(img.astype(np.float32), 5, 4).dtype == np.float32


@pytest.mark.parametrize("shape", [(9, 12), (16, 16), (5, 40), (3, 4)])
def test_sobel5_vs_scipy_correlate(shape):
    """cv2.Sobel(ksize=5): the 5x5 kernel is the outer product of [1,4,6,4,1] and [-1,-2,0,2,1]; BORDER_REFLECT_101 is
    scipy's mode="mirror".  scipy.ndimage.correlate with the full 2-D kernel is an independent implementation."""
    img = np.random.RandomState(shape[0]).rand(*shape, 3)
    kx = np.outer(cv2_ref.SOBEL5_S, cv2_ref.SOBEL5_D)          # rows: smoothing (y), columns: derivative (x)
    ky = np.outer(cv2_ref.SOBEL5_D, cv2_ref.SOBEL5_S)
    for c in range(3):
        np.testing.assert_allclose(cv2_ref.sobel5(img, 1, 0)[..., c], ndi.correlate(img[..., c], kx, mode="mirror"), atol=1e-12)
        np.testing.assert_allclose(cv2_ref.sobel5(img, 0, 1)[..., c], ndi.correlate(img[..., c], ky, mode="mirror"), atol=1e-12)
    # scipy's own Sobel-type separable filters agree on the border rule: reflect101 of an index sequence
    assert [cv2_ref.reflect101(i, 5) for i in range(-3, 8)] == [3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1]
    ramp = np.tile(np.arange(12, dtype=np.float64), (9, 1))[..., None]
    assert cv2_ref.sobel5(ramp, 1, 0)[4, 5, 0] == pytest.approx(128.0)     # (2 + 2 + 2 + 2) * 16
    assert np.abs(cv2_ref.sobel5(ramp, 0, 1)).max() == pytest.approx(0.0)


@pytest.mark.parametrize("n,sigma", [(5, 1.0), (7, 1.5), (101, 0.2), (256, 51.2), (383, 229.8), (2, 0.4)])
def test_gaussian_kernel_vs_scipy_window(n, sigma):
    k = cv2_ref.get_gaussian_kernel(n, sigma)
    ref = scipy.signal.windows.gaussian(n, sigma)
    np.testing.assert_allclose(k, ref / ref.sum(), rtol=1e-13, atol=1e-300)
    assert k.sum() == pytest.approx(1.0)
    if (n, sigma) == (5, 1.0):           # the values OpenCV's documentation example prints
        np.testing.assert_allclose(k, [0.05448868, 0.24420134, 0.40261995, 0.24420134, 0.05448868], atol=5e-9)


def test_gaussian_blur_vs_scipy():
    img = np.random.RandomState(3).rand(20, 28, 3)
    for ksize, sigma in ((101, 0.2), (9, 1.7), (5, 0.8)):
        k = cv2_ref.get_gaussian_kernel(ksize, sigma)
        ref = ndi.correlate1d(ndi.correlate1d(img, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
        np.testing.assert_allclose(cv2_ref.gaussian_blur(img, ksize, sigma), ref, atol=1e-13)
    # the reference's blur (101 taps, sigma 0.2) is the identity to 4e-6: exp(-1/(2*0.04)) = 3.7e-6
    assert np.abs(cv2_ref.gaussian_blur(img, 101, 0.2) - img).max() < 2e-5         # 4 neighbours x 3.7e-6


def test_host_mirror_vs_oracle():
    """artstyletransfer_amd/host_image.py (what the product's `resize`, `gaussian_mask`, `make_style_noise` are, and what
    the device kernels were first compared with) against the cross-checked oracle: every operator and the whole job
    set-up - pyramid level sizes, noise map (same numpy RNG stream), blend weight, the three init methods."""
    from artstyletransfer_amd import config, host_image as hi
    rs = np.random.RandomState(5)
    img = rs.rand(37, 53, 3).astype(np.float32)
    # fp32 implementations (torch's kernel here, OpenCV's own float path, the device kernel) hold the source coordinate
    # (d + 0.5) * scale - 0.5 in fp32: its rounding - half an ulp of a coordinate f, i.e. 6e-8 f, twice where the
    # coordinate is also FORMED in fp32 as torch does - moves the four weights by as much and the result by up to that
    # times the pixel contrast (white noise here): measured 3.7e-6 on a 37x53 -> 18x26 shrink, 1.9e-5 on 150x200 -> 256x341.
    # The oracle works in double.
    tol = lambda h, w: 2 * (2e-6 + 1.5e-7 * max(h, w))
    for nh, nw in ((74, 106), (18, 26), (256, 367), (9, 13)):
        np.testing.assert_allclose(hi.bicubic_resize(img, nh, nw), cv2_ref.resize_cubic(img, nh, nw), atol=tol(37, 53))
    for args in ((2875, 4312, 0), (2875, 4312, 2), (391, 470, 0), (500, 500, 1), (300, 200, 0)):
        assert hi.level_size(*args) == cv2_ref.level_size(*args)
    np.testing.assert_allclose(hi.sobel5(img, 1, 0), cv2_ref.sobel5(img, 1, 0), atol=1e-12)
    np.testing.assert_allclose(hi.sobel5(img, 0, 1), cv2_ref.sobel5(img, 0, 1), atol=1e-12)
    np.testing.assert_allclose(hi.gaussian_kernel(101, 0.2), cv2_ref.get_gaussian_kernel(101, 0.2), atol=1e-16)
    np.testing.assert_allclose(hi.gaussian_blur(img, 101, 0.2), cv2_ref.gaussian_blur(img, 101, 0.2), atol=1e-13)
    np.testing.assert_allclose(hi.gaussian_mask((40, 60, 3), 0.3, 0.2, 0.2), cv2_ref.gaussian_mask((40, 60, 3), 0.3, 0.2, 0.2), atol=1e-15)
    np.testing.assert_allclose(hi.gradient_weight(img, 0.95), cv2_ref.gradient_weight(img, 0.95), atol=1e-12)
    cfg = config.Config()
    content = rs.rand(150, 200, 3).astype(np.float32)
    style = rs.rand(90, 140, 3).astype(np.float32)
    ct, st = hi.resize_to_level(content, 0), hi.resize_to_level(style, 0)
    np.testing.assert_allclose(ct, cv2_ref.resize_cubic(content, *cv2_ref.level_size(150, 200, 0)), atol=tol(150, 200))
    args = (cfg.noise_factor, cfg.noise_levels, cfg.noise_levels_central_amplitude, cfg.noise_levels_peripheral_amplitude,
            cfg.noise_levels_dispersion)
    for method in ("content+noise", "random", "style"):
        np.random.seed(11)
        a, ta = hi.initial_image(method, content, style, ct, st, 0, *args)
        np.random.seed(11)
        b, tb = cv2_ref.initial_image(method, content, style, ct, st, 0, *args)
        assert ta == tb and a.shape == b.shape
        np.testing.assert_allclose(a, b, atol=tol(256, 341))
    # tall image: the other branch of the grid-size rule; a negative granularity
    np.random.seed(3)
    a = hi.noise_map(style, (96, 64, 3), (5, -2, 0), (0.3, 0.2, 0.2), (0.2, 0.1, 0.0), (0.2, 0.6, 0.3))
    np.random.seed(3)
    b = cv2_ref.noise_map(style, (96, 64, 3), (5, -2, 0), (0.3, 0.2, 0.2), (0.2, 0.1, 0.0), (0.2, 0.6, 0.3))
    np.testing.assert_allclose(a, b, atol=tol(96, 64))
