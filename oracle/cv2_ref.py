"""CPU oracle for the OpenCV operators of the reference's job set-up.  TEST INFRASTRUCTURE ONLY.

The reference builds its pyramid and its structured-noise initial image with OpenCV
(neural_style_transfer.py:211-226 resize, :265-357 noise map / Sobel / blur / blend, :396-439 gaussian_mask,
make_style_noise).  `opencv-python` (pinned 4.8.1.78 in requirements-base.txt:2) is absent offline and in no wheelhouse, so
the reference cannot produce fixtures for these rows.  This file restates each operator from OpenCV's
published definition, written out tap by tap in plain numpy - no torch, no scipy - and `tests/test_oracle_cv2.py` holds
every one of them against an INDEPENDENT implementation of the same documented rule that is installed here:

  cv2.resize(..., INTER_CUBIC)   <->  torch.nn.functional.interpolate(mode="bicubic", align_corners=False, antialias=False)
                                      (ATen states it follows OpenCV's rule: UpSample.h:297-309; A = -0.75)
  cv2.Sobel(ksize=5)             <->  scipy.ndimage.correlate with the 5x5 outer product, mode="mirror" (= BORDER_REFLECT_101)
  cv2.getGaussianKernel          <->  scipy.signal.windows.gaussian, normalised
  cv2.GaussianBlur               <->  scipy.ndimage.correlate1d with that kernel, mode="mirror"

Parity status of rows f-1 / f-2 (round 3).  The four OPERATORS above: pinned by independent implementation (OpenCV itself
cannot run here).  The reference's own job-driver LOGIC restated in the second half of this file (level_size, gaussian_mask,
make_style_noise, noise_map, gradient_weight, initial_image): pinned by the reference itself - tests/golden/make_fixtures.py
runs the reference's neural_style_transfer() with a cv2 stand-in whose four functions call the operators of THIS file, and
tests/test_oracle_jobsetup.py holds these restatements to what the reference produced bit for bit (tests/golden/jobsetup.npz).
Only `tests/` (and that fixture generator) may import this module; the product's host mirror is
artstyletransfer_amd/host_image.py and the device kernels are artstyletransfer_amd/csrc/image_ops.hip - both are compared
with this file and with the reference-made fixture.
"""
from __future__ import annotations

import numpy as np

BASE_DIAMETER = 256        # neural_style_transfer.py:213
CUBIC_A = -0.75            # OpenCV's bicubic coefficient (imgproc: interpolateCubic)


# ---------------------------------------------------------------------------------------------------------------
# cv2.resize(img, (nw, nh), interpolation=cv2.INTER_CUBIC), float images (neural_style_transfer.py:226, :304-305, :427)
# ---------------------------------------------------------------------------------------------------------------
def cubic_taps(t: float):
    """The four weights of OpenCV's interpolateCubic for fractional offset t in [0, 1): taps at -1, 0, +1, +2."""
    A = CUBIC_A
    w0 = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
    w1 = ((A + 2) * t - (A + 3)) * t * t + 1
    w2 = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
    w3 = 1.0 - w0 - w1 - w2
    return w0, w1, w2, w3


def _axis_table(n_src: int, n_dst: int):
    """For every destination index: the 4 clamped source indices and weights.  Source coordinate of destination d:
    (d + 0.5) * n_src / n_dst - 0.5 (pixel centres); no antialiasing when shrinking; border replicated."""
    idx = np.empty((n_dst, 4), dtype=np.int64)
    wts = np.empty((n_dst, 4), dtype=np.float64)
    scale = n_src / n_dst
    for d in range(n_dst):
        f = (d + 0.5) * scale - 0.5
        i = int(np.floor(f))
        t = f - i
        wts[d] = cubic_taps(t)
        for k in range(4):
            idx[d, k] = min(max(i - 1 + k, 0), n_src - 1)
    return idx, wts


def resize_cubic(img: np.ndarray, nh: int, nw: int) -> np.ndarray:
    a = np.asarray(img, dtype=np.float64)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    h, w, _ = a.shape
    iy, wy = _axis_table(h, nh)
    ix, wx = _axis_table(w, nw)
    rows = np.zeros((nh, w, a.shape[2]))
    for k in range(4):
        rows += wy[:, k][:, None, None] * a[iy[:, k]]
    out = np.zeros((nh, nw, a.shape[2]))
    for k in range(4):
        out += wx[:, k][None, :, None] * rows[:, ix[:, k]]
    out = out.astype(np.asarray(img).dtype if np.asarray(img).dtype in (np.float32, np.float64) else np.float32)
    return out[..., 0] if squeeze else out


def level_size(height: int, width: int, level: int):
    """neural_style_transfer.py:215-224."""
    if height >= width:
        bw = BASE_DIAMETER
        bh = int(bw * (height / width))
    else:
        bh = BASE_DIAMETER
        bw = int(bh * (width / height))
    return bh * 2 ** level, bw * 2 ** level


# ---------------------------------------------------------------------------------------------------------------
# borders: BORDER_REFLECT_101 (OpenCV's BORDER_DEFAULT): gfedcb|abcdefgh|gfedcba
# ---------------------------------------------------------------------------------------------------------------
def reflect101(i: int, n: int) -> int:
    if n == 1:
        return 0
    while i < 0 or i >= n:
        i = -i if i < 0 else 2 * (n - 1) - i
    return i


def _correlate_axis(a: np.ndarray, k: np.ndarray, axis: int) -> np.ndarray:
    n = a.shape[axis]
    r = len(k) // 2
    out = np.zeros_like(a, dtype=np.float64)
    for j, kj in enumerate(k):
        src = np.array([reflect101(i + j - r, n) for i in range(n)])
        out += kj * np.take(a, src, axis=axis)
    return out


# cv2.Sobel(src, cv2.CV_64F, dx, dy, ksize=5) (neural_style_transfer.py:331-332): separable, derivative taps
# [-1,-2,0,2,1], smoothing taps [1,4,6,4,1] (getSobelKernels), correlation, BORDER_REFLECT_101
SOBEL5_D = np.array([-1.0, -2.0, 0.0, 2.0, 1.0])
SOBEL5_S = np.array([1.0, 4.0, 6.0, 4.0, 1.0])


def sobel5(img: np.ndarray, dx: int, dy: int) -> np.ndarray:
    a = np.asarray(img, dtype=np.float64)
    out = _correlate_axis(a, SOBEL5_D if dx else SOBEL5_S, axis=1)
    return _correlate_axis(out, SOBEL5_D if dy else SOBEL5_S, axis=0)


def get_gaussian_kernel(n: int, sigma: float) -> np.ndarray:
    """cv2.getGaussianKernel(n, sigma) for sigma > 0: exp(-(i - (n-1)/2)^2 / (2 sigma^2)), normalised to sum 1."""
    i = np.arange(n, dtype=np.float64) - (n - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return k / k.sum()


def gaussian_blur(img: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """cv2.GaussianBlur(src, (ksize, ksize), sigma): separable correlation with getGaussianKernel, BORDER_REFLECT_101."""
    k = get_gaussian_kernel(ksize, sigma)
    out = _correlate_axis(np.asarray(img, dtype=np.float64), k, axis=1)
    return _correlate_axis(out, k, axis=0)


# ---------------------------------------------------------------------------------------------------------------
# the reference's own arithmetic on top of those operators
# ---------------------------------------------------------------------------------------------------------------
def gaussian_mask(shape, central, peripheral, dispersion=0.5) -> np.ndarray:
    """neural_style_transfer.py:396-418: outer product of two Gaussian kernels (sigma = size * dispersion), divided by
    its [rows//2, cols//2] value, mask = p + g (c - p), repeated over 3 channels (float64)."""
    rows, cols = shape[:2]
    kx = get_gaussian_kernel(cols, cols * dispersion)
    ky = get_gaussian_kernel(rows, rows * dispersion)
    kernel = ky[:, None] * kx[None, :]
    norm = kernel / kernel[rows // 2, cols // 2]
    mask = peripheral + norm * (central - peripheral)
    return np.repeat(mask[:, :, None], 3, axis=2)


def make_style_noise(style_img: np.ndarray, targ_shape, rng=np.random) -> np.ndarray:
    """neural_style_transfer.py:422-439: the style image resized to the grid, its pixels permuted as whole rows of the
    (n, 3) array by np.random.permutation (the GLOBAL generator, as the reference uses)."""
    nh, nw = targ_shape[0], targ_shape[1]
    small = resize_cubic(style_img, nh, nw)
    vect = small.reshape(nh * nw, -1)
    return rng.permutation(vect).reshape(targ_shape)


def noise_map(style_top: np.ndarray, shape, noise_levels, central, peripheral, dispersion) -> np.ndarray:
    """neural_style_transfer.py:265-313 (float32 accumulator, float64 masks)."""
    nh, nw = shape[0], shape[1]
    acc = np.zeros(shape, dtype=np.float32)
    for gran, c, p, disp in zip(noise_levels, central, peripheral, dispersion):
        if gran == 0:
            acc += gaussian_mask(shape, c, p, disp)
            continue
        if gran > 0:
            dh, dw = (gran, nw * gran // nh) if nh <= nw else (nh * gran // nw, gran)
        else:
            dw, dh = nw // (-gran), nh // (-gran)
        low = make_style_noise(style_top, (dh, dw, shape[2]))
        hi = resize_cubic(low, nh, nw)
        acc += hi * gaussian_mask(hi.shape, c, p, disp)
    return acc


def gradient_weight(content_top: np.ndarray, noise_factor: float) -> np.ndarray:
    """neural_style_transfer.py:331-343: a nf / (a + blur(clip(sqrt(sx^2 + sy^2), 0, 100))), a = 5, blur 101 / sigma 0.2."""
    sx = np.absolute(sobel5(content_top, 1, 0))
    sy = np.absolute(sobel5(content_top, 0, 1))
    mag = np.clip(np.sqrt(sx * sx + sy * sy), 0.0, 100)
    mag = gaussian_blur(mag, 101, 0.2)
    a = 5.0
    return a * noise_factor / (a + mag)


def initial_image(init_method, content_img, style_img, content_top, style_top, top_level, noise_factor, noise_levels,
                  central, peripheral, dispersion):
    """neural_style_transfer.py:265-362 -> (float32 HWC image, tag)."""
    noise = noise_map(style_top, content_top.shape, noise_levels, central, peripheral, dispersion)
    weight = gradient_weight(content_top, noise_factor)
    if init_method == "random":
        return noise * 0.5, "random"
    if init_method == "content+noise":
        nh, nw = level_size(*content_img.shape[:2], top_level)
        base = resize_cubic(content_img, nh, nw)
        return ((1.0 - weight) * base + weight * noise).astype(np.float32), "content"
    nh, nw = level_size(*style_img.shape[:2], top_level)
    return resize_cubic(style_img, nh, nw), "style"
