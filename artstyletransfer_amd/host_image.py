"""Host-side image preparation around the hot path: pyramid resize and the structured-noise
initial image (reference: neural_style_transfer.py:211-226, :249-362, :396-439).

The reference does this with OpenCV, which is absent offline; the OpenCV operators it calls are
restated here from their documented semantics (pinned by independent implementations in
tests/test_oracle_cv2.py; the driver logic around them by what the reference's own
neural_style_transfer() produced, tests/test_oracle_jobsetup.py - see DESIGN.md section 2):
  cv2.resize(..., INTER_CUBIC)   bicubic, A = -0.75, half-pixel centres, replicate border, no antialias
                                 (the rule torch's bicubic follows: torch:include/ATen/native/UpSample.h:297-309)
  cv2.Sobel(ksize=5)             separable [-1,-2,0,2,1] x [1,4,6,4,1], BORDER_REFLECT_101
  cv2.getGaussianKernel(n, s)    exp(-(i-(n-1)/2)^2 / (2 s^2)), normalised to sum 1
  cv2.GaussianBlur               separable correlation with that kernel, BORDER_REFLECT_101
This runs once per job, before the optimisation loop; it is not on the timed path."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from scipy.ndimage import correlate1d

BASE_DIAMETER = 256   # neural_style_transfer.py:213


def level_size(height: int, width: int, level: int):
    """(new_height, new_width) of pyramid level `level` (neural_style_transfer.py:215-224)."""
    if height >= width:
        bw = BASE_DIAMETER
        bh = int(bw * (height / width))
    else:
        bh = BASE_DIAMETER
        bw = int(bh * (width / height))
    return bh * 2 ** level, bw * 2 ** level


def bicubic_resize(img: np.ndarray, new_height: int, new_width: int) -> np.ndarray:
    """cv2.resize(img, (new_width, new_height), interpolation=cv2.INTER_CUBIC) for float HWC images."""
    a = np.asarray(img)
    dt = a.dtype if a.dtype in (np.float32, np.float64) else np.float32
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=dt))
    squeeze = t.ndim == 2
    if squeeze:
        t = t.unsqueeze(-1)
    t = t.permute(2, 0, 1).unsqueeze(0)
    out = F.interpolate(t, size=(new_height, new_width), mode="bicubic", align_corners=False)
    out = out.squeeze(0).permute(1, 2, 0).contiguous().numpy()
    return out[..., 0] if squeeze else out


def resize_to_level(img: np.ndarray, level: int) -> np.ndarray:
    h, w = img.shape[:2]
    nh, nw = level_size(h, w, level)
    return bicubic_resize(img, nh, nw)


def gaussian_kernel(n: int, sigma: float) -> np.ndarray:
    i = np.arange(n, dtype=np.float64) - (n - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return k / k.sum()


def gaussian_mask(shape, central_amplitude, peripheral_amplitude, dispersion_scale=0.5) -> np.ndarray:
    """Gaussian envelope p + g_norm * (c - p), repeated over 3 channels, float64
    (neural_style_transfer.py:396-418)."""
    rows, cols = shape[:2]
    kx = gaussian_kernel(cols, cols * dispersion_scale)
    ky = gaussian_kernel(rows, rows * dispersion_scale)
    kernel = np.outer(ky, kx)
    norm = kernel / kernel[rows // 2, cols // 2]
    mask = peripheral_amplitude + norm * (central_amplitude - peripheral_amplitude)
    return np.repeat(mask[:, :, None], 3, axis=2)


def make_style_noise(style_img: np.ndarray, targ_shape) -> np.ndarray:
    """Style pixels resized to the low-res grid and row-shuffled with the global numpy RNG
    (neural_style_transfer.py:422-439)."""
    nh, nw = targ_shape[0], targ_shape[1]
    small = bicubic_resize(style_img.copy(), nh, nw)
    vect = small.reshape(nh * nw, -1)
    return np.random.permutation(vect).reshape(targ_shape)


_SOBEL_D = np.array([-1.0, -2.0, 0.0, 2.0, 1.0])
_SOBEL_S = np.array([1.0, 4.0, 6.0, 4.0, 1.0])


def sobel5(img: np.ndarray, dx: int, dy: int) -> np.ndarray:
    a = np.asarray(img, dtype=np.float64)
    kx = _SOBEL_D if dx else _SOBEL_S
    ky = _SOBEL_D if dy else _SOBEL_S
    out = correlate1d(a, kx, axis=1, mode="mirror")
    return correlate1d(out, ky, axis=0, mode="mirror")


def gaussian_blur(img: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    k = gaussian_kernel(ksize, sigma)
    out = correlate1d(np.asarray(img, dtype=np.float64), k, axis=1, mode="mirror")
    return correlate1d(out, k, axis=0, mode="mirror")


def noise_map(style_top: np.ndarray, shape, noise_levels, central, peripheral, dispersion) -> np.ndarray:
    """Multi-granularity style noise under Gaussian envelopes (neural_style_transfer.py:265-313)."""
    nh, nw = shape[0], shape[1]
    acc = np.zeros(shape, dtype=np.float32)
    for gran, c, p, disp in zip(noise_levels, central, peripheral, dispersion):
        if gran == 0:
            acc += gaussian_mask(shape, c, p, disp)
            continue
        if gran > 0:
            if nh <= nw:
                dh, dw = gran, nw * gran // nh
            else:
                dw, dh = gran, nh * gran // nw
        else:
            dw, dh = nw // (-gran), nh // (-gran)
        low = make_style_noise(style_top, (dh, dw, shape[2]))
        hi = bicubic_resize(low, nh, nw)
        acc += hi * gaussian_mask(hi.shape, c, p, disp)
    return acc


def gradient_weight(content_top: np.ndarray, noise_factor: float) -> np.ndarray:
    """a*noise_factor/(a + |sobel|) with a = 5 (neural_style_transfer.py:331-343)."""
    sx = np.absolute(sobel5(content_top, 1, 0))
    sy = np.absolute(sobel5(content_top, 0, 1))
    mag = np.clip(np.sqrt(sx * sx + sy * sy), 0.0, 100)
    mag = gaussian_blur(mag, 101, 0.2)
    a = 5.0
    return a * noise_factor / (a + mag)


def initial_image(init_method: str, content_img: np.ndarray, style_img: np.ndarray, content_top: np.ndarray,
                  style_top: np.ndarray, top_level: int, noise_factor, noise_levels, central, peripheral, dispersion):
    """(init image float32 HWC, name tag) per neural_style_transfer.py:265-362."""
    noise = noise_map(style_top, content_top.shape, noise_levels, central, peripheral, dispersion)
    weight = gradient_weight(content_top, noise_factor)
    if init_method == "random":
        return noise * 0.5, "random"
    if init_method == "content+noise":
        base = resize_to_level(content_img, top_level)
        return ((1.0 - weight) * base + weight * noise).astype(np.float32), "content"
    return resize_to_level(style_img, top_level), "style"
