"""Job set-up on the GPU: the pyramid of content / style images and the
structured-noise initial image of the reference's job driver (neural_style_transfer.py:249-362), built from
the device kernels of image_ops.hip instead of OpenCV on the host.  `host_image.py` is the host restatement of
the same algorithm; `tests/test_hip_parity.py` holds the two together.  Random numbers stay on the host:
`make_style_noise` draws its row permutation from the global numpy generator exactly as the reference does,
only the index vector travels to the device."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import host_image
from .engine import StyleEngine


def upload(eng: StyleEngine, img: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).to(eng.device)


def pyramid(eng: StyleEngine, img_dev: torch.Tensor, levels_num: int) -> List[torch.Tensor]:
    """Levels highest resolution first, each resized from the original (reference :249-263)."""
    h, w = img_dev.shape[:2]
    out = []
    for level in range(levels_num - 1, -1, -1):
        nh, nw = host_image.level_size(h, w, level)
        out.append(eng.resize(img_dev, nh, nw))
    return out


def noise_map(eng: StyleEngine, style_top: torch.Tensor, shape: Tuple[int, int, int], noise_levels: Sequence[int],
              central: Sequence[float], peripheral: Sequence[float], dispersion: Sequence[float]) -> torch.Tensor:
    """Multi-granularity style-pixel noise under Gaussian envelopes (reference :265-313)."""
    nh, nw, ch = shape
    acc = torch.zeros((nh, nw, ch), dtype=torch.float32, device=eng.device)
    for gran, c, p, disp in zip(noise_levels, central, peripheral, dispersion):
        if gran == 0:
            eng.gaussian_mask_accumulate(acc, None, c, p, disp)
            continue
        if gran > 0:
            if nh <= nw:
                dh, dw = gran, nw * gran // nh
            else:
                dw, dh = gran, nh * gran // nw
        else:
            dw, dh = nw // (-gran), nh // (-gran)
        low = eng.resize(style_top, dh, dw)
        # the reference shuffles the (dh*dw, 3) pixel rows with np.random.permutation(array): same index stream
        perm = torch.from_numpy(np.random.permutation(dh * dw).astype(np.int64)).to(eng.device)
        low = eng.gather_rows(low.reshape(dh * dw, ch), perm).reshape(dh, dw, ch)
        eng.gaussian_mask_accumulate(acc, eng.resize(low, nh, nw), c, p, disp)
    return acc


def initial_image(eng: StyleEngine, init_method: str, content: torch.Tensor, style: torch.Tensor,
                  content_top: torch.Tensor, style_top: torch.Tensor, top_level: int, noise_factor, noise_levels,
                  central, peripheral, dispersion):
    """(device HWC float32 initial image, tag) - reference :265-362."""
    noise = noise_map(eng, style_top, tuple(content_top.shape), noise_levels, central, peripheral, dispersion)
    if init_method == "random":
        return eng.scale(noise, 0.5), "random"
    if init_method == "content+noise":
        return eng.noise_blend(content_top, noise, float(noise_factor)), "content"
    h, w = style.shape[:2]
    nh, nw = host_image.level_size(h, w, top_level)
    return eng.resize(style, nh, nw), "style"
