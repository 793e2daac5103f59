"""Seeded synthetic inputs for benchmarks, smoke tests and demos: there is no network for the
pretrained VGG19 file or for datasets.  Same generators as the test oracle uses."""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

VGG19_CONV_SHAPES = ((3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 256),
                     (256, 512), (512, 512), (512, 512), (512, 512), (512, 512))


def vgg19_weights(seed: int = 1234, bias_std: float = 0.0) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """Kaiming fan-out weights, b = 0 (the bench workload).  bias_std > 0 draws seeded biases from a
    second generator (the parity tests use 2.0: pretrained VGG19 biases are not zero)."""
    g = torch.Generator().manual_seed(seed)
    gb = torch.Generator().manual_seed(seed + 1)
    out = []
    for cin, cout in VGG19_CONV_SHAPES:
        w = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cout * 9))
        b = torch.randn(cout, generator=gb) * bias_std if bias_std else torch.zeros(cout)
        out.append((w, b))
    return out


def image(h: int, w: int, seed: int) -> np.ndarray:
    lo = np.random.RandomState(seed).rand(max(h // 16, 1), max(w // 16, 1), 3).astype(np.float32)
    t = torch.from_numpy(lo).permute(2, 0, 1).unsqueeze(0)
    up = F.interpolate(t, size=(h, w), mode="bicubic", align_corners=False)
    return up.squeeze(0).permute(1, 2, 0).clamp(0.0, 1.0).contiguous().numpy()
