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


def vgg19_weights(seed: int = 1234) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    g = torch.Generator().manual_seed(seed)
    out = []
    for cin, cout in VGG19_CONV_SHAPES:
        w = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cout * 9))
        out.append((w, torch.zeros(cout)))
    return out


def image(h: int, w: int, seed: int) -> np.ndarray:
    lo = np.random.RandomState(seed).rand(max(h // 16, 1), max(w // 16, 1), 3).astype(np.float32)
    t = torch.from_numpy(lo).permute(2, 0, 1).unsqueeze(0)
    up = F.interpolate(t, size=(h, w), mode="bicubic", align_corners=False)
    return up.squeeze(0).permute(1, 2, 0).clamp(0.0, 1.0).contiguous().numpy()
