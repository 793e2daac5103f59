"""math_utils surface of the reference (math_utils.py:9-47) on the HIP engine."""
from __future__ import annotations

import math
from functools import reduce

import torch

from .neural_nets import Vgg19, shared_engine


def prepare_model(model, device):
    """(net, content index 4, style indices [0,1,2,3,5]) - math_utils.py:9-23."""
    if model == "vgg19":
        net = Vgg19(requires_grad=False, show_progress=True)
    else:
        raise ValueError(f"{model} not supported.")
    return net.to(device).eval(), net.content_feature_maps_index, net.style_feature_maps_indices


def gram_matrix(x: torch.Tensor, should_normalize=True) -> torch.Tensor:
    """(b, ch, ch) Gram matrices, divided by ch*h*w when normalising - math_utils.py:26-34."""
    eng = shared_engine(x.device)
    return torch.cat([eng.gram(x[i:i + 1].contiguous(), normalize=should_normalize) for i in range(x.shape[0])])


def total_variation(y: torch.Tensor) -> torch.Tensor:
    """mean|dx|^2 + mean|dy|^2 over the whole (b,c,h,w) tensor - math_utils.py:37-41."""
    return shared_engine(y.device).total_variation(y.contiguous())[0]


def regularization(y: torch.Tensor) -> torch.Tensor:
    """Unused by the reference's loss (math_utils.py:44-47); kept for surface parity."""
    els = reduce(lambda a, b: a * b, y.shape)
    return torch.sum(torch.pow(y / 128.0, 10)) / math.pow(els, 10)
