"""Feature network surface of the reference (neural_nets.py:10-68) on top of the HIP engine.

`Vgg19` keeps the reference's attributes (`layer_names`, `content_feature_maps_index`,
`style_feature_maps_indices`) and returns the same `VggOutputs` namedtuple, but holds no
torch modules: the 13 frozen conv layers live, re-laid-out, inside a `StyleEngine` per GPU.
The reference downloads torchvision's pretrained checkpoint into TORCH_HOME = its own directory
(neural_nets.py:19, neural_style_transfer.py:8-10).  Here the file is only ever read from disk:
NST_VGG19_WEIGHTS (a path), else `$TORCH_HOME/hub/checkpoints/vgg19-*.pth`, else the same under the
directory of the drop-in modules and under ~/.cache/torch.  Without a file `load_weights` RAISES - a
switched-over deployment must not silently paint with random weights - unless NST_SYNTHETIC_WEIGHTS=1
opts into the seeded synthetic set (benchmarks, demos without a checkpoint)."""
from __future__ import annotations

import glob
import os
import threading
import warnings
from collections import namedtuple
from typing import Dict, List, Tuple

import torch

from . import synthetic
from .engine import StyleEngine

_FEATURE_CONV_INDICES = (0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28)   # torchvision vgg19.features
_weights_cache = None
_engines: Dict[int, StyleEngine] = {}
_lock = threading.Lock()


def state_dict_to_weights(sd) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """torchvision vgg19 state dict -> the 13 (weight, bias) pairs conv1_1..conv5_1 (features.0 ... features.28)."""
    ws = []
    for i, (cin, cout) in zip(_FEATURE_CONV_INDICES, synthetic.VGG19_CONV_SHAPES):
        try:
            w, b = sd[f"features.{i}.weight"], sd[f"features.{i}.bias"]
        except KeyError as e:
            raise KeyError(f"not a torchvision vgg19 state dict: {e} is missing") from e
        if tuple(w.shape) != (cout, cin, 3, 3) or tuple(b.shape) != (cout,):
            raise ValueError(f"features.{i}: expected weight {(cout, cin, 3, 3)} and bias {(cout,)}, got "
                             f"{tuple(w.shape)} and {tuple(b.shape)}")
        ws.append((w.detach().float().contiguous(), b.detach().float().contiguous()))
    return ws


def find_checkpoint() -> str | None:
    """The pretrained file where the reference's download would have put it (never fetched here)."""
    explicit = os.environ.get("NST_VGG19_WEIGHTS")
    if explicit:
        return explicit
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # directory of the drop-in modules
    homes = [os.environ.get("TORCH_HOME"), root, os.path.join(os.path.expanduser("~"), ".cache", "torch")]
    for home in homes:
        if not home:
            continue
        hits = sorted(glob.glob(os.path.join(home, "hub", "checkpoints", "vgg19-*.pth")))
        if hits:
            return hits[0]
    return None


def load_weights() -> List[Tuple[torch.Tensor, torch.Tensor]]:
    global _weights_cache
    if _weights_cache is not None:
        return _weights_cache
    path = find_checkpoint()
    if path:
        ws = state_dict_to_weights(torch.load(path, map_location="cpu"))
    elif os.environ.get("NST_SYNTHETIC_WEIGHTS") == "1":
        warnings.warn("NST_SYNTHETIC_WEIGHTS=1: using seeded synthetic VGG19 weights "
                      "(results are not artistically meaningful)")
        ws = synthetic.vgg19_weights()
    else:
        raise FileNotFoundError(
            "no VGG19 checkpoint found: set NST_VGG19_WEIGHTS to a torchvision vgg19 state-dict file, or put "
            "vgg19-*.pth under $TORCH_HOME/hub/checkpoints (where the reference's download caches it); "
            "NST_SYNTHETIC_WEIGHTS=1 opts into seeded synthetic weights (benchmarks only)")
    _weights_cache = ws
    return ws


def set_weights(weights) -> None:
    """Install (weight, bias) pairs conv1_1..conv5_1; drops engines built on the old ones."""
    global _weights_cache
    with _lock:
        _weights_cache = list(weights)
        for e in _engines.values():
            e.close()
        _engines.clear()
        for pool in _idle.values():
            for e in pool:
                e.close()
        _idle.clear()


_idle: Dict[int, List[StyleEngine]] = {}       # engines (contexts with the weights uploaded) waiting for their next job, per GPU
_IDLE_MAX = 2                                  # config.simultaneous_tasks_count jobs per GPU is the scheduler's default


def lease_engine(device) -> StyleEngine:
    """An engine for ONE job's targets and workspace (a LossBuilder, a job of the loop).  Contexts are re-used: building
    one means re-laying out and uploading 13 layers of weights (0.25 s, 260 MB), configuring one for the next job only
    re-allocates the pyramid workspace.  Give it back with `return_engine`."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        pool = _idle.get(idx)
        if pool:
            return pool.pop()
    return StyleEngine(load_weights(), idx)


def return_engine(eng: StyleEngine) -> None:
    if getattr(eng, "ctx", None) is None:
        return
    idx = eng.device.index
    with _lock:
        pool = _idle.setdefault(idx, [])
        keep = len(pool) < _IDLE_MAX and _weights_cache is not None and getattr(eng, "weights_id", None) == id(_weights_cache)
    if keep:
        try:
            eng.release_job()                  # the workspace goes back now, only the weights stay resident
        except Exception:
            keep = False
    if keep:
        with _lock:
            _idle.setdefault(idx, []).append(eng)
        return
    eng.close()


def shared_engine(device) -> StyleEngine:
    """The per-GPU engine used by the stand-alone helpers (gram_matrix, Vgg19.forward, ...)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        if idx not in _engines:
            _engines[idx] = StyleEngine(load_weights(), idx)
        return _engines[idx]


class Vgg19:
    """Only the layers the original NST paper uses are exposed (relu1_1, relu2_1, relu3_1, relu4_1,
    conv4_2, relu5_1); 'conv4_2' carries ReLU(conv4_2) exactly as the reference's in-place ReLU
    leaves it."""

    def __init__(self, requires_grad=False, show_progress=False, use_relu=True):
        if requires_grad:
            raise NotImplementedError("the feature network is frozen; only the image is optimised")
        if not use_relu:
            raise NotImplementedError("use_relu=False (pre-activation taps) is not on the reference's path")
        self.layer_names = ["relu1_1", "relu2_1", "relu3_1", "relu4_1", "conv4_2", "relu5_1"]
        self.offset = 1
        self.content_feature_maps_index = 4
        self.style_feature_maps_indices = [0, 1, 2, 3, 5]
        self.weights = load_weights()
        self.device = None

    def to(self, device):
        self.device = torch.device(device)
        return self

    def eval(self):
        return self

    def parameters(self):
        for w, b in self.weights:
            yield w
            yield b

    def forward(self, x: torch.Tensor):
        eng = shared_engine(x.device)
        outs = eng.vgg_features(x.contiguous())
        return namedtuple("VggOutputs", self.layer_names)(*outs)

    __call__ = forward
