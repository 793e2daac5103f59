"""Thin Python handle on the C ABI: device buffers are torch CUDA tensors (plumbing only), every
computation happens in libnst_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import NST_LOSS_ROW, NstError, StepInfo

TAP_CHANNELS = (64, 128, 256, 512, 512, 512)
TAP_SCALE = (0, 1, 2, 3, 3, 4)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _chk_dev(t: torch.Tensor, device, shape=None):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise NstError("expected a contiguous float32 CUDA tensor")
    if t.device != device:
        raise NstError(f"tensor is on {t.device}, context on {device}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise NstError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")


class StyleEngine:
    """One nst_ctx: VGG19 weights on one GPU + the pyramid workspace of one job."""

    def __init__(self, weights: Sequence[Tuple[torch.Tensor, torch.Tensor]], device: int | str | torch.device = 0,
                 conv_mode: Optional[str] = None, batched: Optional[bool] = None, single_stream: Optional[bool] = None,
                 use_graph: Optional[bool] = None, h2_band_rows: Optional[int] = None, lbfgs_gram: Optional[bool] = None,
                 h2_mfma16: Optional[bool] = None, h2_wg256: Optional[bool] = None,
                 h2_tile_rows: Optional[int] = None, gram_overlap: Optional[bool] = None,
                 h2_persist: Optional[bool] = None, level_split: Optional[bool] = None,
                 h2_winograd: Optional[bool] = None):
        """Options (nst_options): None = environment variable (NST_CONV, NST_BATCH, NST_SINGLE_STREAM, NST_GRAPH,
        NST_H2_BAND_ROWS, NST_LBFGS_GRAM, NST_H2_MFMA16; read once, here) and otherwise the default (f16x2, batched, ...)."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise NstError("no GPU visible: the style-transfer hot path runs only on the HIP device")
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.type != "cuda":
            raise NstError("StyleEngine needs a cuda (HIP) device; there is no CPU fallback")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        if len(weights) != _lib.NST_VGG19_CONVS:
            raise NstError("expected 13 (weight, bias) pairs conv1_1..conv5_1")
        ws = [np.ascontiguousarray(w.detach().cpu().numpy(), dtype=np.float32) for w, _ in weights]
        bs = [np.ascontiguousarray(b.detach().cpu().numpy(), dtype=np.float32) for _, b in weights]
        wp = (C.c_void_p * 13)(*[a.ctypes.data for a in ws])
        bp = (C.c_void_p * 13)(*[a.ctypes.data for a in bs])
        opts = _lib.Options()
        self.lib.nst_options_default(C.byref(opts))
        if conv_mode is not None:
            if conv_mode not in _lib.CONV_MODES:
                raise NstError(f"conv_mode must be one of {sorted(_lib.CONV_MODES)}")
            opts.conv_mode = _lib.CONV_MODES[conv_mode]
        for name, val in (("batched", batched), ("single_stream", single_stream), ("use_graph", use_graph),
                          ("h2_band_rows", h2_band_rows), ("lbfgs_gram", lbfgs_gram), ("h2_mfma16", h2_mfma16),
                          ("h2_wg256", h2_wg256), ("h2_tile_rows", h2_tile_rows),
                          ("gram_overlap", gram_overlap), ("h2_persist", h2_persist), ("level_split", level_split), ("h2_winograd", h2_winograd)):
            if val is not None:
                setattr(opts, name, int(val))
        ctx = C.c_void_p()
        _lib.check(None, self.lib.nst_ctx_create_ex(idx, wp, bp, C.byref(opts), C.byref(ctx)), "nst_ctx_create_ex")
        self.ctx = ctx
        self.weights_id = id(weights)          # which weight set this context carries (neural_nets' engine pool)
        self.levels = 0
        self.shape = None

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.nst_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- job ------------------------------------------------------------------------------------
    def configure(self, levels_num: int, H0: int, W0: int) -> None:
        _lib.check(self.ctx, self.lib.nst_job_configure(self.ctx, levels_num, H0, W0), "nst_job_configure")
        self.levels = levels_num
        self.shape = (H0, W0)

    def release_job(self) -> None:
        """Give the job's pyramid workspace back (4.7 GB at L=2) and keep the context with its uploaded weights: what an
        engine waiting in neural_nets' pool holds is the smallest job nst_job_configure accepts."""
        self.configure(1, 16, 16)

    def level_shape(self, level: int) -> Tuple[int, int]:
        h, w = self.shape
        return h >> level, w >> level

    def set_targets(self, level: int, content: torch.Tensor, style: torch.Tensor) -> None:
        h, w = self.level_shape(level)
        content = content.contiguous().reshape(3, h, w)
        _chk_dev(content, self.device)
        style = style.contiguous().reshape(3, style.shape[-2], style.shape[-1])
        _chk_dev(style, self.device)
        _lib.check(self.ctx, self.lib.nst_level_set_targets(self.ctx, level, _ptr(content), _ptr(style),
                                                            style.shape[1], style.shape[2], _stream(self.device)),
                   "nst_level_set_targets")

    def closure(self, x: torch.Tensor, cw: float, sw: float, tvw: float,
                grad: Optional[torch.Tensor] = None, losses: Optional[torch.Tensor] = None):
        """Asynchronous on the current stream. Returns (grad (3,H,W), losses (4*levels+1,)) device tensors."""
        H, W = self.shape
        _chk_dev(x, self.device)
        if x.numel() != 3 * H * W:
            raise NstError("x has the wrong number of elements")
        if grad is None:
            grad = torch.empty((1, 3, H, W), dtype=torch.float32, device=self.device)
        if losses is None:
            losses = torch.empty(NST_LOSS_ROW * self.levels + 1, dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_closure(self.ctx, _ptr(x), cw, sw, tvw, _ptr(grad), _ptr(losses),
                                                  _stream(self.device)), "nst_closure")
        return grad, losses

    def closure_levels(self, x: torch.Tensor, cw: float, sw: float, tvw: float, mask: int,
                       grad: Optional[torch.Tensor] = None, losses: Optional[torch.Tensor] = None):
        """The closure restricted to the levels in `mask` (level sharding); see nst_closure_levels."""
        H, W = self.shape
        _chk_dev(x, self.device)
        if grad is None:
            grad = torch.empty((1, 3, H, W), dtype=torch.float32, device=self.device)
        if losses is None:
            losses = torch.empty(NST_LOSS_ROW * self.levels + 1, dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_closure_levels(self.ctx, _ptr(x), cw, sw, tvw, mask, _ptr(grad),
                                                         _ptr(losses), _stream(self.device)), "nst_closure_levels")
        return grad, losses

    # ---- stripe (window) closure: this engine evaluates a horizontal stripe of a larger image (sharding.StripePlan)
    def window_sums_count(self) -> int:
        n = C.c_size_t()
        _lib.check(None, self.lib.nst_window_sums_count(C.byref(n)), "nst_window_sums_count")
        return n.value

    def window_begin(self, xs: torch.Tensor, row0: int, rows: int, H0: int, sums: Optional[torch.Tensor] = None):
        """Forward pass of the stripe image xs (1,3,ext,W0); returns the un-normalised Gram / content / TV sums of the
        owned rows [row0, row0+rows) (see nst_window_begin)."""
        _chk_dev(xs, self.device)
        if sums is None:
            sums = torch.empty(self.window_sums_count(), dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_window_begin(self.ctx, _ptr(xs), row0, rows, H0, _ptr(sums), _stream(self.device)),
                   "nst_window_begin")
        return sums

    def window_end(self, xs: torch.Tensor, row0: int, rows: int, H0: int, cw: float, sw: float, tvw: float,
                   sums: torch.Tensor):
        """Backward pass for the loss terms of the owned rows, given the sums of ALL stripes; returns (d loss / d xs,
        level loss row (total, content, style, tv, total)) (see nst_window_end)."""
        _chk_dev(xs, self.device)
        _chk_dev(sums, self.device)
        gxs = torch.empty_like(xs)
        losses = torch.empty(NST_LOSS_ROW + 1, dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_window_end(self.ctx, _ptr(xs), row0, rows, H0, cw, sw, tvw, _ptr(sums), _ptr(gxs),
                                                     _ptr(losses), _stream(self.device)), "nst_window_end")
        return gxs, losses

    # ---- the optimisers' update arithmetic alone (unit parity) -------------------------------------------
    def adam_step(self, x: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, k: int, lr: float) -> None:
        """One torch.optim.Adam update in place (nst_adam_step): x, m, v from g at step count k with group lr."""
        for t in (x, g, m, v):
            _chk_dev(t, self.device)
            if t.numel() != x.numel():
                raise NstError("x, g, m, v must have the same number of elements")
        _lib.check(self.ctx, self.lib.nst_adam_step(self.ctx, _ptr(x), _ptr(g), _ptr(m), _ptr(v), x.numel(), k, float(lr),
                                                    _stream(self.device)), "nst_adam_step")

    def lbfgs_direction(self, g: torch.Tensor, ys: Sequence[torch.Tensor], ss: Sequence[torch.Tensor], ro: Sequence[float],
                        h_diag: float, form: int = 0) -> torch.Tensor:
        """d = -H g from the curvature pairs (nst_lbfgs_direction); form 0 = inner products, 1 = sequential recursion."""
        _chk_dev(g, self.device)
        m = len(ys)
        for t in list(ys) + list(ss):
            _chk_dev(t, self.device)
            if t.numel() != g.numel():
                raise NstError("history vectors must have g's size")
        yp = (C.c_void_p * max(m, 1))(*[t.data_ptr() for t in ys])
        sp = (C.c_void_p * max(m, 1))(*[t.data_ptr() for t in ss])
        rp = (C.c_float * max(m, 1))(*[float(r) for r in ro])
        d = torch.empty_like(g)
        _lib.check(self.ctx, self.lib.nst_lbfgs_direction(self.ctx, _ptr(g), yp, sp, rp, m, float(h_diag), g.numel(), form,
                                                          _ptr(d), _stream(self.device)), "nst_lbfgs_direction")
        return d

    def conv_mode(self) -> str:
        """How the 3x3 convolutions are evaluated (nst_options.conv_mode; env NST_CONV by default): 'f16x2' (default: two scaled
        fp16 pieces per fp32 operand, 3 MFMAs per product block, fp32 accumulate), 'bf16x3' (three exact bf16
        pieces, 6 MFMAs) or 'f32' (fp32 MFMA)."""
        return {0: "f32", 1: "bf16x3", 2: "f16x2"}[self.lib.nst_conv_mode(self.ctx)]

    def bytes(self) -> int:
        n = C.c_size_t()
        _lib.check(self.ctx, self.lib.nst_ctx_bytes(self.ctx, C.byref(n)), "nst_ctx_bytes")
        return n.value

    # ---- timing ---------------------------------------------------------------------------------
    def set_timing(self, mode: int) -> None:
        _lib.check(self.ctx, self.lib.nst_set_timing(self.ctx, mode), "nst_set_timing")

    def last_closure_ms(self) -> float:
        ms = C.c_float()
        _lib.check(self.ctx, self.lib.nst_last_closure_ms(self.ctx, C.byref(ms)), "nst_last_closure_ms")
        return ms.value

    def last_closure_class(self, cls: int):
        ms, n, fl = C.c_float(), C.c_int(), C.c_double()
        _lib.check(self.ctx, self.lib.nst_last_closure_class(self.ctx, cls, C.byref(ms), C.byref(n), C.byref(fl)),
                   "nst_last_closure_class")
        return ms.value, n.value, fl.value

    def timing_totals(self, cls: int, reset: bool = False):
        """(ms, launches, flops) accumulated since the last reset; cls -1 = whole closures."""
        ms, n, fl = C.c_double(), C.c_long(), C.c_double()
        _lib.check(self.ctx, self.lib.nst_timing_totals(self.ctx, cls, C.byref(ms), C.byref(n), C.byref(fl),
                                                        int(reset)), "nst_timing_totals")
        return ms.value, n.value, fl.value

    def timing_mfma_flops(self, cls: int) -> float:
        """Executed matrix-pipe FLOPs of the launches accumulated in timing_totals(cls) (read it BEFORE a resetting call)."""
        fl = C.c_double()
        _lib.check(self.ctx, self.lib.nst_timing_mfma_flops(self.ctx, cls, C.byref(fl)), "nst_timing_mfma_flops")
        return fl.value

    # ---- standalone pieces (unit parity) ----------------------------------------------------------
    def vgg_features(self, x: torch.Tensor) -> List[torch.Tensor]:
        x = x.reshape(3, x.shape[-2], x.shape[-1])
        _chk_dev(x, self.device)
        h, w = x.shape[1], x.shape[2]
        outs = [torch.empty((1, c, h >> s, w >> s), dtype=torch.float32, device=self.device)
                for c, s in zip(TAP_CHANNELS, TAP_SCALE)]
        arr = (C.c_void_p * 6)(*[o.data_ptr() for o in outs])
        _lib.check(self.ctx, self.lib.nst_vgg_features(self.ctx, _ptr(x), h, w, arr, _stream(self.device)),
                   "nst_vgg_features")
        return outs

    def vgg_features_backward(self, x: torch.Tensor, gouts: Sequence[Optional[torch.Tensor]]) -> torch.Tensor:
        x = x.reshape(3, x.shape[-2], x.shape[-1])
        _chk_dev(x, self.device)
        h, w = x.shape[1], x.shape[2]
        for g in gouts:
            if g is not None:
                _chk_dev(g, self.device)
        arr = (C.c_void_p * 6)(*[(g.data_ptr() if g is not None else 0) for g in gouts])
        gx = torch.empty((1, 3, h, w), dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_vgg_features_backward(self.ctx, _ptr(x), h, w, arr, _ptr(gx),
                                                                _stream(self.device)), "nst_vgg_features_backward")
        return gx

    LAYER_CHANNELS = (64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512)
    LAYER_SCALE = (0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4)

    def vgg_activations(self, x: torch.Tensor) -> List[torch.Tensor]:
        """All 13 post-ReLU conv outputs of a forward pass of x (nst_vgg_activations), each (1,C,h,w)."""
        x = x.reshape(3, x.shape[-2], x.shape[-1])
        _chk_dev(x, self.device)
        h, w = x.shape[1], x.shape[2]
        outs = [torch.empty((1, c, h >> sc, w >> sc), dtype=torch.float32, device=self.device)
                for c, sc in zip(self.LAYER_CHANNELS, self.LAYER_SCALE)]
        arr = (C.c_void_p * 13)(*[o.data_ptr() for o in outs])
        _lib.check(self.ctx, self.lib.nst_vgg_activations(self.ctx, _ptr(x), h, w, arr, _stream(self.device)),
                   "nst_vgg_activations")
        return outs

    def level_activation(self, level: int, layer: int) -> torch.Tensor:
        h, w = self.level_shape(level)
        c, sc = self.LAYER_CHANNELS[layer], self.LAYER_SCALE[layer]
        t = torch.empty((1, c, h >> sc, w >> sc), dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_level_activation(self.ctx, level, layer, _ptr(t), _stream(self.device)),
                   "nst_level_activation")
        return t

    def level_image(self, level: int) -> torch.Tensor:
        """The (1,3,h,w) image of pyramid level `level` >= 1 that the last closure evaluated (nst_level_image)."""
        h, w = self.level_shape(level)
        t = torch.empty((1, 3, h, w), dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_level_image(self.ctx, level, _ptr(t), _stream(self.device)), "nst_level_image")
        return t

    def level_activations(self, level: int) -> List[torch.Tensor]:
        """The 13 post-ReLU conv outputs the last closure left in the workspace of `level` (nst_level_activation),
        each (1,C,h,w): what the parity tests derive the device pass's ReLU / pooling decisions from."""
        h, w = self.level_shape(level)
        outs = []
        for l, (c, sc) in enumerate(zip(self.LAYER_CHANNELS, self.LAYER_SCALE)):
            t = torch.empty((1, c, h >> sc, w >> sc), dtype=torch.float32, device=self.device)
            _lib.check(self.ctx, self.lib.nst_level_activation(self.ctx, level, l, _ptr(t), _stream(self.device)),
                       "nst_level_activation")
            outs.append(t)
        return outs

    def gram(self, f: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        _chk_dev(f, self.device)
        b, c, h, w = f.shape
        assert b == 1
        g = torch.empty((1, c, c), dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_gram(self.ctx, _ptr(f), c, h, w, int(normalize), _ptr(g),
                                               _stream(self.device)), "nst_gram")
        return g

    def total_variation(self, y: torch.Tensor, want_grad: bool = False):
        _chk_dev(y, self.device)
        b, c, h, w = y.shape
        val = torch.empty(1, dtype=torch.float32, device=self.device)
        grad = torch.empty_like(y) if want_grad else None
        _lib.check(self.ctx, self.lib.nst_total_variation(self.ctx, _ptr(y), b * c, h, w, _ptr(val), _ptr(grad),
                                                          _stream(self.device)), "nst_total_variation")
        return (val, grad) if want_grad else val

    def bicubic_half(self, x: torch.Tensor) -> torch.Tensor:
        _chk_dev(x, self.device)
        b, c, h, w = x.shape
        y = torch.empty((b, c, h // 2, w // 2), dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_bicubic_half(self.ctx, _ptr(x), b * c, h, w, _ptr(y), _stream(self.device)),
                   "nst_bicubic_half")
        return y

    def bicubic_half_backward(self, gy: torch.Tensor, h: int, w: int) -> torch.Tensor:
        _chk_dev(gy, self.device)
        b, c = gy.shape[0], gy.shape[1]
        gx = torch.empty((b, c, h, w), dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_bicubic_half_backward(self.ctx, _ptr(gy), b * c, h, w, _ptr(gx),
                                                                _stream(self.device)), "nst_bicubic_half_backward")
        return gx

    def prepare_img(self, hwc: torch.Tensor) -> torch.Tensor:
        _chk_dev(hwc, self.device)
        h, w, _ = hwc.shape
        out = torch.empty((1, 3, h, w), dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_prepare_img(self.ctx, _ptr(hwc), h, w, _ptr(out), _stream(self.device)),
                   "nst_prepare_img")
        return out

    def unprepare_img(self, chw: torch.Tensor) -> torch.Tensor:
        _chk_dev(chw, self.device)
        h, w = chw.shape[-2], chw.shape[-1]
        out = torch.empty((h, w, 3), dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_unprepare_img(self.ctx, _ptr(chw), h, w, _ptr(out), _stream(self.device)),
                   "nst_unprepare_img")
        return out


    # ---- job set-up on the device -------------------------------------
    def resize(self, img: torch.Tensor, nh: int, nw: int) -> torch.Tensor:
        """cv2.INTER_CUBIC resize of an (h,w,c) float32 device image."""
        _chk_dev(img, self.device)
        h, w, c = img.shape
        out = torch.empty((nh, nw, c), dtype=torch.float32, device=self.device)
        _lib.check(self.ctx, self.lib.nst_resize_bicubic(self.ctx, _ptr(img), h, w, c, _ptr(out), nh, nw,
                                                         _stream(self.device)), "nst_resize_bicubic")
        return out

    def gather_rows(self, src: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
        _chk_dev(src, self.device)
        assert perm.dtype == torch.int64 and perm.is_cuda and perm.is_contiguous()
        out = torch.empty_like(src)
        _lib.check(self.ctx, self.lib.nst_gather_rows(self.ctx, _ptr(src), _ptr(perm), perm.numel(), src.shape[-1],
                                                      _ptr(out), _stream(self.device)), "nst_gather_rows")
        return out

    def gaussian_mask_accumulate(self, acc: torch.Tensor, src: Optional[torch.Tensor], central: float,
                                 peripheral: float, dispersion: float) -> None:
        _chk_dev(acc, self.device)
        h, w, c = acc.shape
        if src is not None:
            _chk_dev(src, self.device, acc.shape)
        _lib.check(self.ctx, self.lib.nst_gaussian_mask_accumulate(self.ctx, _ptr(acc), _ptr(src), h, w, c, central,
                                                                   peripheral, dispersion, _stream(self.device)),
                   "nst_gaussian_mask_accumulate")

    def noise_blend(self, content: torch.Tensor, noise: torch.Tensor, noise_factor: float) -> torch.Tensor:
        _chk_dev(content, self.device)
        _chk_dev(noise, self.device, content.shape)
        h, w, c = content.shape
        out = torch.empty_like(content)
        _lib.check(self.ctx, self.lib.nst_noise_blend(self.ctx, _ptr(content), _ptr(noise), h, w, c, noise_factor,
                                                      _ptr(out), _stream(self.device)), "nst_noise_blend")
        return out

    def scale(self, src: torch.Tensor, alpha: float) -> torch.Tensor:
        _chk_dev(src, self.device)
        out = torch.empty_like(src)
        _lib.check(self.ctx, self.lib.nst_scale(self.ctx, _ptr(src), alpha, src.numel(), _ptr(out),
                                                _stream(self.device)), "nst_scale")
        return out


class PixelOptimizer:
    """nst_opt: torch.optim.Adam / LBFGS as the reference constructs them, driving the closure."""

    def __init__(self, engine: StyleEngine, name: str, lr_start: float = 10.0, lbfgs_max_eval: int = 1):
        if name == "adam":
            kind = _lib.NST_OPT_ADAM
        elif name == "lbfgs":
            kind = _lib.NST_OPT_LBFGS
        else:
            raise RuntimeError("Unknown optimizer")   # neural_style_transfer.py:137-138
        self.engine = engine
        self.name = name
        h = C.c_void_p()
        _lib.check(engine.ctx, engine.lib.nst_opt_create(engine.ctx, kind, lr_start, lbfgs_max_eval, C.byref(h)),
                   "nst_opt_create")
        self.h = h
        self.row = NST_LOSS_ROW * engine.levels + 1
        self.cap = 32 if name == "lbfgs" else 1
        self._rows = np.zeros((self.cap, self.row), dtype=np.float32)

    def shard_levels_comm(self, comm: "Communicator") -> None:
        """Level sharding with the collective behind the C ABI (nst_opt_shard_levels_comm): one ncclAllReduce of the packed
        gradient + loss row per closure on the job's stream, no Python in the loop."""
        from . import sharding
        e = self.engine
        self._comm = comm
        _lib.check(e.ctx, e.lib.nst_opt_shard_levels_comm(self.h, sharding.level_mask(e.levels, comm.rank, comm.world),
                                                          comm.h), "nst_opt_shard_levels_comm")

    def history(self):
        """(curvature pairs held, optimiser iteration count)."""
        p, n = C.c_int(), C.c_int()
        _lib.check(self.engine.ctx, self.engine.lib.nst_opt_history(self.h, C.byref(p), C.byref(n)), "nst_opt_history")
        return p.value, n.value

    def shard_levels(self, rank: int, world: int, dist_mod=None, group=None) -> None:
        """Level sharding (BASELINE config 4): this rank evaluates only its levels; after every closure
        the partial gradient and loss rows are all-reduced (RCCL) before the driver reads them."""
        from . import sharding
        e = self.engine
        H, W = e.shape
        self._g = torch.zeros((1, 3, H, W), dtype=torch.float32, device=e.device)
        self._l = torch.zeros(self.row, dtype=torch.float32, device=e.device)

        def hook(_user):
            sharding.allreduce_closure(self._g, self._l, dist_mod, group)

        self._hook = _lib.REDUCE_HOOK(hook)          # keep the callback object alive
        _lib.check(e.ctx, e.lib.nst_opt_shard_levels(self.h, sharding.level_mask(e.levels, rank, world),
                                                     _ptr(self._g), _ptr(self._l), self._hook, None),
                   "nst_opt_shard_levels")

    def shard_stripes(self, rank: int, world: int, weights, content_t, style_t,
                      dist_mod=None, group=None, comm: "Communicator" = None) -> None:
        """Spatial sharding of the large levels (SURVEY 8(e) partition B, halo recompute) on top of level sharding of the
        rest: every rank evaluates a horizontal stripe (its rows + a 96-row halo, a single-level engine of its own) of
        every STRIPED level and its share of the other levels.  content_t / style_t: the prepared (1,3,h,w) content and
        (1,3,hs,ws) style images of the striped levels, level 0 first - a tensor (level 0 only, 75 % of the work) or a
        list (levels 0, 1, ...: with level 1 striped as well 94 % of the work is cut evenly).  Per closure: ONE
        all-reduce of the Gram / content / TV sums of all striped levels between the stripes' forward and backward passes,
        one of the pixel gradient and the loss rows at the end.  A striped level l >= 1 works on the rows of
        x_l = D^l x (the whole down-sampled image is formed on every rank: HBM-bound, 1/4 of the pixels) and hands its
        partial gradient back through the transpose of the down-sampling, which is linear - the final all-reduce sums the
        ranks' parts.
        `comm` (a Communicator): both collectives go through the C ABI's RCCL communicator (nst_comm_allreduce_sum on
        the job's stream; gradient and loss row are ONE packed buffer, as in nst_opt_shard_levels_comm) and rank / world
        are the communicator's; otherwise through `dist_mod` (torch.distributed: gloo rehearsals, or nccl)."""
        from . import sharding
        if comm is not None:
            rank, world = comm.rank, comm.world
        elif dist_mod is None:
            import torch.distributed as dist_mod
        contents = list(content_t) if isinstance(content_t, (list, tuple)) else [content_t]
        styles = list(style_t) if isinstance(style_t, (list, tuple)) else [style_t]
        e = self.engine
        H, W = e.shape
        nstriped = min(len(contents), len(styles), e.levels)
        plans, stripes = [], []
        for l in range(nstriped):
            plan = sharding.StripePlan(H >> l, world, rank)
            st = StyleEngine(weights, e.device)
            st.configure(1, plan.ext_rows, W >> l)
            st.set_targets(0, plan.cut(contents[l]), styles[l].contiguous())
            plans.append(plan)
            stripes.append(st)
        self._stripes, self._plans = stripes, plans
        self._stripe, self._plan = stripes[0], plans[0]
        n = 3 * H * W
        self._pack = torch.zeros(n + self.row, dtype=torch.float32, device=e.device)
        self._g = self._pack[:n].view(1, 3, H, W)
        self._l = self._pack[n:]
        count = stripes[0].window_sums_count()
        sums_all = torch.empty(nstriped * count, dtype=torch.float32, device=e.device)
        # the other levels: dealt largest first onto the least-loaded rank (every rank carries an equal stripe of the
        # striped levels); a striped level is nobody's in the level mask (its stripes are added here)
        mask = 0
        for l in sharding.deal_levels(range(nstriped, e.levels), world)[rank]:
            mask |= 1 << l

        def hook(_user):
            x, (cw, sw, tvw) = self._x, self._w
            imgs, cuts = [x], []
            for l in range(nstriped):
                if l > 0:
                    imgs.append(e.bicubic_half(imgs[-1]))               # x_l = D x_{l-1}, whole image
                xs = plans[l].cut(imgs[l])
                cuts.append(xs)
                stripes[l].window_begin(xs, plans[l].row0, plans[l].rows, H >> l, sums_all[l * count:(l + 1) * count])
            if comm is not None:
                comm.allreduce_sum(sums_all)
            else:
                dist_mod.all_reduce(sums_all, op=dist_mod.ReduceOp.SUM, group=group)
            for l in range(nstriped):
                gxs, row = stripes[l].window_end(cuts[l], plans[l].row0, plans[l].rows, H >> l, cw, sw, tvw,
                                                 sums_all[l * count:(l + 1) * count])
                if l == 0:
                    plans[0].add_into(self._g, gxs)
                else:
                    gl = torch.zeros_like(imgs[l])
                    plans[l].add_into(gl, gxs)
                    for k in range(l, 0, -1):                           # back through the down-sampling chain: D^T
                        gl = e.bicubic_half_backward(gl, H >> (k - 1), W >> (k - 1))
                    self._g += gl
                if rank == 0:                  # every rank holds the same row of a striped level: one contributor
                    self._l[4 * l:4 * l + 4] = row[0:4]
            if comm is not None:
                comm.allreduce_sum(self._pack)
                sharding.reform_total(self._l)
            else:
                sharding.allreduce_closure(self._g, self._l, dist_mod, group)

        self._hook = _lib.REDUCE_HOOK(hook)
        _lib.check(e.ctx, e.lib.nst_opt_shard_levels(self.h, mask, _ptr(self._g), _ptr(self._l), self._hook, None),
                   "nst_opt_shard_levels")

    def step(self, x: torch.Tensor, cw: float, sw: float, tvw: float, want_losses: bool = True):
        """One optimizer.step(closure). Returns (StepInfo, rows[closures, 4*levels+1] or None)."""
        e = self.engine
        _chk_dev(x, e.device)
        self._x, self._w = x, (cw, sw, tvw)          # what a stripe hook evaluates (x is updated in place)
        info = StepInfo()
        ptr = C.c_void_p(self._rows.ctypes.data) if want_losses else C.c_void_p(0)
        _lib.check(e.ctx, e.lib.nst_opt_step(self.h, _ptr(x), cw, sw, tvw, ptr, self.cap, C.byref(info),
                                             _stream(e.device)), "nst_opt_step")
        rows = self._rows[:min(info.closures, self.cap)].copy() if want_losses else None
        return info, rows

    def close(self):
        if getattr(self, "h", None) and getattr(self.engine, "ctx", None):
            self.engine.lib.nst_opt_destroy(self.h)
        self.h = None
        for st in getattr(self, "_stripes", ()):       # the stripe engines of shard_stripes
            st.close()
        self._stripes = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Communicator:
    """nst_comm: an RCCL communicator behind the C ABI (one rank per GPU).  `id_bytes`: the NST_COMM_ID_BYTES of
    `Communicator.unique_id()` made on rank 0 and handed to every rank (e.g. with torch.distributed.broadcast_object_list
    over gloo, or a file)."""

    def __init__(self, device: int, rank: int, world: int, id_bytes: bytes):
        self.lib = _lib.load()
        if len(id_bytes) != _lib.NST_COMM_ID_BYTES:
            raise NstError("communicator id must be NST_COMM_ID_BYTES long")
        buf = C.create_string_buffer(bytes(id_bytes), _lib.NST_COMM_ID_BYTES)
        h = C.c_void_p()
        _lib.check(None, self.lib.nst_comm_create(int(device), rank, world, buf, C.byref(h)), "nst_comm_create")
        self.h, self.rank, self.world, self.device = h, rank, world, torch.device("cuda", int(device))

    @staticmethod
    def unique_id() -> bytes:
        lib = _lib.load()
        buf = C.create_string_buffer(_lib.NST_COMM_ID_BYTES)
        _lib.check(None, lib.nst_comm_unique_id(buf), "nst_comm_unique_id")
        return buf.raw

    def allreduce_sum(self, t: torch.Tensor) -> None:
        _chk_dev(t, self.device)
        _lib.check(None, self.lib.nst_comm_allreduce_sum(self.h, _ptr(t), t.numel(), _stream(self.device)),
                   "nst_comm_allreduce_sum")

    def info(self):
        """(rank, world, all-reduce calls so far, bytes carried)."""
        r, w, n, b = C.c_int(), C.c_int(), C.c_long(), C.c_double()
        _lib.check(None, self.lib.nst_comm_info(self.h, C.byref(r), C.byref(w), C.byref(n), C.byref(b)), "nst_comm_info")
        return r.value, w.value, n.value, b.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.nst_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
