"""CPU oracle for the pyramid style-transfer hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the algorithm the
reference executes inside ``optimizer.step(closure)``.  Nothing in the shipped
package imports it: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may, and there only as the checker or as
the timed CPU baseline - never as the thing measured or shipped.

Parity pin: the functions below are checked against outputs of the reference
itself (its own ``math_utils``, ``neural_nets.Vgg19``, ``LossBuilder`` and
``NeuralStyleTransfer.process`` executed on CPU torch in the build container)
stored as fixtures under ``tests/golden/`` by ``tests/golden/make_fixtures.py``.
The reference has no tests or golden vectors of its own; at the torchvision
(VGG19 topology/weights) and cv2 boundaries parity is therefore *unpinned*:
those packages are absent offline and the topology is restated from
torchvision's published cfg "E".

Each function cites the reference file:line (paths relative to the reference
checkout; ``torch:`` = the installed torch package) it follows.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# neural_style_transfer.py:22-23
IMAGENET_MEAN_255 = (123.675, 116.28, 103.53)

# torchvision VGG cfg "E", features[0:30] (neural_nets.py:19, :31-48)
VGG19_CONVS = (
    ("conv1_1", 3, 64), ("conv1_2", 64, 64),
    ("conv2_1", 64, 128), ("conv2_2", 128, 128),
    ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv3_4", 256, 256),
    ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), ("conv4_4", 512, 512),
    ("conv5_1", 512, 512),
)
# a 2x2/2 max-pool follows these convs (after their ReLU)
POOL_AFTER = ("conv1_2", "conv2_2", "conv3_4", "conv4_4")
# outputs the reference exposes (neural_nets.py:22, :53-68). "conv4_2" is really
# ReLU(conv4_2): slice6 opens with an in-place ReLU that rewrites the storage
# slice5 returned.
TAPS = ("conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv4_2", "conv5_1")
CONTENT_INDEX = 4                 # neural_nets.py:25
STYLE_INDICES = (0, 1, 2, 3, 5)   # neural_nets.py:27-28


# --------------------------------------------------------------------------
# synthetic, seeded inputs; mirrored by artstyletransfer_amd.synthetic
# --------------------------------------------------------------------------
# Standard deviation of the seeded conv biases of the weight set the parity tests and the golden fixtures use.
# SURVEY 8(d) defines the BENCH weights with b = 0; pretrained VGG19 biases are not zero, so every parity test runs
# with biases that matter (pre-activations under these weights have a standard deviation of 9..22).
TEST_BIAS_STD = 2.0


def synthetic_vgg19_weights(seed: int = 1234, bias_std: float = 0.0) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """Kaiming fan-out weights; biases b = bias_std * randn from a generator of their own (seed + 1), so
    the weight values do not depend on bias_std."""
    g = torch.Generator().manual_seed(seed)
    gb = torch.Generator().manual_seed(seed + 1)
    out = []
    for _, cin, cout in VGG19_CONVS:
        w = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cout * 9))
        b = torch.randn(cout, generator=gb) * bias_std if bias_std else torch.zeros(cout)
        out.append((w, b))
    return out


def synthetic_image(h: int, w: int, seed: int) -> np.ndarray:
    """(h, w, 3) float32 RGB in [0, 1]: low-res uniform noise, bicubic-upsampled, clipped."""
    lo = np.random.RandomState(seed).rand(max(h // 16, 1), max(w // 16, 1), 3).astype(np.float32)
    t = torch.from_numpy(lo).permute(2, 0, 1).unsqueeze(0)
    up = F.interpolate(t, size=(h, w), mode="bicubic", align_corners=False)
    return up.squeeze(0).permute(1, 2, 0).clamp(0.0, 1.0).contiguous().numpy()


# --------------------------------------------------------------------------
# image <-> tensor (neural_style_transfer.py:375-393)
# --------------------------------------------------------------------------
def prepare_img(img: np.ndarray) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).permute(2, 0, 1)
    t = t.mul(255.0)
    mean = torch.tensor(IMAGENET_MEAN_255, dtype=torch.float32).view(3, 1, 1)
    return (t - mean).unsqueeze(0)


def unprepare_img(t: torch.Tensor) -> np.ndarray:
    a = t.detach().permute(0, 2, 3, 1).squeeze(0).cpu().numpy().copy()
    a += np.array(IMAGENET_MEAN_255).reshape(1, 1, 3)
    return a.astype(np.float32) / 255


# --------------------------------------------------------------------------
# math_utils.py:26-41
# --------------------------------------------------------------------------
def gram_matrix(x: torch.Tensor, should_normalize: bool = True) -> torch.Tensor:
    b, ch, h, w = x.shape
    f = x.reshape(b, ch, h * w)
    g = f.bmm(f.transpose(1, 2))
    if should_normalize:
        g = g / (ch * h * w)
    return g


def total_variation(y: torch.Tensor, signs=None) -> torch.Tensor:
    """math_utils.py:37-41.  `signs` = (sx, sy): evaluate |d| as s * d with the signs ANOTHER evaluation took (see
    Decisions.tv): on flat image regions the neighbour differences of a down-sampled level are rounding noise whose
    sign is arbitrary, and abs' sub-gradient sign(d) turns that into a gradient difference of order one."""
    dx = y[:, :, :, :-1] - y[:, :, :, 1:]
    dy = y[:, :, :-1, :] - y[:, :, 1:, :]
    if signs is None:
        mx, my = torch.mean(torch.abs(dx)), torch.mean(torch.abs(dy))
    else:
        mx, my = torch.mean(signs[0] * dx), torch.mean(signs[1] * dy)
    return mx * mx + my * my


# --------------------------------------------------------------------------
# neural_nets.py:53-68 (forward), topology from torchvision cfg "E"
# --------------------------------------------------------------------------
class Decisions:
    """The ReLU and max-pool DECISIONS of another evaluation of the same network - the device pass - read off its 13
    post-ReLU activation maps: a unit is on where its value is > 0; a pooling window hands its gradient to its first
    maximum.  Passed to vgg19_features / closure_eval they replace this evaluation's own decisions, so that two fp32
    evaluations whose pre-activations differ in the last bits are compared under EQUAL decisions: a pre-activation
    within rounding of 0 otherwise lands on different sides and changes the gradient over that unit's whole receptive
    field.  (The tests also check that the two evaluations' own decisions differ only at such near-ties.)"""

    def __init__(self, activations: Sequence[torch.Tensor], level_image: Optional[torch.Tensor] = None):
        assert len(activations) == len(VGG19_CONVS)
        # the signs the other pass's total-variation term took (None: this evaluation's own)
        self.tv = None
        if level_image is not None:
            y = level_image
            self.tv = (torch.sign(y[:, :, :, :-1] - y[:, :, :, 1:]), torch.sign(y[:, :, :-1, :] - y[:, :, 1:, :]))
        self.relu = [a > 0 for a in activations]
        self.pool = {}
        for (name, _, _), a in zip(VGG19_CONVS, activations):
            if name in POOL_AFTER:
                self.pool[name] = F.max_pool2d(a, kernel_size=2, stride=2, return_indices=True)[1]


def vgg19_features(x: torch.Tensor, weights: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                   decisions: Optional[Decisions] = None, record: Optional[list] = None):
    """Returns the 6 maps (relu1_1, relu2_1, relu3_1, relu4_1, ReLU(conv4_2), relu5_1).  `decisions`: see Decisions
    (None = this evaluation's own: the reference's forward).  `record` (a list) receives the 13 pre-activations."""
    outs = []
    for li, ((name, _, _), (w, b)) in enumerate(zip(VGG19_CONVS, weights)):
        pre = F.conv2d(x, w, b, stride=1, padding=1)
        if record is not None:
            record.append(pre.detach())
        x = F.relu(pre) if decisions is None else pre * decisions.relu[li].to(pre.dtype)
        if name in TAPS:
            outs.append(x)
        if name in POOL_AFTER:
            if decisions is None:
                x = F.max_pool2d(x, kernel_size=2, stride=2)
            else:
                idx = decisions.pool[name]
                x = x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
    return outs


# --------------------------------------------------------------------------
# pyramid down-sample of the optimised image (neural_style_transfer.py:173-176)
# --------------------------------------------------------------------------
def bicubic_half(x: torch.Tensor) -> torch.Tensor:
    h, w = x.shape[2], x.shape[3]
    return F.interpolate(x, size=(h // 2, w // 2), mode="bicubic")


_HALF_TAPS = (-0.09375, 0.59375, 0.59375, -0.09375)


def bicubic_half_fixed(x: torch.Tensor) -> torch.Tensor:
    """Closed form of bicubic_half for even sizes: separable 4-tap stride-2 filter on
    rows/cols 2d-1..2d+2 with clamped indices (torch:include/ATen/native/UpSample.h:289-312,
    :373-423 with A=-0.75, t=0.5).  Used only to cross-check the kernel's tap table."""
    w1 = torch.tensor(_HALF_TAPS, dtype=x.dtype)
    k = torch.outer(w1, w1).view(1, 1, 4, 4).repeat(x.shape[1], 1, 1, 1)
    xp = F.pad(x, (1, 1, 1, 1), mode="replicate")
    return F.conv2d(xp, k, stride=2, groups=x.shape[1])


# --------------------------------------------------------------------------
# LossBuilder (neural_style_transfer.py:66-112)
# --------------------------------------------------------------------------
class LevelTargets:
    """Target representations of one pyramid level (neural_style_transfer.py:78-82)."""

    def __init__(self, content_img_t: torch.Tensor, style_img_t: torch.Tensor, weights):
        with torch.no_grad():
            self.content = vgg19_features(content_img_t, weights)[CONTENT_INDEX].squeeze(0)
            sf = vgg19_features(style_img_t, weights)
            self.grams = [gram_matrix(sf[i]) for i in STYLE_INDICES]


def level_loss(x: torch.Tensor, tg: LevelTargets, weights, cw: float, sw: float, tvw: float,
               decisions: Optional[Decisions] = None, record: Optional[list] = None):
    """(total, content, style, tv) of one level (neural_style_transfer.py:84-112).
    The reference's per-closure ``0 * randn`` noise term (:91-93) contributes exactly 0."""
    feats = vgg19_features(x, weights, decisions, record)
    content = F.mse_loss(tg.content, feats[CONTENT_INDEX].squeeze(0), reduction="mean")
    style = 0.0
    for g_gt, i in zip(tg.grams, STYLE_INDICES):
        style = style + F.mse_loss(g_gt[0], gram_matrix(feats[i])[0], reduction="mean")
    style = style / len(tg.grams)
    tv = total_variation(x, decisions.tv if decisions is not None else None)
    total = cw * content + sw * style + tvw * tv
    return total, content, style, tv


def closure_eval(x: torch.Tensor, targets: Sequence[LevelTargets], weights,
                 cw: float, sw: float, tvw: float, decisions: Optional[Sequence[Decisions]] = None,
                 record: Optional[list] = None):
    """One closure evaluation (neural_style_transfer.py:152-199, without the LR decay and
    prints): returns (total_loss float32 tensor, grad (1,3,H,W), per-level rows
    [(total, content, style, tv), ...]).  `decisions` (one Decisions per level): evaluate under another pass's
    ReLU / pooling decisions; `record` receives one list of 13 pre-activations per level."""
    x = x.detach().clone().requires_grad_(True)
    levels, total, rows = [x], None, []
    for i, tg in enumerate(targets):
        if i > 0:      # same op order as the reference: autograd's accumulation order follows it
            levels.append(bicubic_half(levels[i - 1]))
        rec = None
        if record is not None:
            rec = []
            record.append(rec)
        t, c, s, tv = level_loss(levels[i], tg, weights, cw, sw, tvw, decisions[i] if decisions is not None else None, rec)
        total = t if total is None else 1.0 * total + t
        rows.append((float(t.detach()), float(c.detach()), float(s.detach()), float(tv.detach())))
    total.backward()
    return total.detach(), x.grad.detach(), rows


# --------------------------------------------------------------------------
# optimisers, restated so that they can drive any closure (torch's or the HIP one)
# closure signature: f(x_flat_tensor) -> (loss: float, grad_flat_tensor)
# --------------------------------------------------------------------------
class AdamState:
    """torch:optim/adam.py:457-546 (_single_tensor_adam, amsgrad=False, weight_decay=0,
    maximize=False, capturable=False), one tensor."""

    def __init__(self, n: int, beta1=0.9, beta2=0.999, eps=1e-8):
        self.m = torch.zeros(n)
        self.v = torch.zeros(n)
        self.k = 0
        self.b1, self.b2, self.eps = beta1, beta2, eps

    def update(self, x: torch.Tensor, g: torch.Tensor, lr: float) -> None:
        self.k += 1
        self.m.lerp_(g, 1 - self.b1)
        self.v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
        bc1 = 1 - self.b1 ** self.k
        bc2 = 1 - self.b2 ** self.k
        step_size = lr / bc1
        denom = (self.v.sqrt() / math.sqrt(bc2)).add_(self.eps)
        x.addcdiv_(self.m, denom, value=-step_size)


class LbfgsState:
    """torch:optim/lbfgs.py:332-537 with the reference's constructor arguments
    (neural_style_transfer.py:136: max_iter=1, line_search_fn='strong_wolfe'; defaults
    history_size=100, tolerance_grad=1e-7, tolerance_change=1e-9).  ``max_eval`` is a
    parameter: torch's default ``max_iter*5//4 = 1`` gives ``max_ls = 0`` in torch 2.10
    (one trial point, kept only if the loss dropped); ``max_eval=26``
    reproduces the legacy full line search."""

    def __init__(self, max_eval: int = 1, history_size: int = 100,
                 tolerance_grad: float = 1e-7, tolerance_change: float = 1e-9):
        self.max_eval = max_eval
        self.history_size = history_size
        self.tol_g = tolerance_grad
        self.tol_c = tolerance_change
        self.n_iter = 0
        self.func_evals = 0
        self.d = None
        self.t = None
        self.old_dirs: List[torch.Tensor] = []
        self.old_stps: List[torch.Tensor] = []
        self.ro: List[torch.Tensor] = []
        self.H_diag = 1
        self.prev_flat_grad = None
        self.prev_loss = None
        self.last_accept: Optional[bool] = None   # diagnostics for parity tests


def _cubic_interpolate(x1, f1, g1, x2, f2, g2, bounds=None):
    # torch:optim/lbfgs.py:12-37
    if bounds is not None:
        xmin_bound, xmax_bound = bounds
    else:
        xmin_bound, xmax_bound = (x1, x2) if x1 <= x2 else (x2, x1)
    d1 = g1 + g2 - 3 * (f1 - f2) / (x1 - x2)
    d2_square = d1 ** 2 - g1 * g2
    if d2_square >= 0:
        d2 = d2_square.sqrt() if isinstance(d2_square, torch.Tensor) else math.sqrt(d2_square)
        if x1 <= x2:
            min_pos = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2 * d2))
        else:
            min_pos = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2 * d2))
        return min(max(min_pos, xmin_bound), xmax_bound)
    return (xmin_bound + xmax_bound) / 2.0


def _strong_wolfe(obj_func, t, d, f, g, gtd, c1=1e-4, c2=0.9, tolerance_change=1e-9, max_ls=25):
    # torch:optim/lbfgs.py:40-209
    d_norm = d.abs().max()
    g = g.clone()
    f_new, g_new = obj_func(t)
    ls_func_evals = 1
    gtd_new = g_new.dot(d)
    t_prev, f_prev, g_prev, gtd_prev = 0, f, g, gtd
    done = False
    ls_iter = 0
    while ls_iter < max_ls:
        if f_new > (f + c1 * t * gtd) or (ls_iter > 1 and f_new >= f_prev):
            bracket = [t_prev, t]
            bracket_f = [f_prev, f_new]
            bracket_g = [g_prev, g_new.clone()]
            bracket_gtd = [gtd_prev, gtd_new]
            break
        if abs(gtd_new) <= -c2 * gtd:
            bracket = [t]
            bracket_f = [f_new]
            bracket_g = [g_new]
            done = True
            break
        if gtd_new >= 0:
            bracket = [t_prev, t]
            bracket_f = [f_prev, f_new]
            bracket_g = [g_prev, g_new.clone()]
            bracket_gtd = [gtd_prev, gtd_new]
            break
        min_step = t + 0.01 * (t - t_prev)
        max_step = t * 10
        tmp = t
        t = _cubic_interpolate(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, bounds=(min_step, max_step))
        t_prev = tmp
        f_prev = f_new
        g_prev = g_new.clone()
        gtd_prev = gtd_new
        f_new, g_new = obj_func(t)
        ls_func_evals += 1
        gtd_new = g_new.dot(d)
        ls_iter += 1
    if ls_iter == max_ls:
        bracket = [0, t]
        bracket_f = [f, f_new]
        bracket_g = [g, g_new]
        bracket_gtd = [gtd, gtd_new]   # unused when max_ls == 0 (zoom loop is skipped)
    insuf_progress = False
    low_pos, high_pos = (0, 1) if bracket_f[0] <= bracket_f[-1] else (1, 0)
    while not done and ls_iter < max_ls:
        if abs(bracket[1] - bracket[0]) * d_norm < tolerance_change:
            break
        t = _cubic_interpolate(bracket[0], bracket_f[0], bracket_gtd[0],
                               bracket[1], bracket_f[1], bracket_gtd[1])
        eps = 0.1 * (max(bracket) - min(bracket))
        if min(max(bracket) - t, t - min(bracket)) < eps:
            if insuf_progress or t >= max(bracket) or t <= min(bracket):
                if abs(t - max(bracket)) < abs(t - min(bracket)):
                    t = max(bracket) - eps
                else:
                    t = min(bracket) + eps
                insuf_progress = False
            else:
                insuf_progress = True
        else:
            insuf_progress = False
        f_new, g_new = obj_func(t)
        ls_func_evals += 1
        gtd_new = g_new.dot(d)
        ls_iter += 1
        if f_new > (f + c1 * t * gtd) or f_new >= bracket_f[low_pos]:
            bracket[high_pos] = t
            bracket_f[high_pos] = f_new
            bracket_g[high_pos] = g_new.clone()
            bracket_gtd[high_pos] = gtd_new
            low_pos, high_pos = (0, 1) if bracket_f[0] <= bracket_f[1] else (1, 0)
        else:
            if abs(gtd_new) <= -c2 * gtd:
                done = True
            elif gtd_new * (bracket[high_pos] - bracket[low_pos]) >= 0:
                bracket[high_pos] = bracket[low_pos]
                bracket_f[high_pos] = bracket_f[low_pos]
                bracket_g[high_pos] = bracket_g[low_pos]
                bracket_gtd[high_pos] = bracket_gtd[low_pos]
            bracket[low_pos] = t
            bracket_f[low_pos] = f_new
            bracket_g[low_pos] = g_new.clone()
            bracket_gtd[low_pos] = gtd_new
    t = bracket[low_pos]
    return bracket_f[low_pos], bracket_g[low_pos], t, ls_func_evals


def lbfgs_step(st: LbfgsState, x: torch.Tensor, lr: float,
               closure: Callable[[torch.Tensor], Tuple[float, torch.Tensor]]) -> float:
    """One ``LBFGS.step(closure)`` with max_iter=1 (torch:optim/lbfgs.py:332-537).
    ``lr`` is the group's lr read BEFORE the first closure call decays it (:349).
    ``x`` (flat fp32) is updated in place; returns the loss of the first closure call."""
    loss, flat_grad = closure(x)
    orig_loss = loss
    current_evals = 1
    st.func_evals += 1
    if flat_grad.abs().max() <= st.tol_g:
        return orig_loss
    st.n_iter += 1
    if st.n_iter == 1:
        d = flat_grad.neg()
        st.old_dirs, st.old_stps, st.ro = [], [], []
        st.H_diag = 1
    else:
        y = flat_grad.sub(st.prev_flat_grad)
        s = st.d.mul(st.t)
        ys = y.dot(s)
        if ys > 1e-10:
            if len(st.old_dirs) == st.history_size:
                st.old_dirs.pop(0)
                st.old_stps.pop(0)
                st.ro.pop(0)
            st.old_dirs.append(y)
            st.old_stps.append(s)
            st.ro.append(1.0 / ys)
            st.H_diag = ys / y.dot(y)
        num_old = len(st.old_dirs)
        al = [None] * num_old
        q = flat_grad.neg()
        for i in range(num_old - 1, -1, -1):
            al[i] = st.old_stps[i].dot(q) * st.ro[i]
            q.add_(st.old_dirs[i], alpha=-al[i])
        d = r = torch.mul(q, st.H_diag)
        for i in range(num_old):
            be_i = st.old_dirs[i].dot(r) * st.ro[i]
            r.add_(st.old_stps[i], alpha=al[i] - be_i)
    if st.prev_flat_grad is None:
        st.prev_flat_grad = flat_grad.clone()
    else:
        st.prev_flat_grad.copy_(flat_grad)
    st.prev_loss = loss
    if st.n_iter == 1:
        t = min(1.0, 1.0 / flat_grad.abs().sum()) * lr
    else:
        t = lr
    gtd = flat_grad.dot(d)
    st.last_accept = None
    if not (gtd > -st.tol_c):
        x_init = x.clone()

        def obj_func(tt):
            x.copy_(x_init).add_(d, alpha=float(tt))
            l, g = closure(x)
            x.copy_(x_init)
            return l, g

        loss, flat_grad, t, ls_evals = _strong_wolfe(
            obj_func, t, d, loss, flat_grad, gtd, max_ls=st.max_eval - current_evals)
        x.add_(d, alpha=float(t))
        st.last_accept = bool(float(t) != 0.0)
        st.func_evals += ls_evals
    st.d = d
    st.t = t
    return orig_loss


# --------------------------------------------------------------------------
# the optimisation loop (neural_style_transfer.py:123-208)
# --------------------------------------------------------------------------
def run_process(content_levels: Sequence[np.ndarray], style_levels: Sequence[np.ndarray],
                init_img: np.ndarray, weights, optimizer: str, iters_num: int,
                cw: float = 1e3, sw: float = 4e5, tvw: float = 1e2, lr_start: float = 10.0,
                lbfgs_max_eval: int = 1, record: Optional[list] = None,
                as_reference: bool = False):
    """Generator restating ``NeuralStyleTransfer.process``: yields (img HWC float32, step)
    after every optimiser step.  ``record`` (a list) receives one dict per closure:
    {"loss", "rows", "lr"}.  ``as_reference=True`` also reproduces the reference's
    per-closure overheads for the timed CPU baseline (anomaly mode, the zero-weighted
    randn per level: neural_style_transfer.py:150, :91-93)."""
    x = prepare_img(init_img)
    shape = x.shape
    targets = [LevelTargets(prepare_img(c), prepare_img(s), weights)
               for c, s in zip(content_levels, style_levels)]
    state = {"lr": lr_start, "step": 0}
    if as_reference:
        torch.autograd.set_detect_anomaly(True)

    def closure(xf: torch.Tensor):
        state["lr"] *= 0.999                                   # :155-158
        if as_reference:
            for tg in targets:                                  # :91-93
                _ = 0 * torch.clip(0.5 * torch.randn(tg.content.shape) + 0.5, min=0.0, max=1.0)
        loss, grad, rows = closure_eval(xf.view(shape), targets, weights, cw, sw, tvw)
        state["step"] += 1                                     # :198
        if record is not None:
            record.append({"loss": float(loss), "rows": rows, "lr": state["lr"]})
        return float(loss), grad.reshape(-1)

    xf = x.reshape(-1).clone()
    try:
        if optimizer == "adam":
            adam = AdamState(xf.numel())
            while state["step"] < iters_num:
                _, g = closure(xf)
                adam.update(xf, g, state["lr"])                # lr already decayed
                yield unprepare_img(xf.view(shape)), state["step"]
        elif optimizer == "lbfgs":
            lb = LbfgsState(max_eval=lbfgs_max_eval)
            while state["step"] < iters_num:
                lbfgs_step(lb, xf, state["lr"], closure)
                yield unprepare_img(xf.view(shape)), state["step"]
        else:
            raise RuntimeError("Unknown optimizer")            # :137-138
    finally:
        if as_reference:
            torch.autograd.set_detect_anomaly(False)
