"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Run only in the build container (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py [--only NAME]

The reference's hot path is pure Python on top of torch; torch (CPU) is installed, so
its own ``math_utils``, ``neural_nets.Vgg19``, ``LossBuilder``, ``prepare_img`` and
``NeuralStyleTransfer.process`` are imported and executed unmodified.  Two packages it
imports are absent offline and are supplied as in-process stand-in modules that carry
NO hot-path arithmetic:

* ``torchvision``: ``models.vgg19()`` returns the cfg-"E" ``features`` ``nn.Sequential``
  (Conv2d 3x3 pad 1 / ReLU(inplace) / MaxPool2d 2x2) filled with the seeded synthetic
  weights (no pretrained file exists offline); ``transforms.Compose/ToTensor/Lambda/
  Normalize`` with torchvision's documented semantics for float32 HWC input.
* ``cv2``: the four operators the reference's job set-up calls - ``resize(INTER_CUBIC)``, ``Sobel(ksize=5)``,
  ``GaussianBlur``, ``getGaussianKernel`` - delegate to ``oracle/cv2_ref.py`` (the tap-by-tap restatement of OpenCV's
  published definitions that ``tests/test_oracle_cv2.py`` pins against torch / scipy).  Everything AROUND those four
  calls - the pyramid order and size rule, the granularity -> spot-grid rule, the envelope accumulation, the draw order
  of ``np.random.permutation``, the Sobel / clip / blur / ``a = 5`` weight, the init-method branch
  (neural_style_transfer.py:211-226, :249-362, :396-439) - is the reference's own Python, executed unmodified by the
  ``jobsetup`` and ``config3`` fixtures.  The hot-path fixtures call no cv2 function.

Fixtures hold data only (inputs, expected outputs); no reference source text.
"""
from __future__ import annotations

import argparse
import asyncio
import contextlib
import io
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("NST_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import cpu_ref  # noqa: E402  (synthetic input generators only)


# ------------------------------------------------------------------ stand-ins
def install_standins(weights):
    from oracle import cv2_ref

    cv2 = types.ModuleType("cv2")
    cv2.INTER_CUBIC = 2
    cv2.CV_64F = 6

    def _absent(*a, **k):
        raise RuntimeError("cv2 is absent offline; this function is not supplied by the stand-in")

    def resize(src, dsize=None, interpolation=None, **kw):
        assert interpolation == cv2.INTER_CUBIC and not kw, (interpolation, kw)
        nw, nh = dsize
        return cv2_ref.resize_cubic(np.asarray(src), int(nh), int(nw))

    def Sobel(src, ddepth=None, dx=None, dy=None, ksize=3, **kw):
        assert ddepth == cv2.CV_64F and ksize == 5 and not kw and (dx, dy) in ((1, 0), (0, 1))
        return cv2_ref.sobel5(src, dx, dy)

    def GaussianBlur(src, ksize=None, sigmaX=None, **kw):
        assert ksize[0] == ksize[1] and sigmaX > 0 and not kw
        return cv2_ref.gaussian_blur(src, int(ksize[0]), float(sigmaX))

    def getGaussianKernel(n, sigma):
        assert sigma > 0
        return cv2_ref.get_gaussian_kernel(int(n), float(sigma)).reshape(-1, 1)        # cv2 returns an (n, 1) column

    cv2.resize, cv2.Sobel, cv2.GaussianBlur, cv2.getGaussianKernel = resize, Sobel, GaussianBlur, getGaussianKernel
    for name in ("imwrite", "cvtColor"):
        setattr(cv2, name, _absent)
    sys.modules["cv2"] = cv2

    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    transforms = types.ModuleType("torchvision.transforms")

    cfg_e = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M",
             512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]

    def vgg19(pretrained=False, progress=True, **kw):
        layers, cin, wi = [], 3, 0
        for v in cfg_e:
            if v == "M":
                layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                conv = torch.nn.Conv2d(cin, v, kernel_size=3, padding=1)
                if wi < len(weights):
                    with torch.no_grad():
                        conv.weight.copy_(weights[wi][0])
                        conv.bias.copy_(weights[wi][1])
                wi += 1
                layers += [conv, torch.nn.ReLU(inplace=True)]
                cin = v
        net = types.SimpleNamespace()
        net.features = torch.nn.Sequential(*layers)
        return net

    models.vgg19 = vgg19

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor:
        def __call__(self, pic):
            assert isinstance(pic, np.ndarray) and pic.dtype == np.float32 and pic.ndim == 3
            return torch.from_numpy(np.ascontiguousarray(pic.transpose(2, 0, 1)))

    class Lambda:
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, x):
            return self.fn(x)

    class Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
            self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    transforms.Compose, transforms.ToTensor = Compose, ToTensor
    transforms.Lambda, transforms.Normalize = Lambda, Normalize
    tv.models, tv.transforms = models, transforms
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models
    sys.modules["torchvision.transforms"] = transforms


def import_reference():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import math_utils as ref_mu
        import neural_nets as ref_nn
        import neural_style_transfer as ref_nst
    return ref_mu, ref_nn, ref_nst


def sample_idx(n, k, seed):
    return np.random.RandomState(seed).randint(0, n, size=k).astype(np.int64)


def summarize(t: torch.Tensor, k=64, seed=7):
    a = t.detach().reshape(-1).double()
    idx = sample_idx(a.numel(), k, seed)
    return {
        "shape": np.array(t.shape, dtype=np.int64),
        "sum": np.float64(a.sum()),
        "abs_sum": np.float64(a.abs().sum()),
        "sq_sum": np.float64((a * a).sum()),
        "idx": idx,
        "val": t.detach().reshape(-1)[torch.from_numpy(idx)].numpy().astype(np.float32),
    }


def save(name, **arrays):
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}.{kk}"] = vv
        else:
            flat[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


# ------------------------------------------------------------------ fixtures
def fx_kat(ref_mu, ref_nn, ref_nst, weights):
    """Weight-free known answers straight from the reference's own functions."""
    x = torch.arange(24, dtype=torch.float32).reshape(1, 2, 3, 4)
    g = ref_mu.gram_matrix(x)
    gu = ref_mu.gram_matrix(x, should_normalize=False)
    y = (torch.arange(120).reshape(2, 3, 4, 5) ** 2 % 7).float()
    tv = ref_mu.total_variation(y)
    img = np.linspace(0, 1, 72, dtype=np.float32).reshape(4, 6, 3)
    p = ref_nst.prepare_img(img, "cpu")
    back = ref_nst.unprepare_img(p.clone())
    r = torch.Generator().manual_seed(5)
    xr = torch.randn(1, 5, 7, 9, generator=r)
    save("kat",
         gram_in=x.numpy(), gram=g.numpy(), gram_unnorm=gu.numpy(),
         tv_in=y.numpy(), tv=np.float32(tv),
         prep_in=img, prep=p.numpy(), unprep=back,
         gram_rand_in=xr.numpy(), gram_rand=ref_mu.gram_matrix(xr).numpy(),
         tv_rand=np.float32(ref_mu.total_variation(xr)))


def fx_bicubic(ref_mu, ref_nn, ref_nst, weights):
    """The reference's down-sample call (neural_style_transfer.py:173-176) fwd + autograd bwd,
    on an even size (exact 1/2) and on an odd size (general scale)."""
    out = {}
    for tag, (h, w) in (("even", (16, 24)), ("odd", (15, 23)), ("tiny", (4, 6))):
        g = torch.Generator().manual_seed(11)
        x = torch.randn(1, 3, h, w, generator=g, requires_grad=True)
        sw, sh = x.shape[2], x.shape[3]
        y = torch.nn.functional.interpolate(x, size=(sw // 2, sh // 2), mode="bicubic")
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        out[f"{tag}_x"] = x.detach().numpy()
        out[f"{tag}_y"] = y.detach().numpy()
        out[f"{tag}_gy"] = gy.numpy()
        out[f"{tag}_gx"] = x.grad.numpy()
    save("bicubic", **out)


def fx_vgg(ref_mu, ref_nn, ref_nst, weights):
    """Reference Vgg19 forward on a (1,3,48,80) input + d(sum of weighted outputs)/dx."""
    with contextlib.redirect_stdout(io.StringIO()):
        net, cidx, sidx = ref_mu.prepare_model("vgg19", "cpu")
    img = cpu_ref.synthetic_image(48, 80, seed=3)
    x = ref_nst.prepare_img(img, "cpu").requires_grad_(True)
    outs = net(x)
    arrays = {"img": img, "content_index": np.int64(cidx), "style_indices": np.array(sidx, dtype=np.int64),
              "layer_names": np.array(list(net.layer_names))}
    loss = 0
    g = torch.Generator().manual_seed(21)
    for i, o in enumerate(outs):
        arrays[f"out{i}"] = summarize(o, seed=100 + i)
        wgt = torch.randn(o.shape, generator=g) / o.numel()
        loss = loss + (o * wgt).sum()
        arrays[f"gram{i}"] = summarize(ref_mu.gram_matrix(o), seed=200 + i)
    loss.backward()
    arrays["grad_seed"] = np.int64(21)
    arrays["grad"] = x.grad.numpy()
    # smallest full maps kept whole (cheap): relu5_1 and ReLU(conv4_2)
    arrays["out5_full"] = outs[5].detach().numpy()
    arrays["out4_full"] = outs[4].detach().numpy()
    save("vgg_48x80", **arrays)


def _levels(h, w, nlev, seed):
    """Pyramid of synthetic images, highest-res first, each level generated from the
    level-0 'original' by bicubic resize (same rule cv2 INTER_CUBIC documents)."""
    top = cpu_ref.synthetic_image(h, w, seed)
    out = [top]
    t = torch.from_numpy(top).permute(2, 0, 1).unsqueeze(0)
    for l in range(1, nlev):
        d = torch.nn.functional.interpolate(t, size=(h >> l, w >> l), mode="bicubic", align_corners=False)
        out.append(d.squeeze(0).permute(1, 2, 0).contiguous().numpy())
    return out


def _ref_closure(ref_mu, ref_nst, content_levels, style_levels, x_img, cw, sw, tvw):
    """Teacher-forced closure through the reference's LossBuilder (one call, no optimiser)."""
    with contextlib.redirect_stdout(io.StringIO()):
        net, cidx, sidx = ref_mu.prepare_model("vgg19", "cpu")
    builders = [ref_nst.LossBuilder(cidx, sidx, ref_nst.prepare_img(c, "cpu"), ref_nst.prepare_img(s, "cpu"),
                                    net, cw, sw, tvw) for c, s in zip(content_levels, style_levels)]
    x = ref_nst.prepare_img(x_img, "cpu").requires_grad_(True)
    levels, total, rows = [x], None, []
    for i, b in enumerate(builders):
        if i > 0:
            p = levels[i - 1]
            levels.append(torch.nn.functional.interpolate(p, size=(p.shape[2] // 2, p.shape[3] // 2), mode="bicubic"))
        t, c, s, tv = b.build(levels[i])
        total = t if total is None else 1.0 * total + t
        rows.append([float(t), float(c), float(s), float(tv)])
    total.backward()
    return float(total), np.array(rows, dtype=np.float64), x.grad.detach()


def _ref_term_grads(ref_mu, ref_nst, content_levels, style_levels, x_img):
    """The pixel gradient of each loss term alone - (cw,0,0), (0,sw,0), (0,0,tvw) through the same LossBuilder -
    so that a parity test can hold every term separately (a term that is < 3e-3 of the summed gradient's norm could
    otherwise be wrong or absent unnoticed)."""
    out = {}
    for tag, wts in (("c", (1e3, 0.0, 0.0)), ("s", (0.0, 4e5, 0.0)), ("tv", (0.0, 0.0, 1e2))):
        total, _, grad = _ref_closure(ref_mu, ref_nst, content_levels, style_levels, x_img, *wts)
        out[f"grad_{tag}"] = grad.numpy()
        out[f"total_{tag}"] = np.float64(total)
    return out


def fx_closure_small(ref_mu, ref_nn, ref_nst, weights):
    """2 levels, 64x96 + 32x48: inputs, losses and the whole gradient."""
    cl = _levels(64, 96, 2, seed=1)
    sl = _levels(64, 96, 2, seed=2)
    x_img = (0.6 * cl[0] + 0.4 * cpu_ref.synthetic_image(64, 96, seed=9)).astype(np.float32)
    total, rows, grad = _ref_closure(ref_mu, ref_nst, cl, sl, x_img, 1e3, 4e5, 1e2)
    save("closure_64x96_L1", content0=cl[0], content1=cl[1], style0=sl[0], style1=sl[1], x_img=x_img,
         total=np.float64(total), rows=rows, grad=grad.numpy(),
         **_ref_term_grads(ref_mu, ref_nst, cl, sl, x_img))


def fx_closure_odd(ref_mu, ref_nn, ref_nst, weights):
    """1 level with awkward (non multiple of 16) size 50x76 and a different-size style image."""
    c = [cpu_ref.synthetic_image(50, 76, seed=1)]
    s = [cpu_ref.synthetic_image(44, 58, seed=2)]
    x_img = (0.5 * c[0] + 0.5 * cpu_ref.synthetic_image(50, 76, seed=9)).astype(np.float32)
    total, rows, grad = _ref_closure(ref_mu, ref_nst, c, s, x_img, 1e3, 4e5, 1e2)
    save("closure_50x76_L0", content0=c[0], style0=s[0], x_img=x_img,
         total=np.float64(total), rows=rows, grad=grad.numpy(), **_ref_term_grads(ref_mu, ref_nst, c, s, x_img))


def fx_closure_L0(ref_mu, ref_nn, ref_nst, weights):
    """256x384 single level (BASELINE config 1 geometry): losses + gradient summary; inputs from seeds."""
    cl = _levels(256, 384, 1, seed=1)
    sl = _levels(256, 384, 1, seed=2)
    x_img = cl[0]
    total, rows, grad = _ref_closure(ref_mu, ref_nst, cl, sl, x_img, 1e3, 4e5, 1e2)
    save("closure_256x384_L0", total=np.float64(total), rows=rows, grad=summarize(grad, k=256, seed=31),
         content_sum=np.float64(cl[0].astype(np.float64).sum()), style_sum=np.float64(sl[0].astype(np.float64).sum()))


def _run_reference_process(ref_mu, ref_nst, content_levels, style_levels, init_img, optimizer, iters,
                           lbfgs_max_eval=None):
    """Runs the reference's NeuralStyleTransfer.process unmodified; records per-closure rows by
    wrapping LossBuilder.build, and the image after every optimiser step."""
    rec, imgs = [], []
    orig_build = ref_nst.LossBuilder.build

    def build(self, x):
        out = orig_build(self, x)
        rec.append([float(v) for v in out])
        return out

    ref_nst.LossBuilder.build = build
    orig_lbfgs = ref_nst.LBFGS
    if lbfgs_max_eval is not None:
        # legacy line-search behaviour through the constructor argument only
        ref_nst.LBFGS = lambda params, **kw: orig_lbfgs(params, max_eval=lbfgs_max_eval, **kw)
    try:
        nst = ref_nst.NeuralStyleTransfer(torch.device("cpu"), "vgg19", style_levels, optimizer)

        async def go():
            async for img, step in nst.process(content_levels, init_img, 10.0, iters, 1e3, 4e5, 1e2, "fx"):
                imgs.append((img.copy(), step))

        with contextlib.redirect_stdout(io.StringIO()):
            asyncio.run(go())
    finally:
        ref_nst.LossBuilder.build = orig_build
        ref_nst.LBFGS = orig_lbfgs
        torch.autograd.set_detect_anomaly(False)
    nlev = len(content_levels)
    rows = np.array(rec, dtype=np.float64).reshape(-1, nlev, 4)
    return rows, imgs


def fx_adam_L0(ref_mu, ref_nn, ref_nst, weights):
    """BASELINE config 1: single 384x256 level, 50 Adam iterations, content image as init."""
    cl = _levels(256, 384, 1, seed=1)
    sl = _levels(256, 384, 1, seed=2)
    rows, imgs = _run_reference_process(ref_mu, ref_nst, cl, sl, cl[0], "adam", 50)
    final = torch.from_numpy(imgs[-1][0])
    save("traj_adam_256x384_50", rows=rows, steps=np.array([s for _, s in imgs], dtype=np.int64),
         final=summarize(final, k=256, seed=41),
         img_after_1=summarize(torch.from_numpy(imgs[0][0]), k=64, seed=42))


def fx_adam_small(ref_mu, ref_nn, ref_nst, weights):
    """2-level 64x96 Adam, 12 iterations, whole final image stored."""
    cl = _levels(64, 96, 2, seed=1)
    sl = _levels(64, 96, 2, seed=2)
    rows, imgs = _run_reference_process(ref_mu, ref_nst, cl, sl, cl[0], "adam", 12)
    save("traj_adam_64x96_L1_12", rows=rows, steps=np.array([s for _, s in imgs], dtype=np.int64),
         final=imgs[-1][0], after_1=imgs[0][0])


def fx_lbfgs_small(ref_mu, ref_nn, ref_nst, weights):
    """2-level 128x192 L-BFGS, 40 closures, as shipped (torch 2.10: max_ls = 0) and with
    max_eval=26 (legacy line search)."""
    cl = _levels(128, 192, 2, seed=1)
    sl = _levels(128, 192, 2, seed=2)
    for tag, me, iters in (("shipped", None, 40), ("legacy", 26, 40)):
        rows, imgs = _run_reference_process(ref_mu, ref_nst, cl, sl, cl[0], "lbfgs", iters, lbfgs_max_eval=me)
        accepted = [True]
        for (a, _), (b, _) in zip(imgs[:-1], imgs[1:]):
            accepted.append(bool(np.any(a != b)))
        accepted[0] = bool(np.any(imgs[0][0] != cl[0]))
        save(f"traj_lbfgs_128x192_L1_{tag}", rows=rows,
             steps=np.array([s for _, s in imgs], dtype=np.int64),
             moved=np.array(accepted, dtype=np.bool_),
             final=summarize(torch.from_numpy(imgs[-1][0]), k=256, seed=51))


def _moved(imgs, init):
    out, prev = [], init
    for img, _ in imgs:
        out.append(bool(np.any(img != prev)))
        prev = img
    return np.array(out, dtype=np.bool_)


def fx_lbfgs_config2(ref_mu, ref_nn, ref_nst, weights):
    """BASELINE config 2 at its full iteration count: L=1 (768x512 + 384x256), L-BFGS exactly as the reference
    constructs it, 500 closures, content top level as the initial image.  ~25 min of CPU."""
    cl = _levels(512, 768, 2, seed=1)
    sl = _levels(512, 768, 2, seed=2)
    rows, imgs = _run_reference_process(ref_mu, ref_nst, cl, sl, cl[0], "lbfgs", 500)
    save("traj_lbfgs_512x768_L1_500", rows=rows, steps=np.array([s for _, s in imgs], dtype=np.int64),
         moved=_moved(imgs, cl[0]), final=summarize(torch.from_numpy(imgs[-1][0]), k=1024, seed=61),
         after_1=summarize(torch.from_numpy(imgs[0][0]), k=1024, seed=62))


def fx_adam_config2geo(ref_mu, ref_nn, ref_nst, weights):
    """The config-2 geometry under Adam (a job whose image moves at every step), 100 iterations."""
    cl = _levels(512, 768, 2, seed=1)
    sl = _levels(512, 768, 2, seed=2)
    rows, imgs = _run_reference_process(ref_mu, ref_nst, cl, sl, cl[0], "adam", 100)
    save("traj_adam_512x768_L1_100", rows=rows, steps=np.array([s for _, s in imgs], dtype=np.int64),
         final=summarize(torch.from_numpy(imgs[-1][0]), k=1024, seed=63),
         after_1=summarize(torch.from_numpy(imgs[0][0]), k=1024, seed=64),
         after_10=summarize(torch.from_numpy(imgs[9][0]), k=1024, seed=65))


def fx_lbfgs_config2geo_legacy(ref_mu, ref_nn, ref_nst, weights):
    """The config-2 geometry under L-BFGS with the legacy 25-evaluation line search (max_eval=26 through the
    constructor argument only): steps are accepted and the curvature history grows; 100 closures."""
    cl = _levels(512, 768, 2, seed=1)
    sl = _levels(512, 768, 2, seed=2)
    rows, imgs = _run_reference_process(ref_mu, ref_nst, cl, sl, cl[0], "lbfgs", 100, lbfgs_max_eval=26)
    save("traj_lbfgs_512x768_L1_legacy_100", rows=rows, steps=np.array([s for _, s in imgs], dtype=np.int64),
         moved=_moved(imgs, cl[0]), final=summarize(torch.from_numpy(imgs[-1][0]), k=1024, seed=66))


def fx_config3_prefix(ref_mu, ref_nn, ref_nst, weights):
    """BASELINE config 3's workload - the headline one: L=2 (1536x1024 + 768x512 + 384x256) - for as many iterations as
    the CPU reference affords here (~10 s per closure on 8 cores): 16 Adam iterations (the image moves at every step) and
    16 closures of L-BFGS as the reference constructs it.  Initial image: content top level blended with synthetic noise
    (the reference's own content+noise init needs cv2).  Per-closure loss rows of all three levels, sampled pixels of
    the images after step 1 and at the end."""
    cl = _levels(1024, 1536, 3, seed=1)
    sl = _levels(1024, 1536, 3, seed=2)
    init = (0.7 * cl[0] + 0.3 * cpu_ref.synthetic_image(1024, 1536, seed=3)).astype(np.float32)
    for opt, iters in (("adam", 16), ("lbfgs", 16)):
        rows, imgs = _run_reference_process(ref_mu, ref_nst, cl, sl, init, opt, iters)
        save(f"traj_{opt}_1024x1536_L2_16", rows=rows, steps=np.array([s for _, s in imgs], dtype=np.int64),
             moved=_moved(imgs, init), final=summarize(torch.from_numpy(imgs[-1][0]), k=2048, seed=71),
             after_1=summarize(torch.from_numpy(imgs[0][0]), k=2048, seed=72))


# ------------------------------------------------------------------ the reference's own job driver (f-1 / f-2)
def _run_reference_job(ref_nst, content, style, cfg, seed, full_run=False):
    """Runs the reference's own ``neural_style_transfer()`` (neural_style_transfer.py:229-372) with the cv2 stand-in
    above.  ``NeuralStyleTransfer.process`` is wrapped to capture what the job driver hands to the hot path - the
    content / style pyramids and the initial image - and, unless ``full_run``, to return at once.  ``np.random.seed(seed)``
    is set immediately before the call (the reference draws its permutations from the global generator).  Returns
    (captured dict, per-closure loss rows, [(percent, img), ...])."""
    cap, rec, out = {}, [], []
    orig_process = ref_nst.NeuralStyleTransfer.process
    orig_init = ref_nst.NeuralStyleTransfer.__init__
    orig_build = ref_nst.LossBuilder.build

    def init(self, device, model_name, style_imgs, optimizer_name):
        cap["style_imgs"] = [np.array(a, copy=True) for a in style_imgs]
        cap["device"] = str(device)
        orig_init(self, device, model_name, style_imgs, optimizer_name)

    async def process(self, content_imgs, init_img, lr_start, iters_num, cw, sw, tvw, init_img_name):
        cap["content_imgs"] = [np.array(a, copy=True) for a in content_imgs]
        cap["init_img"] = np.array(init_img, copy=True)
        cap["init_img_name"] = init_img_name
        cap["lr_start"], cap["iters_num"], cap["weights"] = lr_start, iters_num, (cw, sw, tvw)
        if full_run:
            async for item in orig_process(self, content_imgs, init_img, lr_start, iters_num, cw, sw, tvw, init_img_name):
                yield item

    def build(self, x):
        res = orig_build(self, x)
        rec.append([float(v) for v in res])
        return res

    ref_nst.NeuralStyleTransfer.process = process
    ref_nst.NeuralStyleTransfer.__init__ = init
    ref_nst.LossBuilder.build = build
    try:
        pair = ref_nst.ContentStylePair(("content-name", content), ("style-name", style))

        async def go():
            np.random.seed(seed)
            async for percent, img in ref_nst.neural_style_transfer(
                    pair, cfg.content_weight, cfg.style_weight, cfg.tv_weight, cfg.optimizer, cfg.model, cfg.init_method,
                    cfg.iters_num, cfg.levels_num, cfg.noise_factor, cfg.noise_levels, cfg.noise_levels_central_amplitude,
                    cfg.noise_levels_peripheral_amplitude, cfg.noise_levels_dispersion):
                out.append((float(percent), img.copy()))
                if len(out) % 10 == 0:
                    print(f"   ... {len(out)} optimiser steps, {len(rec)} level evaluations", file=sys.stderr, flush=True)

        with contextlib.redirect_stdout(io.StringIO()):
            asyncio.run(go())
    finally:
        ref_nst.NeuralStyleTransfer.process = orig_process
        ref_nst.NeuralStyleTransfer.__init__ = orig_init
        ref_nst.LossBuilder.build = orig_build
        torch.autograd.set_detect_anomaly(False)
    nlev = len(cap["content_imgs"])
    rows = np.array(rec, dtype=np.float64).reshape(-1, nlev, 4) if rec and full_run else np.zeros((0, nlev, 4))
    return cap, rows, out


from jobsetup_cases import JOBSETUP_CASES  # noqa: E402  (numbers only: sizes, seeds, Config keyword arguments)


def fx_jobsetup(ref_mu, ref_nn, ref_nst, weights):
    """What the reference's job driver hands to the hot path (rows f-1 / f-2): level shapes, ordering, the content and
    style pyramids and the structured-noise initial image, produced by the reference's own neural_style_transfer()."""
    import config as ref_cfg
    arrays = {"cases": np.array([c[0] for c in JOBSETUP_CASES])}
    for tag, (ch, cw_, cs), (sh, sw_, ss), seed, kw in JOBSETUP_CASES:
        content = cpu_ref.synthetic_image(ch, cw_, cs)
        style = cpu_ref.synthetic_image(sh, sw_, ss)
        cfg = ref_cfg.Config(iters_num=1, **kw)
        cap, _, _ = _run_reference_job(ref_nst, content, style, cfg, seed)
        n = len(cap["content_imgs"])
        assert n == cfg.levels_num == len(cap["style_imgs"])
        arrays[f"{tag}.content_shapes"] = np.array([a.shape for a in cap["content_imgs"]], dtype=np.int64)
        arrays[f"{tag}.style_shapes"] = np.array([a.shape for a in cap["style_imgs"]], dtype=np.int64)
        for l in range(n):
            for k, v in summarize(torch.from_numpy(np.ascontiguousarray(cap["content_imgs"][l])), k=2048, seed=300 + l).items():
                arrays[f"{tag}.content{l}.{k}"] = v
            for k, v in summarize(torch.from_numpy(np.ascontiguousarray(cap["style_imgs"][l])), k=2048, seed=400 + l).items():
                arrays[f"{tag}.style{l}.{k}"] = v
        init = np.ascontiguousarray(cap["init_img"])
        arrays[f"{tag}.init_dtype"] = np.array(str(init.dtype))
        for k, v in summarize(torch.from_numpy(init), k=8192, seed=500).items():
            arrays[f"{tag}.init.{k}"] = v
        arrays[f"{tag}.init_name"] = np.array(cap["init_img_name"])
        arrays[f"{tag}.lr_start"] = np.float64(cap["lr_start"])
        print(f"   {tag}: levels {[tuple(a.shape) for a in cap['content_imgs']]}, init {init.shape} {init.dtype} "
              f"'{cap['init_img_name']}'")
    save("jobsetup", **arrays)


def _config3_inputs():
    return cpu_ref.synthetic_image(1024, 1536, seed=1), cpu_ref.synthetic_image(1024, 1536, seed=2)


def _fx_config3(ref_nst, tag, optimizer, iters, max_eval=None):
    """BASELINE config 3 through the reference's OWN job driver: neural_style_transfer() with Config() defaults except
    levels_num = 3 (L=2: 1536x1024 + 768x512 + 384x256), init_method 'content+noise' with np.random.seed(0) immediately
    before the call, synthetic 3:2 originals at the top level's size.  The start image is therefore the
    reference's own structured-noise image (cv2 operators from oracle/cv2_ref).  Recorded: per-closure loss rows of all
    three levels, sampled pixels of the start image and of the yielded images, the percent values."""
    import config as ref_cfg
    content, style = _config3_inputs()
    cfg = ref_cfg.Config(levels_num=3, optimizer=optimizer, iters_num=iters)
    orig_lbfgs = ref_nst.LBFGS
    if max_eval is not None:
        ref_nst.LBFGS = lambda params, **kw: orig_lbfgs(params, max_eval=max_eval, **kw)
    try:
        cap, rows, out = _run_reference_job(ref_nst, content, style, cfg, 0, full_run=True)
    finally:
        ref_nst.LBFGS = orig_lbfgs
    init = np.ascontiguousarray(cap["init_img"])
    imgs = [(img, p) for p, img in out]
    arrays = dict(rows=rows, percent=np.array([p for p, _ in out], dtype=np.float64),
                  moved=_moved(imgs, init), init=summarize(torch.from_numpy(init), k=8192, seed=500),
                  final=summarize(torch.from_numpy(out[-1][1]), k=4096, seed=71),
                  after_1=summarize(torch.from_numpy(out[0][1]), k=4096, seed=72))
    for j in (1, 3, 9, 49):
        if j < len(out) - 1:
            arrays[f"after_{j + 1}"] = summarize(torch.from_numpy(out[j][1]), k=4096, seed=73 + j)
    # closures consumed by every optimizer.step: percent = step / iters * 100 with step = the closure counter (:370)
    arrays["steps"] = np.rint(arrays["percent"] / 100.0 * iters).astype(np.int64)
    save(f"job_{tag}_1024x1536_L2", **arrays)


def fx_config3_adam(ref_mu, ref_nn, ref_nst, weights):
    """100 Adam iterations of the config-3 job (~20 min of CPU)."""
    _fx_config3(ref_nst, "adam100", "adam", 100)


def fx_config3_lbfgs(ref_mu, ref_nn, ref_nst, weights):
    """24 closures of L-BFGS as the reference constructs it, and 40 closures of the legacy 25-evaluation line search."""
    _fx_config3(ref_nst, "lbfgs24", "lbfgs", 24)
    _fx_config3(ref_nst, "lbfgs_legacy40", "lbfgs", 40, max_eval=26)


ALL = {f.__name__[3:]: f for f in (fx_kat, fx_bicubic, fx_vgg, fx_closure_small, fx_closure_odd, fx_closure_L0,
                                   fx_adam_small, fx_lbfgs_small, fx_adam_L0, fx_adam_config2geo,
                                   fx_lbfgs_config2geo_legacy, fx_lbfgs_config2, fx_config3_prefix, fx_jobsetup,
                                   fx_config3_adam, fx_config3_lbfgs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    torch.manual_seed(0)
    weights = cpu_ref.synthetic_vgg19_weights(bias_std=cpu_ref.TEST_BIAS_STD)     # non-zero biases
    install_standins(weights)
    ref_mu, ref_nn, ref_nst = import_reference()
    for name, fn in ALL.items():
        if args.only and name not in args.only:
            continue
        print(f"== {name}")
        fn(ref_mu, ref_nn, ref_nst, weights)


if __name__ == "__main__":
    main()
