import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.nn.functional as F
from oracle import cpu_ref
import hip_helpers as hh
from artstyletransfer_amd.engine import StyleEngine
w = cpu_ref.synthetic_vgg19_weights(bias_std=2.0)
e = StyleEngine(w, 0)
def where(err):
    i = np.unravel_index(np.abs(err).argmax(), err.shape); return i
for (h, wd) in [(89, 320), (103, 151), (356, 151), (128, 190), (336, 77)]:
    g = torch.Generator().manual_seed(h)
    x = torch.randn(1, 3, h, wd, generator=g, requires_grad=True)
    y = cpu_ref.bicubic_half(x)
    for kind in ("random", "border"):
        gy = torch.randn(y.shape, generator=g)
        if kind == "border":
            m = torch.zeros_like(gy); m[:, :, 0, :] = 1; m[:, :, -1, :] = 1; m[:, :, :, 0] = 1; m[:, :, :, -1] = 1
            gy = gy * m
        x.grad = None
        (cpu_ref.bicubic_half(x) * gy).sum().backward()
        gx = e.bicubic_half_backward(hh.dev(gy), h, wd).cpu()
        err = (gx - x.grad).numpy()
        print(f"bicubic bwd {h}x{wd} gy {kind}: rel-L2 {hh.rel_l2(gx.numpy(), x.grad.numpy()):.2e} worst at {where(err)}", flush=True)
    out = e.bicubic_half(hh.dev(x.detach())).cpu()
    print(f"bicubic fwd {h}x{wd}: rel-L2 {hh.rel_l2(out.numpy(), y.detach().numpy()):.2e}")
    # TV gradient on the level-1 image of a smooth picture
    img = cpu_ref.prepare_img(cpu_ref.synthetic_image(h, wd, 1))
    l1 = cpu_ref.bicubic_half(img).detach().requires_grad_(True)
    cpu_ref.total_variation(l1).backward()
    val, gt = e.total_variation(hh.dev(l1.detach()), want_grad=True)
    err = (gt.cpu() - l1.grad).numpy()
    print(f"tv grad on level-1 {tuple(l1.shape[2:])}: rel-L2 {hh.rel_l2(gt.cpu().numpy(), l1.grad.numpy()):.2e} worst at {where(err)}", flush=True)
    # TV-only closure, level 1 only
    c, s = hh.levels(h, wd, 2, 1), hh.levels(h, wd, 2, 2)
    hh.setup(e, c, s)
    tg = hh.oracle_targets(c, s, w)
    xt = cpu_ref.prepare_img((0.7 * c[0] + 0.3 * cpu_ref.synthetic_image(h, wd, seed=9)).astype(np.float32))
    for mask in (1, 2, 3):
        gd, l = e.closure_levels(hh.dev(xt), 0.0, 0.0, 1e2, mask)
        xx = xt.clone().requires_grad_(True)
        lv = [xx, cpu_ref.bicubic_half(xx)]
        tot = sum(1e2 * cpu_ref.total_variation(lv[i]) for i in range(2) if (mask >> i) & 1)
        tot.backward()
        err = (gd.cpu() - xx.grad).numpy()
        print(f"tv-only closure {h}x{wd} mask {mask}: rel-L2 {hh.rel_l2(gd.cpu().numpy(), xx.grad.numpy()):.2e} worst at {where(err)}; loss {float(l[-1]):.6e} vs {float(tot):.6e}", flush=True)
e.close()
