"""GPU: where the exact-f32 per-level closure's gradient leaves the oracle's (under equal decisions) on odd geometries."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import cpu_ref
import hip_helpers as hh
from artstyletransfer_amd.engine import StyleEngine

w = cpu_ref.synthetic_vgg19_weights(bias_std=cpu_ref.TEST_BIAS_STD)
geos = [(89, 320, 2, 89, 320), (103, 151, 3, 136, 329)]
for geo in geos:
    h, wd, nlev, hs, ws = geo
    c, s = hh.levels(h, wd, nlev, 1), hh.levels(hs, ws, nlev, 2)
    xt = cpu_ref.prepare_img((0.7 * c[0] + 0.3 * cpu_ref.synthetic_image(h, wd, seed=9)).astype(np.float32))
    tg = hh.oracle_targets(c, s, w)
    for opts in (dict(conv_mode="f32", batched=False), dict(conv_mode="f32", batched=False, single_stream=True),
                 dict(conv_mode="f16x2", batched=False), dict(conv_mode="bf16x3", batched=False), dict(conv_mode="bf16x3")):
        e = StyleEngine(w, 0, **opts)
        hh.setup(e, c, s)
        for name, wts in hh.TERMS[:3]:
            for mask in [None] + [1 << l for l in range(nlev)]:
                if mask is None:
                    g, l = e.closure(hh.dev(xt), *wts)
                else:
                    g, l = e.closure_levels(hh.dev(xt), *wts, mask)
                dec = hh.device_decisions(e)
                # oracle restricted to the same levels: zero weights elsewhere is not expressible; evaluate per level by hand
                x = xt.clone().requires_grad_(True)
                lv = [x]
                for i in range(1, nlev):
                    lv.append(cpu_ref.bicubic_half(lv[-1]))
                tot = 0
                for i in range(nlev):
                    if mask is None or (mask >> i) & 1:
                        tot = tot + cpu_ref.level_loss(lv[i], tg[i], w, *wts, decisions=dec[i])[0]
                tot.backward()
                gr = x.grad.numpy()
                gd = g.cpu().numpy()
                err = np.abs(gd - gr)[0]
                # where is the error: border rows/cols vs interior
                tot_e = float(np.linalg.norm(err) / np.linalg.norm(gr))
                inner = err[:, 8:-8, 8:-8]
                print(f"{geo} {opts} [{name}] levels {'all' if mask is None else bin(mask)}: rel-L2 {tot_e:.2e}; interior(8px) share of err^2 "
                      f"{float((inner ** 2).sum() / max((err ** 2).sum(), 1e-300)):.2f}; argmax err at {np.unravel_index(err.argmax(), err.shape)} of {err.shape}", flush=True)
        e.close()
