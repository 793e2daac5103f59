"""GPU: random geometries (odd sizes, 1-3 levels, foreign-size styles) - the default f16x2 closure against the
exact-f32-MFMA closure (NST_CONV=f32, per-level schedule) of the same library.  Losses must agree to 1e-5 and the
gradient to the flip-noise bound; a real indexing bug shows up as errors of order 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from artstyletransfer_amd import synthetic
from artstyletransfer_amd.engine import StyleEngine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
w = synthetic.vgg19_weights()


def levels(h, wd, n, seed):
    img = synthetic.image(h, wd, seed=seed)
    out = [img]
    t = torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0)
    for l in range(1, n):
        out.append(F.interpolate(t, size=(h >> l, wd >> l), mode="bicubic", align_corners=False).squeeze(0).permute(1, 2, 0).contiguous().numpy())
    return out


def run(mode, batch, geo, x):
    os.environ["NST_CONV"] = mode
    os.environ["NST_BATCH"] = batch
    eng = StyleEngine(w, 0)
    h, wd, n, hs, ws = geo
    c, s = levels(h, wd, n, 1), levels(hs, ws, n, 2)
    eng.configure(n, h, wd)
    for l in range(n):
        eng.set_targets(l, eng.prepare_img(torch.from_numpy(c[l]).cuda()), eng.prepare_img(torch.from_numpy(s[l]).cuda()))
    g, ls = eng.closure(x, 1e3, 4e5, 1e2)
    torch.cuda.synchronize()
    out = (g.double().cpu(), ls.double().cpu())
    eng.close()
    return out


worst = (0.0, 0.0)
for case in range(n_cases):
    n = int(rng.randint(1, 4))
    lo = 16 << (n - 1)
    h = int(rng.randint(lo, 420)); wd = int(rng.randint(lo, 520))
    if rng.rand() < 0.3: h = (h // 16) * 16 or lo
    if rng.rand() < 0.3: wd = (wd // 16) * 16 or lo
    hs, ws = (h, wd) if rng.rand() < 0.5 else (int(rng.randint(lo, 400)), int(rng.randint(lo, 400)))
    geo = (h, wd, n, hs, ws)
    x = StyleEngine(w, 0)
    xi = x.prepare_img(torch.from_numpy((0.7 * synthetic.image(h, wd, seed=1) + 0.3 * synthetic.image(h, wd, seed=9)).astype(np.float32)).cuda())
    x.close()
    g0, l0 = run("f32", "0", geo, xi)
    g1, l1 = run("f16x2", "1", geo, xi)
    lrel = float(((l1 - l0).abs() / l0.abs().clamp_min(1e-30))[-1])
    rows = float(((l1 - l0).abs()[:-1].reshape(n, 4)[:, 0] / l0[:-1].reshape(n, 4)[:, 0].abs()).max())
    grel = float((g1 - g0).norm() / g0.norm())
    ok = lrel < 1e-5 and rows < 2e-5 and grel < 1e-2 and bool(torch.isfinite(g1).all())
    worst = (max(worst[0], lrel), max(worst[1], grel))
    print(f"{case:3d} {geo} total rel {lrel:.1e} level rel {rows:.1e} grad rel-L2 {grel:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        sys.exit(1)
print("worst", worst)
