"""GPU: accuracy of the VGG19 maps and of the closure against an fp64 evaluation, for the current NST_CONV mode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cpu_ref
from artstyletransfer_amd.engine import StyleEngine

def rel(a, b):
    a = a.double(); b = b.double()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))

w = cpu_ref.synthetic_vgg19_weights()
w64 = [(a.double(), b.double()) for a, b in w]
eng = StyleEngine(w, 0)
print("NST_CONV =", os.environ.get("NST_CONV", "(default f16x2)"))
for h, wd in [(64, 96), (128, 192), (256, 384)]:
    img = cpu_ref.synthetic_image(h, wd, seed=3)
    x = cpu_ref.prepare_img(img).contiguous()
    ref64 = cpu_ref.vgg19_features(x.double(), w64)
    ref32 = cpu_ref.vgg19_features(x, w)
    outs = eng.vgg_features(x.cuda())
    print(f"{h}x{wd} maps vs fp64:  hip " + " ".join(f"{rel(o.cpu(), r):.1e}" for o, r in zip(outs, ref64))
          + "   torch-fp32 " + " ".join(f"{rel(o, r):.1e}" for o, r in zip(ref32, ref64)))

# closure: losses and gradient against the same closure in fp64
import torch.nn.functional as F
def levels(h, w, n, seed):
    img = cpu_ref.synthetic_image(h, w, seed=seed)
    out = [img]
    t = torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0)
    for l in range(1, n):
        out.append(F.interpolate(t, size=(h >> l, w >> l), mode="bicubic", align_corners=False).squeeze(0).permute(1, 2, 0).contiguous().numpy())
    return out
for h, wd, n in [(128, 192, 2), (256, 384, 2)]:
    c, s = levels(h, wd, n, 1), levels(h, wd, n, 2)
    eng.configure(n, h, wd)
    for l in range(n):
        eng.set_targets(l, cpu_ref.prepare_img(c[l]).contiguous().cuda(), cpu_ref.prepare_img(s[l]).contiguous().cuda())
    xt = cpu_ref.prepare_img((0.7 * c[0] + 0.3 * cpu_ref.synthetic_image(h, wd, seed=9)).astype(np.float32)).contiguous()
    tg32 = [cpu_ref.LevelTargets(cpu_ref.prepare_img(a), cpu_ref.prepare_img(b), w) for a, b in zip(c, s)]
    tg64 = [cpu_ref.LevelTargets(cpu_ref.prepare_img(a).double(), cpu_ref.prepare_img(b).double(), w64) for a, b in zip(c, s)]
    l64, g64, _ = cpu_ref.closure_eval(xt.double(), tg64, w64, 1e3, 4e5, 1e2)
    l32, g32, _ = cpu_ref.closure_eval(xt, tg32, w, 1e3, 4e5, 1e2)
    g, ls = eng.closure(xt.cuda(), 1e3, 4e5, 1e2)
    print(f"{h}x{wd} L{n-1} closure vs fp64: loss rel hip {abs(float(ls[-1]) - float(l64)) / float(l64):.1e} torch-fp32 "
          f"{abs(float(l32) - float(l64)) / float(l64):.1e}; grad rel-L2 hip {rel(g.cpu(), g64):.2e} torch-fp32 {rel(g32, g64):.2e}")
