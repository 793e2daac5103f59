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
print("NST_CONV =", os.environ.get("NST_CONV", "(default bf16x3)"))
for h, wd in [(64, 96), (128, 192), (256, 384)]:
    img = cpu_ref.synthetic_image(h, wd, seed=3)
    x = cpu_ref.prepare_img(img).contiguous()
    ref64 = cpu_ref.vgg19_features(x.double(), w64)
    ref32 = cpu_ref.vgg19_features(x, w)
    outs = eng.vgg_features(x.cuda())
    print(f"{h}x{wd} maps vs fp64:  hip " + " ".join(f"{rel(o.cpu(), r):.1e}" for o, r in zip(outs, ref64))
          + "   torch-fp32 " + " ".join(f"{rel(o, r):.1e}" for o, r in zip(ref32, ref64)))
