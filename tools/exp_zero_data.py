"""GPU experiment: the same closure on all-zero data (weights and images) vs random data.
Identical instruction stream; zero operands draw less power, so a large speed-up = the kernel is clock/power bound."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from artstyletransfer_amd import synthetic
from artstyletransfer_amd.engine import StyleEngine

def run(zero):
    w = synthetic.vgg19_weights()
    if zero:
        w = [(torch.zeros_like(a), torch.zeros_like(b)) for a, b in w]
    eng = StyleEngine(w, 0)
    H, W, n = 1024, 1536, 3
    eng.configure(n, H, W)
    for l in range(n):
        h, wd = H >> l, W >> l
        img = torch.zeros(1, 3, h, wd, device="cuda") if zero else torch.randn(1, 3, h, wd, device="cuda") * 50
        eng.set_targets(l, img, img.clone())
    x = torch.zeros(1, 3, H, W, device="cuda") if zero else torch.randn(1, 3, H, W, device="cuda") * 50
    for _ in range(3):
        eng.closure(x, 1e3, 4e5, 1e2)
    torch.cuda.synchronize()
    eng.set_timing(3)
    eng.timing_totals(0, reset=True)
    for _ in range(8):
        eng.closure(x, 1e3, 4e5, 1e2)
    torch.cuda.synchronize()
    ms, nl, fl = eng.timing_totals(0)
    cms, cn, _ = eng.timing_totals(-1)
    print(f"{'zero' if zero else 'random'} data: closure {cms/cn:.2f} ms, conv3x3 {ms/cn:.2f} ms, {fl/(ms*1e-3)/1e12:.1f} TFLOP/s algorithmic")
    eng.close()

run(False); run(True); run(False)
