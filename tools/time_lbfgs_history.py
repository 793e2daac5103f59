"""GPU: cost of the L-BFGS update as its history fills up.  The bench job rejects every step after the first (its
history stays empty), so this drives the same L=2 job with the line search older torch builds ran (max_eval = 26:
steps are accepted, the history grows to 100 pairs) and prints, per window of steps, the wall time per step, the
closures per step and the time per step that is NOT closure evaluation (two-loop recursion, line-search algebra,
read-backs).   python tools/time_lbfgs_history.py [steps=130] [lr=1.0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from artstyletransfer_amd.engine import PixelOptimizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 130
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
eng, x, cfg, _ = bench.build_job(3, 0, 0)
opt = PixelOptimizer(eng, "lbfgs", lr, 26)
cw, sw, tvw = cfg.content_weight, cfg.style_weight, cfg.tv_weight
# closure cost alone
g = torch.empty_like(x); l = torch.empty(13, device="cuda")
for _ in range(3): eng.closure_levels(x, cw, sw, tvw, 0xFFFFFFFF, g, l)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): eng.closure_levels(x, cw, sw, tvw, 0xFFFFFFFF, g, l)
torch.cuda.synchronize(); closure_ms = (time.perf_counter() - t0) * 100
print(f"closure alone: {closure_ms:.2f} ms")
win = 10
acc = 0
for w0 in range(0, steps, win):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ncl = 0; nacc = 0
    for _ in range(win):
        info, rows = opt.step(x, cw, sw, tvw)
        ncl += info.closures; nacc += int(info.accepted)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    acc += nacc
    print(f"steps {w0:4d}-{w0 + win - 1:4d}: {dt / win:7.2f} ms/step, {ncl / win:4.1f} closures/step, accepted so far {acc:3d}, "
          f"non-closure {dt / win - ncl / win * closure_ms:6.2f} ms/step, loss {float(info.loss):.4e}", flush=True)
