"""GPU: what plain streaming kernels reach on this board (torch's fill / copy / read-reduce), as the yardstick for the
HBM-bound kernels of the closure (conv1_1, Gram, pixel kernels)."""
import torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for mb in (100, 403, 1600):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    w = t(lambda: x.fill_(1.0)); c = t(lambda: y.copy_(x)); r = t(lambda: x.sum())
    print(f"{mb} MB: fill {mb / w / 1e3:.2f} TB/s written; copy {2 * mb / c / 1e3:.2f} TB/s read+written; sum {mb / r / 1e3:.2f} TB/s read")
