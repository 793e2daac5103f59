import sys, torch
sys.path.insert(0, ".")
from artstyletransfer_amd import synthetic
from artstyletransfer_amd.engine import StyleEngine, PixelOptimizer
w = synthetic.vgg19_weights()
torch.cuda.init()
x = torch.zeros(1, 3, 256, 384, device="cuda:0")
base = None
for i in range(12):
    e = StyleEngine(w, 0)
    e.configure(2, 256, 384)
    for l in range(2):
        t = torch.rand(1, 3, 256 >> l, 384 >> l, device="cuda:0")
        e.set_targets(l, t, t)
    o = PixelOptimizer(e, "lbfgs" if i % 2 == 0 else "adam", 10.0, 1)
    for k in range(3):
        o.step(x, 1e3, 4e5, 1e2)
    o.close(); e.close()
    torch.cuda.synchronize()
    free, _ = torch.cuda.mem_get_info()
    if base is None: base = free
    print(i, (base - free) / 2**20, "MiB since first iteration", flush=True)
