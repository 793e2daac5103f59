import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artstyletransfer_amd import synthetic
from artstyletransfer_amd.engine import StyleEngine
w = synthetic.vgg19_weights()
for wino in (False, True, False, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = StyleEngine(w, 0, h2_winograd=wino)
    torch.cuda.synchronize()
    print("h2_winograd", wino, "context creation", round((time.perf_counter() - t0) * 1e3, 1), "ms", flush=True)
    e.close()
