"""GPU: a few single-stream L=2 closures, for rocprofv3 --pmc runs."""
import os, sys
os.environ.setdefault("NST_SINGLE_STREAM", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng, x, cfg, _ = bench.build_job(levels, 0, 0)
for _ in range(n):
    eng.closure(x, 1e3, 4e5, 1e2)
torch.cuda.synchronize()
