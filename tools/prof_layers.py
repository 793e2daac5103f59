"""GPU: per-launch timing of one L=2 closure (single stream) -> stderr."""
import os, sys
os.environ.setdefault("NST_SINGLE_STREAM", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eng, x, cfg, _ = bench.build_job(levels, 0, 0)
for _ in range(3):
    eng.closure(x, 1e3, 4e5, 1e2)
torch.cuda.synchronize()
eng.set_timing(2)
g, l = eng.closure(x, 1e3, 4e5, 1e2)
torch.cuda.synchronize()
eng.lib.nst_dump_last_closure(eng.ctx)
print("closure ms", eng.last_closure_ms())
