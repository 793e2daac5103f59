"""Does any kernel of a job read device memory that nobody wrote?  Runs one small two-level L-BFGS job in a child process
with NST_POISON_ALLOC (csrc/nst_api.cpp: chosen allocations are filled with 0xFF bytes = NaN as floats) and compares its
loss rows and final image with an unpoisoned run.  With `--bisect` the range of poisoned allocation numbers is halved
until one allocation is left.  GPU only:  python tools/check_uninit_reads.py [--bisect] [--optimizer lbfgs|adam]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, %(root)r)
from artstyletransfer_amd import synthetic
from artstyletransfer_amd.engine import PixelOptimizer, StyleEngine

def levels(h, w, n, seed):
    top = synthetic.image(h, w, seed) * 255.0
    out, t = [top], torch.from_numpy(top).permute(2, 0, 1).unsqueeze(0)
    for l in range(1, n):
        d = F.interpolate(t, size=(h >> l, w >> l), mode="bicubic", align_corners=False)
        out.append(d.squeeze(0).permute(1, 2, 0).contiguous().numpy())
    return out

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
e = StyleEngine(synthetic.vgg19_weights(bias_std=2.0), 0)
c, s = levels(%(h)d, %(w)d, %(n)d, 1), levels(%(h)d, %(w)d, %(n)d, 2)
e.configure(%(n)d, %(h)d, %(w)d)
for i in range(%(n)d):
    e.set_targets(i, e.prepare_img(dev(c[i])), e.prepare_img(dev(s[i])))
x = e.prepare_img(dev(c[0])).clone()
opt = PixelOptimizer(e, "lbfgs", 1.0, 26) if %(opt)r == "lbfgs" else PixelOptimizer(e, "adam")
rows = []
for _ in range(%(steps)d):
    info, r = opt.step(x, 1e3, 4e5, 1e2)
    rows.append(np.asarray(r, np.float32).copy())
torch.cuda.synchronize()
np.save(sys.argv[1], np.concatenate([np.concatenate(rows).reshape(-1), x.cpu().numpy().reshape(-1)[:4096]]))
'''


def run(spec, args, out):
    env = dict(os.environ)
    env.pop("NST_POISON_ALLOC", None)
    if spec:
        env["NST_POISON_ALLOC"] = spec
    code = CHILD % dict(root=ROOT, h=args.h, w=args.w, n=args.levels, opt=args.optimizer, steps=args.steps)
    p = subprocess.run([sys.executable, "-c", code, out], env=env, capture_output=True, text=True)
    if p.returncode:
        print(p.stderr[-2000:])
        raise SystemExit("child failed")
    import numpy as np
    return np.load(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bisect", action="store_true")
    ap.add_argument("--optimizer", default="lbfgs")
    ap.add_argument("--h", type=int, default=256)
    ap.add_argument("--w", type=int, default=384)
    ap.add_argument("--levels", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--last", type=int, default=400, help="highest allocation number to consider")
    args = ap.parse_args()
    import numpy as np
    clean = run(None, args, "/tmp/uninit_clean.npy")
    same = lambda a: a.shape == clean.shape and np.array_equal(a, clean, equal_nan=False)
    allp = run("all", args, "/tmp/uninit_all.npy")
    print("all allocations poisoned:", "results unchanged" if same(allp) else "RESULTS DIFFER (some kernel reads memory nobody wrote)")
    if same(allp) or not args.bisect:
        return 0 if same(allp) else 1
    lo, hi = 0, args.last
    while lo < hi:
        mid = (lo + hi) // 2
        r = run(f"{lo}-{mid}", args, "/tmp/uninit_b.npy")
        print(f"  allocations {lo}-{mid}: {'clean' if same(r) else 'differs'}")
        if same(r):
            lo = mid + 1
        else:
            hi = mid
    print("first allocation whose poison changes the results:", lo, "(NST_POISON_TRACE=1 prints the sizes in allocation order)")
    return 1


if __name__ == "__main__":
    sys.exit(main())
