"""CPU only: how much of L-BFGS' FIRST curvature pair on the headline job (config 3 from the reference's own start image) is
signal.  The first step is t d with t = min(1, 1/|g|_1) lr (torch:optim/lbfgs.py:454-457): here at most two ulps of a pixel
value, so x1 - x0 is the rounding of the step and y = g1 - g0 the difference of two gradients whose rounding noise is as large
as their true difference.  Printed: the pair in the oracle's fp32 (= the reference's arithmetic on this host), the same two
gradients evaluated in fp64 at the same two images, and the loss of the second trial point for several torch thread counts
(another summation order of the same library).  Output of the round-3 run: profiles/r03_lbfgs_first_pair_noise.txt.

    python tools/diag_first_pair_noise.py [threads ...]        (a few minutes on 8 cores)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import cpu_ref, cv2_ref

H0, W0, NLEV = 1024, 1536, 3
content = cpu_ref.synthetic_image(H0, W0, seed=1)
style = cpu_ref.synthetic_image(H0, W0, seed=2)
c_lv = [cv2_ref.resize_cubic(content, *cv2_ref.level_size(H0, W0, l)) for l in range(NLEV - 1, -1, -1)]
s_lv = [cv2_ref.resize_cubic(style, *cv2_ref.level_size(H0, W0, l)) for l in range(NLEV - 1, -1, -1)]
np.random.seed(0)
init, _ = cv2_ref.initial_image("content+noise", content, style, c_lv[0], s_lv[0], 2, 0.95, (9, 18, 36, -1, 0),
                                (0.30, 0.20, 0.10, 0.20, 0.20), (0.20, 0.30, 0.40, 0.10, 0.00), (0.20, 0.30, 0.40, 0.60, 0.30))
w = cpu_ref.synthetic_vgg19_weights(bias_std=cpu_ref.TEST_BIAS_STD)
torch.set_num_threads(min(8, os.cpu_count() or 1))
tg = [cpu_ref.LevelTargets(cpu_ref.prepare_img(c), cpu_ref.prepare_img(s), w) for c, s in zip(c_lv, s_lv)]
x0 = cpu_ref.prepare_img(init)
_, g0, _ = cpu_ref.closure_eval(x0, tg, w, 1e3, 4e5, 1e2)
g0f = g0.reshape(-1)
t0 = min(1.0, 1.0 / float(g0f.abs().sum())) * 10.0
s = g0f * (-t0)
x1 = (x0.reshape(-1) + s).view_as(x0)
_, g1, _ = cpu_ref.closure_eval(x1, tg, w, 1e3, 4e5, 1e2)
y = g1.reshape(-1) - g0f
ys, yy = float(y.dot(s)), float(y.dot(y))
print(f"first step: t = {t0:.4e}, largest pixel move {float(s.abs().max()):.2e} against an ulp of {float(np.spacing(np.float32(100.0))):.2e} at "
      f"|x| = 100; {float(((x1 - x0) != 0).float().mean()):.0%} of the pixels moved at all")
print(f"fp32 (the reference's arithmetic on this host): y.s = {ys:.3e}, y.y = {yy:.3e}, H_diag = {ys / yy:.3e}, |g1| / |y| = {float(g1.norm() / y.norm()):.0f}")
w64 = [(a.double(), b.double()) for a, b in w]
tg64 = [cpu_ref.LevelTargets(cpu_ref.prepare_img(c).double(), cpu_ref.prepare_img(s_).double(), w64) for c, s_ in zip(c_lv, s_lv)]
_, g0d, _ = cpu_ref.closure_eval(x0.double(), tg64, w64, 1e3, 4e5, 1e2)
_, g1d, _ = cpu_ref.closure_eval(x1.double(), tg64, w64, 1e3, 4e5, 1e2)
yd, sd = (g1d - g0d).reshape(-1), s.double()
print(f"fp64 at the same two images: y.s = {float(yd.dot(sd)):.3e}, y.y = {float(yd.dot(yd)):.3e}, H_diag = {float(yd.dot(sd) / yd.dot(yd)):.3e}; "
      f"|y(fp32) - y(fp64)| / |y(fp64)| = {float((y.double() - yd).norm() / yd.norm()):.2f}")
for threads in [int(a) for a in sys.argv[1:]] or [8, 5, 3]:
    torch.set_num_threads(threads)
    rec = []
    for _ in cpu_ref.run_process(c_lv, s_lv, init, w, "lbfgs", 4, record=rec):
        pass
    print(f"torch threads {threads}: closure losses " + ", ".join(f"{r['loss']:.7e}" for r in rec))
