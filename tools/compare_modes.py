"""GPU: the same L=2 (or --levels N) closure under the three conv modes; losses and gradient differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {}
for mode in ("f32", "bf16x3", "f16x2"):
    os.environ["NST_CONV"] = mode
    eng, x, cfg, _ = bench.build_job(levels, 0, 0)
    assert eng.conv_mode() == mode
    g, l = eng.closure(x, 1e3, 4e5, 1e2)
    torch.cuda.synchronize()
    res[mode] = (g.double().cpu(), l.double().cpu())
    print(mode, "losses", [f"{v:.6e}" for v in l.cpu().tolist()], "finite", bool(torch.isfinite(g).all()))
    eng.close()
ref_g, ref_l = res["f32"]
for mode in ("bf16x3", "f16x2"):
    g, l = res[mode]
    print(f"{mode} vs f32: total rel {abs(float(l[-1] - ref_l[-1])) / float(ref_l[-1]):.2e}, max row rel "
          f"{float(((l - ref_l).abs() / ref_l.abs().clamp_min(1e-30)).max()):.2e}, grad rel-L2 {float((g - ref_g).norm() / ref_g.norm()):.2e}")
g1, g2 = res["bf16x3"][0], res["f16x2"][0]
print(f"f16x2 vs bf16x3: grad rel-L2 {float((g2 - g1).norm() / g1.norm()):.2e}")
