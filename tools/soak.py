"""GPU: soak test - several jobs in sequence through the public entry point, device memory before / after (leaks),
finiteness of every yielded image."""
import asyncio, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import neural_style_transfer as nst
from artstyletransfer_amd import synthetic
from artstyletransfer_amd.config import Config

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 3


async def one(job, optimizer, levels):
    H, W = 256 << (levels - 1), 384 << (levels - 1)
    cfg = Config(levels_num=levels, iters_num=iters, optimizer=optimizer)
    pair = nst.ContentStylePair(("c", synthetic.image(H, W, seed=10 + job)), ("s", synthetic.image(H - 40, W + 24, seed=20 + job)))
    n = 0
    t0 = time.perf_counter()
    async for pct, img in nst.neural_style_transfer(pair, cfg.content_weight, cfg.style_weight, cfg.tv_weight, cfg.optimizer,
                                                    cfg.model, cfg.init_method, cfg.iters_num, cfg.levels_num, cfg.noise_factor,
                                                    cfg.noise_levels, cfg.noise_levels_central_amplitude,
                                                    cfg.noise_levels_peripheral_amplitude, cfg.noise_levels_dispersion):
        n += 1
        assert np.isfinite(img).all(), (job, n)
    return n, time.perf_counter() - t0, float(img.min()), float(img.max())


async def main():
    torch.cuda.init()
    free0, total = torch.cuda.mem_get_info()
    for j in range(jobs):
        opt, lv = [("lbfgs", 3), ("adam", 3), ("lbfgs", 2), ("adam", 1)][j % 4]
        n, dt, lo, hi = await one(j, opt, lv)
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        print(f"job {j} {opt} levels {lv}: {n} yields in {dt:.1f} s ({iters / dt:.1f} it/s incl. set-up), image range [{lo:.3f}, {hi:.3f}], "
              f"device memory in use beyond start: {(free0 - free) / 2**20:.0f} MiB", flush=True)

asyncio.run(main())
