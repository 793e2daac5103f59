"""GPU: the public drop-in entry point (neural_style_transfer.neural_style_transfer, the reference's job API) on the
L=2 benchmark job; wall time per closure evaluation including set-up amortisation and the per-step image yield."""
import asyncio, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import neural_style_transfer as nst
from artstyletransfer_amd import synthetic
from artstyletransfer_amd.config import Config

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = 256 << (levels - 1), 384 << (levels - 1)
cfg = Config(levels_num=levels, iters_num=iters)
pair = nst.ContentStylePair(("content", synthetic.image(H, W, seed=1)), ("style", synthetic.image(H, W, seed=2)))


async def main():
    np.random.seed(0)
    t0 = time.perf_counter()
    first = None
    n = 0
    async for pct, img in nst.neural_style_transfer(pair, cfg.content_weight, cfg.style_weight, cfg.tv_weight, cfg.optimizer,
                                                    cfg.model, cfg.init_method, cfg.iters_num, cfg.levels_num, cfg.noise_factor,
                                                    cfg.noise_levels, cfg.noise_levels_central_amplitude,
                                                    cfg.noise_levels_peripheral_amplitude, cfg.noise_levels_dispersion):
        n += 1
        if first is None:
            first = time.perf_counter()
    t1 = time.perf_counter()
    print(f"{iters} closure evaluations ({n} yields of {img.shape} images): total {t1 - t0:.2f} s, "
          f"{iters / (t1 - t0):.1f} it/s including set-up; {(iters - 2) / (t1 - first):.1f} it/s after the first step; "
          f"last percent {pct:.1f}, image range [{img.min():.3f}, {img.max():.3f}]")

asyncio.run(main())
