"""GPU: job set-up latency (pyramid + structured-noise init) on the host (host_image.py) vs on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from artstyletransfer_amd import config, device_image, host_image, synthetic
from artstyletransfer_amd.engine import StyleEngine
cfg = config.Config(levels_num=3)
H, W = 1024, 1536
content, style = synthetic.image(H, W, 1), synthetic.image(H, W, 2)
args = (cfg.noise_factor, cfg.noise_levels, cfg.noise_levels_central_amplitude, cfg.noise_levels_peripheral_amplitude, cfg.noise_levels_dispersion)
eng = StyleEngine(synthetic.vgg19_weights(), 0)
for rep in range(2):
    t0 = time.perf_counter()
    cl = [host_image.resize_to_level(content, l) for l in (2, 1, 0)]
    sl = [host_image.resize_to_level(style, l) for l in (2, 1, 0)]
    np.random.seed(0)
    init_h, _ = host_image.initial_image("content+noise", content, style, cl[0], sl[0], 2, *args)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cd, sd = device_image.upload(eng, content), device_image.upload(eng, style)
    cdl, sdl = device_image.pyramid(eng, cd, 3), device_image.pyramid(eng, sd, 3)
    np.random.seed(0)
    init_d, _ = device_image.initial_image(eng, "content+noise", cd, sd, cdl[0], sdl[0], 2, *args)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    print(f"L=2 job set-up: host {t_host*1e3:.0f} ms, device {t_dev*1e3:.1f} ms (incl. 2 x 18.9 MB H2D), max |init diff| {np.abs(init_d.cpu().numpy()-init_h).max():.2e}")
