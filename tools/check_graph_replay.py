"""GPU check of the opt-in hipGraph replay (NST_GRAPH): plain gradient descent on the pixel buffer with NO host
synchronisation between closures and updates, eager (0) against the replay (1).  The per-closure totals and
the final image must be bit-identical to the eager run.   python tools/check_graph_replay.py 0 1"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
runs = {}
for mode in sys.argv[1:] or ["0", "1"]:
    os.environ["NST_GRAPH"] = mode
    eng, x, cfg, _ = bench.build_job(3, 0, 0)
    g = torch.empty_like(x); l = torch.empty(13, device="cuda")
    totals = []
    for i in range(24):
        eng.closure_levels(x, cfg.content_weight, cfg.style_weight, cfg.tv_weight, 0xFFFFFFFF, g, l)
        totals.append(l[-1:].clone())
        x.add_(g / g.abs().max(), alpha=-0.5)              # torch kernels on the same stream, no sync
    torch.cuda.synchronize()
    runs[mode] = ([float(t) for t in totals], x.clone())
    print("NST_GRAPH", mode, "totals", runs[mode][0][:3], "...", runs[mode][0][-2:], flush=True)
    del eng
first = next(iter(runs.values()))
for mode, (t, xx) in runs.items():
    print("NST_GRAPH", mode, "identical to", next(iter(runs)), ":", t == first[0] and bool(torch.equal(xx, first[1])))
