"""GPU: per-launch durations of one L=2 closure (HIP events around every launch) for engine option sets given on the
command line as key=value groups separated by '/':   python tools/layer_times.py h2_mfma16=0 / h2_mfma16=1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

groups, cur = [], {}
for a in sys.argv[1:]:
    if a == "/":
        groups.append(cur); cur = {}
    else:
        k, v = a.split("=")
        cur[k] = v if k == "conv_mode" else int(v)
groups.append(cur)
levels = int(os.environ.get("LEVELS", "3"))
reps = int(os.environ.get("REPS", "30"))          # closures of the untimed rate (the chip is power-managed: use >= 200 for 1 % decisions)
for opts in groups:
    eng, x, cfg, _ = bench.build_job(levels, 0, 0, **opts)
    cw, sw, tvw = cfg.content_weight, cfg.style_weight, cfg.tv_weight
    for _ in range(5):
        eng.closure(x, cw, sw, tvw)
    torch.cuda.synchronize()
    # untimed rate first (events cost a few %)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        eng.closure(x, cw, sw, tvw)
    t1.record(); torch.cuda.synchronize()
    plain = t0.elapsed_time(t1) / reps
    eng.set_timing(2)
    eng.timing_totals(0, reset=True)
    for _ in range(8):
        eng.closure(x, cw, sw, tvw)
    torch.cuda.synchronize()
    sys.stderr.flush()
    eng.lib.nst_dump_last_closure(eng.ctx)          # (before the totals: reading them folds the event list away)
    tot = {c: eng.timing_totals(c) for c in (0, 1, 2, 3, -1)}
    n = tot[-1][1]
    print(f"== {opts}: closure {plain:.3f} ms untimed; per closure: conv3x3 {tot[0][0] / n:.3f} ms ({tot[0][2] / tot[0][0] / 1e9:.0f} TF alg), "
          f"gram {tot[1][0] / n:.3f}, conv1_1 {tot[2][0] / n:.3f}, other {tot[3][0] / n:.3f}", flush=True)
    eng.close()
