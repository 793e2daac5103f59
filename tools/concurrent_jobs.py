"""GPU: J style-transfer jobs sharing ONE MI355X through the public scheduler (Executor + GpuSlots, the reference's
`simultaneous_tasks_count` jobs per GPU): aggregate closure rate in the window in which all J jobs are stepping.
   NST_SYNTHETIC_WEIGHTS=1 python tools/concurrent_jobs.py [iters=400] [J ...=1 2 3]      (or NST_VGG19_WEIGHTS=<checkpoint>)
Every yielded image is checked for finiteness; a failed job is re-raised by Executor.wait_all."""
import asyncio, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from artstyletransfer_amd import synthetic
from artstyletransfer_amd.config import Config
from artstyletransfer_amd.neural_style_transfer import ContentStylePair
from artstyletransfer_amd.task_executor import Executor, GpuSlots

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
Js = [int(a) for a in sys.argv[2:]] or [1, 2, 3]
H, W = 1024, 1536


async def serve(J):
    cfg = Config(levels_num=3, iters_num=iters, optimizer="lbfgs")
    marks = {}

    async def report(task_id, result):
        pct, img = result
        assert np.isfinite(img).all()
        marks.setdefault(task_id, []).append((time.perf_counter(), pct / 100.0 * iters))

    ex = Executor(cfg, report, gpu_slots=GpuSlots(per_gpu=J, n_gpus=1))
    for j in range(J):
        await ex.add_task(f"job{j}", ContentStylePair(("c", synthetic.image(H, W, seed=1 + 2 * j)),
                                                      ("s", synthetic.image(H, W, seed=2 + 2 * j))))
    await ex.wait_all()
    t_lo = max(m[2][0] for m in marks.values())            # every job past its first steps
    t_hi = min(m[-1][0] for m in marks.values())
    done = 0.0
    for m in marks.values():
        inside = [(t, c) for t, c in m if t_lo <= t <= t_hi]
        done += inside[-1][1] - inside[0][1]
    return done / (t_hi - t_lo), t_hi - t_lo


async def main():
    torch.cuda.init()
    base = None
    for J in Js:
        rate, window = await serve(J)
        base = base or rate
        print(f"{J} job(s) on one GPU: {rate:7.2f} closures/s aggregate over a {window:.1f} s window "
              f"({rate / base:.3f}x of the first line), {rate / J:.2f} per job", flush=True)

asyncio.run(main())
