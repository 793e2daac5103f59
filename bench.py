#!/usr/bin/env python3
"""Benchmark of the pyramid style-transfer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one closure evaluation (the reference's `step` counter, neural_style_transfer.py:198)
of the BASELINE workload - L=2: levels 1536x1024 + 768x512 + 384x256, synthetic 3:2 images,
seeded synthetic VGG19 weights, content+noise init, L-BFGS as the reference constructs it - run
through the optimiser driver, i.e. K closures = K/2 `optimizer.step(closure)` turns, each followed
by the per-step image yield (unprepare + D2H) exactly as NeuralStyleTransfer.process does.
Everything the step reads is resident in HBM before the timed region starts.

N > 1 (launched with torch.distributed.run, one rank per GPU): every rank optimises an independent
content x style job of the same shape (BASELINE config 5, task_executor throughput mode); there is
no data-path collective, `value` is the whole-node closure rate, scaling is weak.

Prints ONE JSON line (rank 0)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

# MI355X_MICROARCH.md dense matrix peaks: fp32 MFMA 157.3 TFLOP/s (v_mfma_f32_32x32x2_f32), bf16 MFMA ~2500 TFLOP/s.
# In the default conv mode every fp32 product is evaluated as 6 bf16 MFMA products (3-piece exact operand split),
# so the matrix pipe executes 6x the algorithmic FLOPs and is priced against the bf16 peak.
MFMA_PEAK = {"f32": 157.3, "bf16x3": 2500.0, "f16x2": 2500.0}
MFMA_WORK_FACTOR = {"f32": 1.0, "bf16x3": 6.0, "f16x2": 3.0}
MFMA_KERNEL = {"f32": "conv_mfma_kernel", "bf16x3": "conv_bf3_kernel", "f16x2": "conv_h2_kernel"}
MFMA_DTYPE = {"f32": "f32",
              "bf16x3": "bf16 (3 exact pieces per fp32 operand, 6 MFMAs per product, fp32 accumulate)",
              "f16x2": "f16 (2 scaled pieces per fp32 operand = 22 significand bits, 3 MFMAs per product, main and "
                       "2^-11-weighted cross terms in separate fp32 accumulators)"}
DTYPE = {"f32": "f32", "bf16x3": "f32 via bf16x3 split (6 bf16 MFMAs per product, f32 accumulate)",
         "f16x2": "f32 via f16x2 split (3 f16 MFMAs per product, f32 accumulate)"}


def pmc_traffic_per_launch(kernel_prefix):
    """HBM GB per launch of the dominant kernel from the committed PMC passes (profiles/r01_pmc_hbm_traffic.json,
    written by tools/summarize_pmc.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of the same L=2
    closure; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when the file is absent or
    the workload is not the L=2 one it was collected on."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
    try:
        kernels = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    n = b = 0
    for name, e in kernels.items():
        if name.startswith(kernel_prefix):
            n += e["launches"]
            b += e["fetch_bytes_corrected"] + e["write_bytes"]
    return (b / n / 1e9) if n else None


def mfma_roofline(mode, algorithmic_flops, ms, launches, extra=None):
    alg = algorithmic_flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    hw = alg * MFMA_WORK_FACTOR[mode]
    out = {"bound": "mfma", "kernel": MFMA_KERNEL[mode],
           "what": "3x3 conv forward + input gradient (+ fused Gram backward)",
           "mfma_dtype": MFMA_DTYPE[mode],
           "achieved": hw, "peak": MFMA_PEAK[mode], "unit": "TFLOP/s", "frac": hw / MFMA_PEAK[mode],
           "algorithmic_tflops": alg, "mfma_work_factor": MFMA_WORK_FACTOR[mode],
           "frac_algorithmic_of_fp32_mfma_peak": alg / MFMA_PEAK["f32"],
           # tools/micro/mfma_power.hip on this pool, random fp16 operands: MFMA-only kernel 1 680 TFLOP/s (zeros: 2 460),
           # with the conv kernel's LDS fragment traffic 1 560
           "frac_of_measured_random_data_mfma_ceiling": (hw / 1680.0) if mode != "f32" else None,
           "frac_of_measured_mfma_plus_lds_ceiling": (hw / 1560.0) if mode == "f16x2" else None,
           "traffic": None, "launches": launches, "avg_launch_ms": ms / max(launches, 1),
           "flops_per_launch_avg": algorithmic_flops / max(launches, 1)}
    if extra:
        out.update(extra)
    return out


def build_job(levels_num: int, seed_shift: int, device, **engine_options):
    """Synthetic L = levels_num-1 job, set up the way neural_style_transfer() does it: pyramid and structured-noise
    initial image on the device (device_image.py), targets through nst_level_set_targets."""
    from artstyletransfer_amd import device_image, synthetic
    from artstyletransfer_amd.config import Config
    from artstyletransfer_amd.engine import StyleEngine

    base_h, base_w = 256, 384
    top = levels_num - 1
    H, W = base_h << top, base_w << top
    content = synthetic.image(H, W, seed=1 + 2 * seed_shift)
    style = synthetic.image(H, W, seed=2 + 2 * seed_shift)
    cfg = Config(levels_num=levels_num)
    weights = synthetic.vgg19_weights()
    eng = StyleEngine(weights, device, **engine_options)
    t0 = time.perf_counter()
    cd, sd = device_image.upload(eng, content), device_image.upload(eng, style)
    content_levels = device_image.pyramid(eng, cd, levels_num)
    style_levels = device_image.pyramid(eng, sd, levels_num)
    np.random.seed(0)
    init, _ = device_image.initial_image(eng, cfg.init_method, cd, sd, content_levels[0], style_levels[0], top,
                                         cfg.noise_factor, cfg.noise_levels, cfg.noise_levels_central_amplitude,
                                         cfg.noise_levels_peripheral_amplitude, cfg.noise_levels_dispersion)
    eng.configure(levels_num, H, W)
    for l in range(levels_num):
        eng.set_targets(l, eng.prepare_img(content_levels[l]), eng.prepare_img(style_levels[l]))
    x = eng.prepare_img(init)
    torch.cuda.synchronize()
    setup_ms = (time.perf_counter() - t0) * 1e3
    host = ([t.cpu().numpy() for t in content_levels], [t.cpu().numpy() for t in style_levels], init.cpu().numpy(), weights)
    cfg.job_setup_ms = setup_ms
    return eng, x, cfg, host


def cpu_baseline(job_host, cfg, closures: int):
    """The oracle (CPU restatement of the reference, validated against it in tests/) timed on this box's
    host cores on a bounded sample of the same workload, run as the reference runs (anomaly mode, the
    zero-weighted randn per level)."""
    from oracle import cpu_ref
    content_levels, style_levels, init, weights = job_host
    # a 1-GPU box owns a 16-core share of a much larger host: more threads than that only thrash
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = int(os.environ.get("NST_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    done = 0
    for _img, step in cpu_ref.run_process(content_levels, style_levels, init, weights, cfg.optimizer, closures,
                                          cfg.content_weight, cfg.style_weight, cfg.tv_weight, as_reference=True):
        done = step
    total = time.perf_counter() - t0
    return {"value": done / total, "unit": "iters/s", "cores": threads, "kind": "port",
            "sample": f"{done} closure evaluations of the same L={len(content_levels) - 1} job (the 2 target "
                      f"forwards per level included), oracle run as the reference runs (anomaly mode on)",
            "seconds": round(total, 2)}


def one_stream_pass(args, cfg, closures: int = 6):
    """Per-kernel durations with every level serialised on one stream (a second engine on the same workload)."""
    os.environ["NST_SINGLE_STREAM"] = "1"
    try:
        eng, x, _, _ = build_job(args.levels, 0, torch.cuda.current_device())
    finally:
        del os.environ["NST_SINGLE_STREAM"]
    cw, sw, tvw = cfg.content_weight, cfg.style_weight, cfg.tv_weight
    for _ in range(2):
        eng.closure(x, cw, sw, tvw)
    torch.cuda.synchronize()
    eng.set_timing(2)
    eng.timing_totals(0, reset=True)
    for _ in range(closures):
        eng.closure(x, cw, sw, tvw)
    torch.cuda.synchronize()
    ms, n, fl = eng.timing_totals(0)
    cms, cn, _ = eng.timing_totals(-1)
    res = mfma_roofline(eng.conv_mode(), fl, ms, n, {"closure_ms": cms / max(cn, 1),
                                                     "conv3x3_ms_per_closure": ms / max(cn, 1)})
    eng.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="closure evaluations in the timed region (even)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--levels", type=int, default=3, help="levels_num (3 = BASELINE L=2)")
    ap.add_argument("--optimizer", default="lbfgs", choices=["lbfgs", "adam"])
    ap.add_argument("--no-yield", action="store_true", help="skip the per-step image yield (unprepare + D2H)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-closures", type=int, default=2)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-all-kernels", action="store_true", help="event pairs around every launch (slower)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) or gloo (rehearsals)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--jobs-per-gpu", type=int, default=1,
                    help="mode 'jobs' only: that many independent jobs per GPU, each on its own HIP stream and host "
                         "thread (the scheduler's simultaneous_tasks_count; 2 gives ~1.09x the aggregate rate). The "
                         "default, 1, is the configuration BASELINE quotes")
    ap.add_argument("--mode", default="jobs", choices=["jobs", "levels", "stripes"],
                    help="N>1: 'jobs' = one independent job per GPU (weak scaling, no collective); 'levels' = ONE "
                         "job, pyramid levels sharded over the ranks, RCCL all-reduce of the pixel gradient per "
                         "closure (BASELINE config 4, strong scaling, capped at 1.33x by the 75/19/5/1 % split)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    from artstyletransfer_amd.engine import PixelOptimizer
    sharded = world > 1 and args.mode in ("levels", "stripes")
    eng, x, cfg, job_host = build_job(args.levels, 0 if sharded else rank, local_rank)
    cfg.optimizer = args.optimizer
    opt = PixelOptimizer(eng, args.optimizer, 10.0, 1)
    if sharded and args.mode == "levels":
        opt.shard_levels(rank, world, dist)
    elif sharded:
        # the top level cut into horizontal stripes (+ halo), the lower levels dealt out by level
        prep = lambda a: eng.prepare_img(torch.from_numpy(a).to(x.device))
        opt.shard_stripes(rank, world, job_host[3], prep(job_host[0][0]), prep(job_host[1][0]), dist)
    cw, sw, tvw = cfg.content_weight, cfg.style_weight, cfg.tv_weight
    H, W = eng.shape
    per_step = 2 if args.optimizer == "lbfgs" else 1

    class JobLoop:
        """One job's optimiser loop with the per-step image yield as NeuralStyleTransfer.process does it: un-prepare on
        the device, D2H into pinned memory on a side stream, so the copy of step k runs under the closures of step
        k+1; it is awaited before the next yield (two host buffers)."""

        def __init__(self, eng, x, opt, stream=None):
            self.eng, self.x, self.opt, self.stream = eng, x, opt, stream
            self.img_host = [torch.empty((H, W, 3), dtype=torch.float32, pin_memory=True) for _ in range(2)]
            self.copy_stream = torch.cuda.Stream()
            self.copy_done = [None, None]
            self.result = (0, None)

        def run(self, closures: int):
            if self.stream is not None:                      # an extra job: its own stream (per-thread current stream)
                with torch.cuda.stream(self.stream):
                    self.result = self._run(closures)
            else:
                self.result = self._run(closures)
            return self.result

        def _run(self, closures: int):
            done = 0
            last = None
            k = 0
            while done < closures:
                info, rows = self.opt.step(self.x, cw, sw, tvw, want_losses=True)
                done += info.closures
                last = rows
                if not args.no_yield:
                    snap = self.eng.unprepare_img(self.x)
                    ready = torch.cuda.Event()
                    ready.record()
                    if self.copy_done[k] is not None:
                        self.copy_done[k].synchronize()           # the consumer is done with this host buffer
                    with torch.cuda.stream(self.copy_stream):
                        self.copy_stream.wait_event(ready)
                        self.img_host[k].copy_(snap, non_blocking=True)
                        snap.record_stream(self.copy_stream)
                        ev = torch.cuda.Event()
                        ev.record()
                    self.copy_done[k] = ev
                    k ^= 1
            for ev in self.copy_done:
                if ev is not None:
                    ev.synchronize()
            return done, last

    jobs = [JobLoop(eng, x, opt)]
    extra = []
    if args.jobs_per_gpu > 1:
        if sharded:
            raise SystemExit("--jobs-per-gpu applies to --mode jobs")
        for j in range(1, args.jobs_per_gpu):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                e2, x2, _, _ = build_job(args.levels, world * j + rank, local_rank)
                o2 = PixelOptimizer(e2, args.optimizer, 10.0, 1)
            extra.append((e2, o2))
            jobs.append(JobLoop(e2, x2, o2, st))
        torch.cuda.synchronize()

    def run(closures: int):
        """All jobs of this rank, each for `closures` closure evaluations; returns (closures done by all, last rows of job 0)."""
        if len(jobs) == 1:
            return jobs[0].run(closures)
        import threading
        threads = [threading.Thread(target=j.run, args=(closures,)) for j in jobs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        return sum(j.result[0] for j in jobs), jobs[0].result[1]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    steps = max(per_step, (args.steps // per_step) * per_step)
    # part of the set-up, not of the --warmup steps: one closure evaluation that leaves x and the optimiser untouched, so
    # that the one-off cold launch of the backward kernels (code-object load, ~17 ms) never lands in a timed region
    for j in jobs:
        j.eng.closure(j.x, cw, sw, tvw)
    torch.cuda.synchronize()
    run(max(args.warmup, 0))
    if args.jobs_per_gpu > 1:
        args.no_kernel_timing = True       # launch durations are not a kernel's own while another job shares the chip
    if not args.no_kernel_timing:
        # HIP events around the dominant kernel's launches (24 per closure) in every fourth closure of the timed region:
        # a pair around each of them in every closure costs 5 % of the closure rate, around all ~50 launches 8 %
        eng.set_timing(4 if not args.time_all_kernels else 2)
        eng.timing_totals(0, reset=True)
    barrier()
    t0 = time.perf_counter()
    done, last_rows = run(steps)
    barrier()
    dt = time.perf_counter() - t0
    from artstyletransfer_amd import sharding
    total_done, dt = sharding.aggregate_throughput(done, dt, dist, device="cuda")
    if sharded:
        total_done = done            # every rank counted the same closures of the one shared job

    if rank == 0:
        px = sum((H >> l) * (W >> l) for l in range(args.levels))
        closure_flops = 1514240.0 * px                        # conv fwd+dgrad + Gram fwd+bwd
        out = {
            "metric": "style-transfer iters/sec at L=2 (1024-px)" if args.levels == 3 else f"style-transfer iters/sec at L={args.levels - 1}",
            "value": total_done / dt,
            "unit": "iters/s",
            "n_gpus": world,
            "steps": done,
            "warmup": args.warmup,
            "ms_per_step": dt / done * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": DTYPE[eng.conv_mode()],
            "data": "synthetic",
            "config": {"workload": f"pyramid style transfer, levels_num={args.levels} "
                                   f"({'+'.join(f'{W >> l}x{H >> l}' for l in range(args.levels))}), "
                                   f"{args.optimizer} as the reference constructs it, content+noise init, "
                                   f"seeded synthetic VGG19 weights, per-step image yield "
                                   f"{'off' if args.no_yield else 'on'}",
                       "iter": "one closure evaluation (forward + losses + backward of every level) + its share of the optimiser update",
                       "parallelism": (("1 GPU" if args.jobs_per_gpu == 1 else f"1 GPU, {args.jobs_per_gpu} jobs on their own streams") if world == 1 else
                                       (f"levels sharded over {world} ranks, RCCL all-reduce of the pixel gradient"
                                        if args.mode == "levels" else
                                        f"top level in {world} stripes (+96-row halo), lower levels by level; all-reduce of the "
                                        f"Gram/content/TV sums and of the pixel gradient per closure")
                                       if sharded else (f"{args.jobs_per_gpu} job(s) per GPU on their own streams, no collective" if args.jobs_per_gpu > 1 else "1 job per GPU, no collective")),
                       "final_loss": float(last_rows[-1][-1]) if last_rows is not None else None,
                       "job_setup_ms_on_device": round(getattr(cfg, "job_setup_ms", 0.0), 1)},
            "closure_tflops_algorithmic": closure_flops / 1e12,
            "closure_rate_tflops": closure_flops * (done / dt) / 1e12,
            # whole-closure algorithmic rate over the fp32 matrix peak (the arithmetic the reference's
            # torch path would need on this chip); > 1 because the products run as fp16 pieces on the 16-bit pipe
            "closure_frac_of_fp32_mfma_peak": closure_flops * (done / dt) / 1e12 / MFMA_PEAK["f32"],
        }
        if not args.no_kernel_timing:
            ms, n, fl = eng.timing_totals(0)
            cms, cn, _ = eng.timing_totals(-1)
            gms, gn, gfl = eng.timing_totals(1)
            oms, on, _ = eng.timing_totals(3)
            c1ms, c1n, c1fl = eng.timing_totals(2)
            out["roofline"] = mfma_roofline(eng.conv_mode(), fl, ms, n)
            if args.levels == 3 and eng.conv_mode() == "f16x2" and not sharded:
                gb = pmc_traffic_per_launch("conv_h2")
                if gb is not None:
                    out["roofline"]["traffic"] = gb
                    out["roofline"]["traffic_unit"] = "GB of HBM per launch (rocprofv3 PMC passes, profiles/r01_pmc_hbm_traffic.json)"
            _, sampled, _ = eng.timing_totals(-2)
            out["roofline"]["sampled_closures"] = sampled
            out["kernel_ms_per_closure"] = {"closure": cms / max(cn, 1), "conv3x3_mfma": ms / max(sampled, 1)}
            if args.time_all_kernels:
                out["kernel_ms_per_closure"].update({"gram_mfma": gms / max(cn, 1), "conv1_1": c1ms / max(cn, 1),
                                                     "streaming": oms / max(cn, 1)})
            if world == 1 and os.environ.get("NST_BATCH") == "0" and not os.environ.get("NST_SINGLE_STREAM"):
                # Under the NST_BATCH=0 schedule the pyramid levels run on separate HIP streams, so the launch durations
                # above are taken while kernels of other levels share the CUs (their sum exceeds the closure time).
                # The same closures re-run on ONE stream give each kernel's duration with the chip to itself.
                out["roofline"]["note"] = ("launch durations measured while kernels of the other pyramid levels run "
                                           "concurrently on their own streams; roofline_one_stream = same closures "
                                           "serialised on one stream")
                out["roofline_one_stream"] = one_stream_pass(args, cfg)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(job_host, cfg, args.cpu_closures)
        print(json.dumps(out), flush=True)
    for e2, o2 in extra:
        o2.close()
        e2.close()
    opt.close()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
