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

Beside the headline (N = 1 only, after the timed region; none of it is inside `value`):
  exact_f32        the same job on the exact fp32 MFMA (nst_options.conv_mode = f32), priced against the 157.3 TF fp32 peak;
  sustained        the headline job kept running for >= 5 s whatever --steps says, with the shader clock sampled from sysfs;
  direct_convolution   the same job with nst_options.h2_winograd = 0 (no launch in the Winograd form)
  progressing_job  jobs whose image moves at every step (Adam; L-BFGS with the 25-evaluation line search), so that the
                   optimiser update with a filling curvature history is in a driver-seen number;
  cpu_baseline     the oracle timed on this box's host cores on a bounded sample of the same job.

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
MFMA_KERNEL = {"f32": "conv_mfma_kernel", "bf16x3": "conv_bf3_batch_kernel",
               "f16x2": "conv_h2_batch_kernel + conv_wino_batch_kernel (the 14 of 24 launches with Cin >= 256 and no second source as Winograd F(2,3))"}
MFMA_DTYPE = {"f32": "f32",
              "bf16x3": "bf16 (3 exact pieces per fp32 operand, 6 MFMAs per product, fp32 accumulate)",
              "f16x2": "f16 (2 scaled pieces per fp32 operand = 22 significand bits, 3 MFMAs per product, main and "
                       "2^-11-weighted cross terms in separate fp32 accumulators)"}
DTYPE = {"f32": "f32", "bf16x3": "f32 via bf16x3 split (6 bf16 MFMAs per product, f32 accumulate)",
         "f16x2": "f32 via f16x2 split (3 f16 MFMAs per product, f32 accumulate; the 14 large-channel conv launches of a closure as Winograd F(2,3) in that arithmetic)"}


TRAFFIC_PROFILES = ("r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json")


def committed_traffic_per_launch(kernel_prefix):
    """HBM GB per launch of the dominant kernel from the COMMITTED PMC passes (profiles/r0N_pmc_hbm_traffic.json, written
    by tools/summarize_pmc.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of the same L=2 closure;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  It is a replayed figure, not a measurement of this
    run (hardware counters need rocprofv3 around the process), and is reported under that name; roofline.traffic stays
    null.  None when no file is there."""
    for name in TRAFFIC_PROFILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            kernels = json.load(open(path))["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        n = b = 0
        for kname, e in kernels.items():
            if any(kname.startswith(pfx) for pfx in kernel_prefix.split("|")):
                n += e["launches"]
                b += e["fetch_bytes_corrected"] + e["write_bytes"]
        if n:
            return {"gb_per_launch": b / n / 1e9, "file": f"profiles/{name}",
                    "what": "HBM GB per launch of the dominant kernel, rocprofv3 PMC passes of an earlier run of this workload"}
    return None


def measured_traffic(kernel_prefix, levels, timeout_s=150):
    """HBM traffic of the dominant kernel measured in THIS run: two rocprofv3 counter passes (FETCH_SIZE, then WRITE_SIZE - they
    do not fit one pass on gfx950) over a few closures of the same workload in a CHILD process (tools/pmc_run.py), after the
    timed region; summed over the last closure's launches of the kernel and divided by their number.  FETCH_SIZE is doubled as
    MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests of 16-byte-per-lane streams are tallied at 64 B); the counter
    sits at the L2's fabric side, so Infinity-Cache hits are included.  None when rocprofv3 is not there or a pass fails."""
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import summarize_pmc
    except Exception:
        return None
    out = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_run.py"), str(levels), "2"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
                if r.returncode != 0:
                    return None
                rows = summarize_pmc.rows_of(d)
                vals, meta = summarize_pmc.per_dispatch(rows)
                n = b = 0
                for disp in summarize_pmc.last_closure(rows):
                    name = summarize_pmc.short(meta[disp]["Kernel_Name"])
                    if any(name.startswith(pfx) for pfx in kernel_prefix.split("|")):
                        n += 1
                        b += vals[disp].get(counter, 0.0) * 1024.0
                out[counter] = (n, b)
            except (SystemExit, Exception):
                return None
    (nf, fb), (nw, wb) = out["FETCH_SIZE"], out["WRITE_SIZE"]
    if not nf or nf != nw:
        return None
    return {"gb_per_launch": (2.0 * fb + wb) / nf / 1e9, "fetch_gb_corrected": 2.0 * fb / 1e9, "write_gb": wb / 1e9, "launches": nf,
            "what": "HBM (L2 fabric side) GB per launch of the dominant kernel over one closure of this workload: rocprofv3 --pmc "
                    "FETCH_SIZE (x2, gfx950) and --pmc WRITE_SIZE, separate passes in a child process of this run"}


class GpuSampler:
    """Shader clock (and board power) of one GPU sampled from sysfs every 50 ms on a thread: what the sustained figure is
    quoted with.  The conv kernel is power/clock bound, so a rate is only meaningful with its clock."""

    def __init__(self, device_index: int):
        import glob
        import threading
        self.paths = {}
        try:
            props = torch.cuda.get_device_properties(device_index)
            bdf = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
            base = f"/sys/bus/pci/devices/{bdf}"
            for key, pat in (("sclk_hz", "hwmon/hwmon*/freq1_input"), ("power_uw", "hwmon/hwmon*/power1_average"),
                             ("power_uw_in", "hwmon/hwmon*/power1_input")):
                hits = glob.glob(os.path.join(base, pat))
                if hits:
                    self.paths[key] = hits[0]
            self.dpm = os.path.join(base, "pp_dpm_sclk") if os.path.exists(os.path.join(base, "pp_dpm_sclk")) else None
        except Exception:
            self.dpm = None
        self.samples = {"sclk_mhz": [], "power_w": []}
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        try:
            if "sclk_hz" in self.paths:
                self.samples["sclk_mhz"].append(int(open(self.paths["sclk_hz"]).read()) / 1e6)
            elif self.dpm:
                for line in open(self.dpm):
                    if "*" in line:
                        self.samples["sclk_mhz"].append(float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
            for k in ("power_uw", "power_uw_in"):
                if k in self.paths:
                    self.samples["power_w"].append(int(open(self.paths[k]).read()) / 1e6)
                    break
        except Exception:
            pass

    def _run(self):
        while not self._stop.is_set():
            self._read()
            self._stop.wait(0.05)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._thread.join()

    def summary(self):
        out = {}
        for k, v in self.samples.items():
            if v:
                out[k] = {"min": round(min(v), 1), "mean": round(sum(v) / len(v), 1), "max": round(max(v), 1), "samples": len(v)}
        return out or None


def mfma_roofline(mode, algorithmic_flops, ms, launches, extra=None, mfma_flops=None):
    """`mfma_flops`: the executed matrix-pipe FLOPs of those launches (nst_timing_mfma_flops: the arithmetic's MFMAs per
    product, 2/3 of them where a launch ran as Winograd F(2,3)); None = algorithmic x the arithmetic's factor."""
    alg = algorithmic_flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    hw = alg * MFMA_WORK_FACTOR[mode] if mfma_flops is None else (mfma_flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
    out = {"bound": "mfma", "kernel": MFMA_KERNEL[mode],
           "what": "3x3 conv forward + input gradient (+ fused Gram backward)",
           "mfma_dtype": MFMA_DTYPE[mode],
           "achieved": hw, "peak": MFMA_PEAK[mode], "unit": "TFLOP/s", "frac": hw / MFMA_PEAK[mode],
           "algorithmic_tflops": alg, "mfma_work_factor": (hw / alg) if alg > 0 else MFMA_WORK_FACTOR[mode],
           # what the same outputs cost as DIRECT convolutions in this arithmetic (3 MFMAs per product for f16x2): the rate
           # comparable with the records of the builds before the Winograd launches existed
           "direct_equivalent_tflops": alg * MFMA_WORK_FACTOR[mode],
           "direct_equivalent_frac": alg * MFMA_WORK_FACTOR[mode] / MFMA_PEAK[mode],
           # NOT a roofline fraction: how many times the fp32-MFMA peak RATE (157.3 TF) the algorithmic rate is
           "algorithmic_rate_over_fp32_mfma_peak_rate": alg / MFMA_PEAK["f32"],
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


def one_stream_pass(args, cfg, closures: int = 6, **engine_options):
    """Per-kernel durations with every level serialised on one stream (a second engine on the same workload): under the
    per-level schedule the levels run on streams of their own, and a launch's duration is not its own while kernels of
    other levels share the CUs."""
    eng, x, _, _ = build_job(args.levels, 0, torch.cuda.current_device(), batched=False, single_stream=True, **engine_options)
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


class JobLoop:
    """One job's optimiser loop with the per-step image yield as NeuralStyleTransfer.process does it: un-prepare on the
    device, D2H into pinned memory on a side stream, so the copy of step k runs under the closures of step k+1; it is
    awaited before the next yield (two host buffers)."""

    def __init__(self, eng, x, opt, weights3, do_yield=True, stream=None):
        self.eng, self.x, self.opt, self.stream = eng, x, opt, stream
        self.cw, self.sw, self.tvw = weights3
        self.do_yield = do_yield
        H, W = eng.shape
        self.img_host = [torch.empty((H, W, 3), dtype=torch.float32, pin_memory=True) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream()
        self.copy_done = [None, None]
        self.result = (0, None)
        self.opt_steps = 0
        self.accepted = 0
        self.history = 0

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
            info, rows = self.opt.step(self.x, self.cw, self.sw, self.tvw, want_losses=True)
            done += info.closures
            self.opt_steps += 1
            self.accepted += int(info.accepted)
            self.history = int(info.history)
            last = rows
            if self.do_yield:
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


def timed(job: JobLoop, closures: int):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done, rows = job.run(closures)
    torch.cuda.synchronize()
    return done, time.perf_counter() - t0, rows


def side_job(args, optimizer, max_eval, closures, warm, **engine_options):
    """A fresh engine on the same workload with another optimiser / arithmetic, timed like the headline (per-step yield
    included).  Returns (dict, engine still open for its timing totals)."""
    from artstyletransfer_amd.engine import PixelOptimizer
    eng, x, cfg, _ = build_job(args.levels, 0, torch.cuda.current_device(), **engine_options)
    opt = PixelOptimizer(eng, optimizer, 10.0, max_eval)
    job = JobLoop(eng, x, opt, (cfg.content_weight, cfg.style_weight, cfg.tv_weight), not args.no_yield)
    eng.closure(x, cfg.content_weight, cfg.style_weight, cfg.tv_weight)      # cold launches outside the timed region
    job.run(warm)
    first = None
    done, dt, rows = timed(job, closures)
    out = {"value": done / dt, "unit": "iters/s", "ms_per_step": dt / done * 1e3, "steps": done,
           "optimizer": optimizer if optimizer == "adam" else f"lbfgs, max_eval {max_eval}",
           "optimizer_steps": job.opt_steps, "accepted_steps": job.accepted, "history_pairs": job.history,
           "final_loss": float(rows[-1][-1]) if rows is not None else None}
    return out, eng, opt, job


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` started plainly (no launcher, no WORLD_SIZE): run N ranks of this script under
    torch.distributed.run as a child process and return its exit code.  Called before anything touches the GPU - a
    process that has initialised HIP must never exec or fork GPU work."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def ranks_present(dist, device) -> int:
    """How many ranks take part in a collective: every rank contributes 1 to an all-reduce."""
    one = torch.ones(1, dtype=torch.float32, device=device if device is not None else "cpu")
    dist.all_reduce(one)
    return int(round(float(one.item())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="closure evaluations in the timed region (even)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--levels", type=int, default=3, help="levels_num (3 = BASELINE L=2)")
    ap.add_argument("--optimizer", default="lbfgs", choices=["lbfgs", "adam"])
    ap.add_argument("--lbfgs-max-eval", type=int, default=1,
                    help="1 = the reference's constructor arguments under torch 2.10 (one trial point per step, almost always "
                         "rejected: SURVEY F5); 26 = the line search older torch performed")
    ap.add_argument("--no-yield", action="store_true", help="skip the per-step image yield (unprepare + D2H)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip exact_f32 / sustained / progressing_job")
    ap.add_argument("--sustained-seconds", type=float, default=5.0)
    ap.add_argument("--cpu-closures", type=int, default=2)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 counter passes behind roofline.traffic (~40 s)")
    ap.add_argument("--time-all-kernels", action="store_true", help="event pairs around every launch (slower)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) or gloo (rehearsals)")
    ap.add_argument("--comm", default="c-abi", choices=["c-abi", "torch"],
                    help="mode 'levels': the per-closure collective behind the C ABI (nst_comm_*: one ncclAllReduce of the "
                         "packed gradient + loss row) or through torch.distributed (needed for gloo rehearsals)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="rehearsal of the launch path (works without a GPU): start the ranks, count them with one "
                         "collective, print {ranks_seen, n_gpus} and leave before any device call")
    ap.add_argument("--jobs-per-gpu", type=int, default=1,
                    help="mode 'jobs' only: that many independent jobs per GPU, each on its own HIP stream and host "
                         "thread (the scheduler's simultaneous_tasks_count; 2 gives ~1.09x the aggregate rate). The "
                         "default, 1, is the configuration BASELINE quotes")
    ap.add_argument("--stripe-levels", type=int, default=2,
                    help="mode 'stripes': how many of the top levels are cut into stripes (level 0 = 75 %% of the work, "
                         "levels 0-1 = 94 %%); the rest is dealt out by level")
    ap.add_argument("--mode", default="jobs", choices=["jobs", "levels", "stripes"],
                    help="N>1: 'jobs' = one independent job per GPU (weak scaling, no collective); 'levels' = ONE "
                         "job, pyramid levels sharded over the ranks, RCCL all-reduce of the pixel gradient per "
                         "closure (BASELINE config 4, strong scaling, capped at 1.33x by the 75/19/5/1 % split)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started plainly: launch the N ranks ourselves, as CHILD processes of a parent that has not touched the GPU
        # (nothing above initialises HIP), and leave with the launcher's exit code.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU, e.g. python -m "
                         f"torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py "
                         f"--gpus {args.gpus} ... (or run bench.py --gpus {args.gpus} without WORLD_SIZE: it launches them)")
    if args.rendezvous_only:
        # rehearsal of the launch path on a box without GPUs: rendezvous, count the ranks, leave before any device call
        import torch.distributed as dist
        dist.init_process_group(args.dist_backend if world > 1 else "gloo")
        seen = ranks_present(dist, None)
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "ranks_seen": seen,
                              "backend": dist.get_backend()}), flush=True)
        dist.destroy_process_group()
        if seen != args.gpus:
            raise SystemExit(f"{seen} ranks answered, --gpus {args.gpus}")
        return
    if not args.share_gpu and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s)")
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
        seen = ranks_present(dist, torch.device("cuda", local_rank) if args.dist_backend == "nccl" else None)
        if seen != args.gpus:
            raise SystemExit(f"{seen} ranks answered the first collective, --gpus {args.gpus}")

    from artstyletransfer_amd.engine import Communicator, PixelOptimizer
    sharded = world > 1 and args.mode in ("levels", "stripes")
    eng, x, cfg, job_host = build_job(args.levels, 0 if sharded else rank, local_rank)
    cfg.optimizer = args.optimizer
    opt = PixelOptimizer(eng, args.optimizer, 10.0, args.lbfgs_max_eval)
    comm = None
    if sharded and args.comm == "c-abi" and args.dist_backend == "nccl" and not args.share_gpu:
        # the collectives behind the C ABI (nst_comm_*: RCCL resolved by the library itself); the id travels over the process group
        ids = [Communicator.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = Communicator(local_rank, rank, world, ids[0])
    prep = lambda a: eng.prepare_img(torch.from_numpy(a).to(x.device))
    if sharded and args.mode == "levels":
        if comm is not None:
            opt.shard_levels_comm(comm)
        else:
            opt.shard_levels(rank, world, dist)
    elif sharded:
        # the top level cut into horizontal stripes (+ halo), the lower levels dealt out by level
        ns = min(args.stripe_levels, args.levels)
        opt.shard_stripes(rank, world, job_host[3], [prep(job_host[0][l]) for l in range(ns)],
                          [prep(job_host[1][l]) for l in range(ns)], dist, comm=comm)
    w3 = (cfg.content_weight, cfg.style_weight, cfg.tv_weight)
    cw, sw, tvw = w3
    H, W = eng.shape
    per_step = 2 if (args.optimizer == "lbfgs" and args.lbfgs_max_eval == 1) else 1

    jobs = [JobLoop(eng, x, opt, w3, not args.no_yield)]
    extra = []
    if args.jobs_per_gpu > 1:
        if sharded:
            raise SystemExit("--jobs-per-gpu applies to --mode jobs")
        for j in range(1, args.jobs_per_gpu):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                e2, x2, _, _ = build_job(args.levels, world * j + rank, local_rank)
                o2 = PixelOptimizer(e2, args.optimizer, 10.0, args.lbfgs_max_eval)
            extra.append((e2, o2))
            jobs.append(JobLoop(e2, x2, o2, w3, not args.no_yield, st))
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
    for j in jobs:
        j.opt_steps = j.accepted = 0
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
        opt_name = args.optimizer if args.optimizer == "adam" else (
            "lbfgs as the reference constructs it (max_eval 1 under torch 2.10)" if args.lbfgs_max_eval == 1
            else f"lbfgs with max_eval {args.lbfgs_max_eval}")
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
                                   f"{opt_name}, content+noise init, "
                                   f"seeded synthetic VGG19 weights, per-step image yield "
                                   f"{'off' if args.no_yield else 'on'}",
                       "iter": "one closure evaluation (forward + losses + backward of every level) + its share of the optimiser update",
                       "parallelism": (("1 GPU" if args.jobs_per_gpu == 1 else f"1 GPU, {args.jobs_per_gpu} jobs on their own streams") if world == 1 else
                                       (f"levels sharded over {world} ranks, one RCCL all-reduce of the packed pixel gradient + loss row per closure"
                                        + (" behind the C ABI (nst_comm)" if comm is not None else " through torch.distributed")
                                        if args.mode == "levels" else
                                        f"top {min(args.stripe_levels, args.levels)} level(s) in {world} stripes (+96-row halo), lower levels by level; all-reduce of the "
                                        f"Gram/content/TV sums and of the packed pixel gradient + loss row per closure"
                                        + (" behind the C ABI (nst_comm)" if comm is not None else " through torch.distributed"))
                                       if sharded else (f"{args.jobs_per_gpu} job(s) per GPU on their own streams, no collective" if args.jobs_per_gpu > 1 else "1 job per GPU, no collective")),
                       "final_loss": float(last_rows[-1][-1]) if last_rows is not None else None,
                       "job_setup_ms_on_device": round(getattr(cfg, "job_setup_ms", 0.0), 1)},
            # what the optimiser did in the timed region: with the reference's L-BFGS arguments under torch 2.10 almost
            # every trial step is rejected, so the curvature history stays (nearly) empty - see progressing_job
            "optimizer_steps": jobs[0].opt_steps, "accepted_steps": jobs[0].accepted, "lbfgs_history_pairs": jobs[0].history,
            "closure_tflops_algorithmic": closure_flops / 1e12,
            "closure_rate_tflops": closure_flops * (done / dt) / 1e12,
        }
        if comm is not None:
            r_, w_, calls, nbytes = comm.info()
            out["comm"] = {"ranks_seen": w_, "allreduce_calls": calls, "bytes_per_call": nbytes / max(calls, 1)}
        elif dist is not None:
            out["comm"] = {"ranks_seen": seen, "backend": args.dist_backend}
        if not args.no_kernel_timing:
            ms, n, fl = eng.timing_totals(0)
            cms, cn, _ = eng.timing_totals(-1)
            gms, gn, gfl = eng.timing_totals(1)
            oms, on, _ = eng.timing_totals(3)
            c1ms, c1n, c1fl = eng.timing_totals(2)
            out["roofline"] = mfma_roofline(eng.conv_mode(), fl, ms, n, mfma_flops=eng.timing_mfma_flops(0))
            if args.levels == 3 and eng.conv_mode() == "f16x2" and not sharded:
                out["roofline"]["traffic_from_committed_profile"] = committed_traffic_per_launch("conv_h2|conv_wino")
            _, sampled, _ = eng.timing_totals(-2)
            out["roofline"]["sampled_closures"] = sampled
            out["kernel_ms_per_closure"] = {"closure": cms / max(cn, 1), "conv3x3_mfma": ms / max(sampled, 1)}
            if args.time_all_kernels:
                out["kernel_ms_per_closure"].update({"gram_mfma": gms / max(cn, 1), "conv1_1": c1ms / max(cn, 1),
                                                     "streaming": oms / max(cn, 1)})
        extras = world == 1 and args.jobs_per_gpu == 1 and not args.no_extras
        if extras:
            # ---- sustained: the same job goes on for >= 5 s, whatever --steps was; clock sampled from sysfs
            eng.set_timing(0)
            chunk = 20 * per_step
            sdone, st0 = 0, time.perf_counter()
            with GpuSampler(local_rank) as smp:
                while time.perf_counter() - st0 < args.sustained_seconds:
                    d, _ = jobs[0].run(chunk)
                    sdone += d
                torch.cuda.synchronize()
                sdt = time.perf_counter() - st0
            out["sustained"] = {"seconds": round(sdt, 2), "steps": sdone, "value": sdone / sdt, "unit": "iters/s",
                                "ms_per_step": sdt / sdone * 1e3, "gpu": smp.summary()}
            # ---- the exact fp32 MFMA on the same job
            f32, e32, o32, _ = side_job(args, args.optimizer, args.lbfgs_max_eval, 4 * per_step, per_step, conv_mode="f32")
            o32.close()
            e32.close()
            f32["dtype"] = DTYPE["f32"]
            f32["schedule"] = "one launch per layer and level, levels on HIP streams of their own (the f32 kernel has no batched form)"
            f32["roofline"] = one_stream_pass(args, cfg, 3, conv_mode="f32")
            f32["roofline"]["note"] = "launch durations from the same closures serialised on ONE stream (each kernel with the chip to itself)"
            out["exact_f32"] = f32
            # ---- jobs that make progress
            prog = {}
            a, ea, oa, _ = side_job(args, "adam", 1, 40, 4)
            prog["adam"] = a
            oa.close()
            ea.close()
            ls, el, ol, _ = side_job(args, "lbfgs", 26, 60, 10)
            prog["lbfgs_line_search"] = ls
            ol.close()
            el.close()
            out["progressing_job"] = prog
            # ---- the same job with every convolution direct (nst_options.h2_winograd = 0: the build of the rounds before)
            wg, ew, ow, _ = side_job(args, args.optimizer, args.lbfgs_max_eval, 20 * per_step, 2 * per_step, h2_winograd=False)
            ow.close()
            ew.close()
            wg["what"] = ("the same job with nst_options.h2_winograd = 0: all 24 conv launches as direct convolutions "
                          "(3 f16 MFMAs per product everywhere)")
            out["direct_convolution"] = wg
        if extras and "roofline" in out and not args.no_traffic and eng.conv_mode() == "f16x2":
            # roofline.traffic: measured, not replayed - two counter passes in a child process (this process keeps its engines)
            tr = measured_traffic("conv_h2_batch|conv_wino", args.levels)
            if tr is not None:
                out["roofline"]["traffic"] = tr["gb_per_launch"]
                out["roofline"]["traffic_unit"] = "GB per launch"
                out["roofline"]["traffic_detail"] = tr
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(job_host, cfg, args.cpu_closures)
        print(json.dumps(out), flush=True)
    for e2, o2 in extra:
        o2.close()
        e2.close()
    opt.close()
    if comm is not None:
        comm.close()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
