"""Job driver and optimisation loop with the reference's public surface
(neural_style_transfer.py:32-439), re-implemented on the MI355X HIP engine.

What changed underneath: the reference builds an autograd graph per closure out of torch ops and
lets torch.optim walk it; here one `nst_opt_step` call runs the whole `optimizer.step(closure)` -
pyramid down-sampling, VGG19 forward of every level, Gram/content/TV losses, the hand-written
backward and the Adam / L-BFGS update - inside libnst_hip.so on the device-resident pixel buffer.
There is no CPU path: without a GPU and the built extension these entry points raise."""
from __future__ import annotations

import asyncio
import os
import traceback
from typing import List, Optional

import numpy as np
import torch

from . import device_image, host_image, math_utils
from .engine import PixelOptimizer, StyleEngine
from .neural_nets import lease_engine, return_engine, shared_engine

# ImageNet statistics (reference :22-23)
IMAGENET_MEAN_255 = [123.675, 116.28, 103.53]
IMAGENET_STD_NEUTRAL = [1, 1, 1]

# how many closure evaluations LBFGS may spend per step; 1 = torch 2.10 semantics of the reference's
# constructor arguments (almost every trial step is rejected), 26 = the line search older torch builds
# performed.  Environment NST_LBFGS_MAX_EVAL overrides it at import; see INTEGRATION.md.
LBFGS_MAX_EVAL = int(os.environ.get("NST_LBFGS_MAX_EVAL", "1"))
VERBOSE = False


class ContentStylePair:
    """content = (name, HWC float32 RGB [0,1] image), style = (name, image)."""

    def __init__(self, content, style):
        self.content = content
        self.style = style


def prepare_img(img, device):
    """HWC [0,1] -> (1,3,H,W) `*255 - ImageNet mean` on `device` (reference :375-383)."""
    dev = torch.device(device)
    hwc = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).to(dev)
    return shared_engine(dev).prepare_img(hwc)


def unprepare_img(img: torch.Tensor):
    """(1,3,H,W) -> HWC float32 numpy, `(+mean)/255`, not clipped (reference :386-393)."""
    t = img.detach().contiguous()
    return shared_engine(t.device).unprepare_img(t).cpu().numpy()


class RepresentationBuilder:
    """Content / style representations of an image from the network's six feature maps."""

    def __init__(self, image, neural_net):
        self.__features = neural_net(image)

    def build_content(self, feature_map_indices):
        listed = isinstance(feature_map_indices, list)
        idx = feature_map_indices if listed else [feature_map_indices]
        rep = [x.squeeze(0) for i, x in enumerate(self.__features) if i in idx]
        return rep if listed else rep[0]

    def build_style(self, feature_map_indices):
        listed = isinstance(feature_map_indices, list)
        idx = feature_map_indices if listed else [feature_map_indices]
        rep = [math_utils.gram_matrix(x) for i, x in enumerate(self.__features) if i in idx]
        return rep if listed else rep[0]


class LossBuilder:
    """Loss of one pyramid level. `build(x)` returns (total, content, style, tv) as device scalars
    evaluated by the fused HIP closure (they carry no autograd graph; the loop obtains the
    gradient from the same closure call)."""

    def __init__(self, content_feature_maps_index, style_feature_maps_indices, target_content_image,
                 target_style_image, neural_net, content_weight, style_weight, tv_weight):
        if content_feature_maps_index != 4 or list(style_feature_maps_indices) != [0, 1, 2, 3, 5]:
            raise ValueError("the HIP closure implements the reference's VGG19 taps (content 4, style [0,1,2,3,5])")
        self.__weights = (float(content_weight), float(style_weight), float(tv_weight))
        c = target_content_image
        # a context from the per-GPU pool (the weights are uploaded once, not per LossBuilder); it goes back when this
        # object is collected
        self.__engine = lease_engine(c.device)
        self.__engine.configure(1, c.shape[-2], c.shape[-1])
        self.__engine.set_targets(0, c.contiguous(), target_style_image.contiguous())

    def __del__(self):
        try:
            return_engine(self.__engine)
        except Exception:
            pass

    def build(self, optimizing_img):
        cw, sw, tvw = self.__weights
        _, losses = self.__engine.closure(optimizing_img.detach().contiguous(), cw, sw, tvw)
        return losses[0], losses[1], losses[2], losses[3]


class _DeviceJob:
    """Everything one job owns on the GPU: the engine (targets, workspace), the optimiser (Adam moments / L-BFGS
    history), the job's HIP stream, the side stream and the two pinned host buffers of the per-step yield.
    `NeuralStyleTransfer.process` drives it through four calls - step (pool thread), snapshot, image, close - and is
    otherwise plain asyncio, so the hand-over and tear-down ordering can be tested with a fake in its place
    (`_make_job`, tests/test_host_api.py)."""

    def __init__(self, device, optimizer_name, style_imgs, content_imgs, init_img, lr_start):
        self.dev = dev = device
        self.optimizer = None
        self.engine = lease_engine(dev)
        h0, w0 = init_img.shape[:2]
        # Every job runs on a HIP stream of its own: the jobs that share a GPU (`config.simultaneous_tasks_count`
        # per GPU, as in the reference) then overlap on the device - one job's launch tails, host round trips and
        # HBM-bound kernels run under the other's MFMA-bound ones - instead of queueing behind each other on the
        # default stream.  The current stream is per THREAD and the jobs' coroutines share the event-loop thread, so
        # it is set around synchronous sections only, never across an await.
        self.job_stream = torch.cuda.Stream(device=dev)
        self.job_stream.wait_stream(torch.cuda.current_stream(dev))     # the caller built the input images there
        self.copy_stream = torch.cuda.Stream(device=dev)
        try:
            engine = self.engine

            def prepared(img):      # numpy HWC (reference) or a device HWC tensor built by device_image
                if isinstance(img, torch.Tensor):
                    return engine.prepare_img(img.to(dev).contiguous())
                return prepare_img(img, dev)

            with torch.cuda.stream(self.job_stream):
                engine.configure(len(content_imgs), h0, w0)
                for lvl, (c_img, s_img) in enumerate(zip(content_imgs, style_imgs)):
                    if tuple(c_img.shape[:2]) != engine.level_shape(lvl):
                        raise ValueError(f"content level {lvl} is {tuple(c_img.shape[:2])}, expected {engine.level_shape(lvl)}")
                    engine.set_targets(lvl, prepared(c_img), prepared(s_img))
                self.x = prepared(init_img)
                self.optimizer = PixelOptimizer(engine, optimizer_name, lr_start, LBFGS_MAX_EVAL)
            # Per-step yield (reference :207-208): the image is un-prepared into its own device buffer, copied to
            # pinned host memory on a side stream, and the NEXT optimiser step is started before that copy is
            # awaited, so the 4*3*H*W-byte D2H hides under the next closures.
            self.host = [torch.empty((h0, w0, 3), dtype=torch.float32, pin_memory=True) for _ in range(2)]
        except BaseException:
            self.close()
            raise

    def step(self, cw, sw, tvw):
        """One optimizer.step(closure) (nst_opt_step).  Runs on a pool thread, whose current stream is its own."""
        with torch.cuda.stream(self.job_stream):
            return self.optimizer.step(self.x, cw, sw, tvw, want_losses=True)

    def snapshot(self, k):
        """Un-prepare the current image on the job's stream (ordered BEFORE the next step, which the caller starts
        after this returns) and start its D2H into host buffer k on the side stream.  Returns a callable that blocks
        until the copy has landed."""
        with torch.cuda.device(self.dev), torch.cuda.stream(self.job_stream):
            snap = self.engine.unprepare_img(self.x)
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(ready)
                self.host[k].copy_(snap, non_blocking=True)
                snap.record_stream(self.copy_stream)
                done = torch.cuda.Event()
                done.record()
        return done.synchronize

    def image(self, k):
        return self.host[k].numpy().copy()

    def close(self):
        """The optimiser goes first (its curvature history alone is up to 200 x 12*H*W bytes), then the engine it was
        created on.  The caller guarantees that no step is running (process() drains the worker thread first)."""
        if self.optimizer is not None:
            self.optimizer.close()
            self.optimizer = None
        self.job_stream.synchronize()
        self.copy_stream.synchronize()
        return_engine(self.engine)             # back to the per-GPU pool: the next job re-uses its uploaded weights


def _make_job(device, optimizer_name, style_imgs, content_imgs, init_img, lr_start):
    return _DeviceJob(device, optimizer_name, style_imgs, content_imgs, init_img, lr_start)


async def _drain(step_future):
    """Returns when the WORKER THREAD has left the optimiser step behind `step_future` (the asyncio future
    run_in_executor returned).  Every await on that future goes through asyncio.shield, so cancelling the task never
    cancels the future itself - a cancelled run_in_executor future says nothing about its thread, which would go on
    inside nst_opt_step while the clean-up frees the optimiser and the engine under it.  A CancelledError that arrives
    while waiting is kept and returned for the caller to re-raise after the clean-up."""
    cancelled = None
    while not step_future.done():
        try:
            await asyncio.shield(step_future)
        except asyncio.CancelledError as e:
            if not step_future.done():
                cancelled = e                    # the task was cancelled (again); the thread is still in the step
        except BaseException:
            pass                                 # the step's own failure: the future is done
    if not step_future.cancelled():
        step_future.exception()                  # retrieved: an abandoned step must not warn at garbage collection
    return cancelled


class NeuralStyleTransfer:
    """The optimisation loop (reference :115-208)."""

    def __init__(self, device, model_name, style_imgs, optimizer_name):
        self.__device = torch.device(device)
        self.__model_name = model_name
        self.__style_imgs = style_imgs
        self.__optimizer_name = optimizer_name

    async def process(self, content_imgs, init_img, lr_start, iters_num, content_weight, style_weight, tv_weight,
                      init_img_name):
        # validates the model name exactly as the reference does (ValueError for anything but vgg19)
        math_utils.prepare_model(self.__model_name, self.__device)
        if self.__optimizer_name not in ("adam", "lbfgs"):
            raise RuntimeError("Unknown optimizer")
        if self.__device.type != "cuda":
            raise RuntimeError("the HIP style-transfer engine needs a GPU; no CPU path exists")
        job = _make_job(self.__device, self.__optimizer_name, self.__style_imgs, content_imgs, init_img, lr_start)
        cw, sw, tvw = float(content_weight), float(style_weight), float(tv_weight)
        loop = asyncio.get_running_loop()

        def one_step():
            try:
                return job.step(cw, sw, tvw)
            except Exception:
                traceback.print_exc()
                raise

        step, k = 0, 0
        pending = loop.run_in_executor(None, one_step) if step < iters_num else None
        cancelled = None
        try:
            while pending is not None:
                info, rows = await asyncio.shield(pending)
                pending = None
                step = info.total_closures
                if VERBOSE:
                    for r in rows:
                        print(f"{self.__optimizer_name} | {init_img_name} | lr={info.lr:.4f} | total loss={r[-1]:.3e}")
                copied = job.snapshot(k)                 # ordered before the next step on the job's stream
                if step < iters_num:
                    pending = loop.run_in_executor(None, one_step)
                await loop.run_in_executor(None, copied)
                img = job.image(k)
                k ^= 1
                yield img, step
        finally:
            # Normal end, a consumer that stops early (aclose / GeneratorExit at the yield), a cancelled task or a
            # failed step all come through here.  First the worker thread itself is waited for (not the cancellable
            # future around it), then the job's device objects go.
            if pending is not None:
                cancelled = await _drain(pending)
            job.close()
            if cancelled is not None:
                raise cancelled


async def resize(img, level):
    """Image resized for pyramid level `level`: short edge 256 * 2**level, bicubic, always from the
    original (reference :211-226)."""
    return host_image.resize_to_level(np.array(img, copy=True), level)


gaussian_mask = host_image.gaussian_mask
make_style_noise = host_image.make_style_noise


async def neural_style_transfer(content_n_style: ContentStylePair,
                                content_weight, style_weight, tv_weight,
                                optimizer, model, init_method,
                                iters_num, levels_num, noise_factor, noise_levels, noise_levels_central_amplitude,
                                noise_levels_peripheral_amplitude, noise_levels_dispersion, device=None):
    """Async generator yielding (percent, HWC float32 image) after every optimiser step
    (reference :229-372). `device` (extension): the GPU to run on; default = current."""
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the HIP style-transfer engine has no CPU path")
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)

    # pyramid + structured-noise initial image, on the device (device_image.py; host_image.py is the
    # host restatement of the same algorithm)
    setup = shared_engine(device)
    content_dev = device_image.upload(setup, content_n_style.content[1])
    style_dev = device_image.upload(setup, content_n_style.style[1])
    content_levels = device_image.pyramid(setup, content_dev, levels_num)
    style_levels = device_image.pyramid(setup, style_dev, levels_num)
    level = max(levels_num - 1, 0)
    init_img, tag = device_image.initial_image(
        setup, init_method, content_dev, style_dev, content_levels[0], style_levels[0], level,
        noise_factor, noise_levels, noise_levels_central_amplitude, noise_levels_peripheral_amplitude,
        noise_levels_dispersion)
    init_name = {"random": "random", "content": content_n_style.content[0], "style": content_n_style.style[0]}[tag]

    nst = NeuralStyleTransfer(device, model, style_levels, optimizer)
    lr_start = 10.0
    async for img, cur_iter in nst.process(content_levels, init_img, lr_start, iters_num, content_weight,
                                           style_weight, tv_weight, init_name):
        yield cur_iter / iters_num * 100.0, img
