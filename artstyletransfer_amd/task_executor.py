"""Job scheduler with the reference's surface (task_executor.py:1-130): `Task`, `Executor`
(`add_task`, `get_progress`, `progress`, `task_ids`, `set_progress`, `run`), progress tuples
`(percent, image)` initialised to `(-1, None)`.

Difference: the reference funnels every job through one global `Semaphore(2)` onto GPU 0; here a
`GpuSlots` pool hands each job a GPU of the node (`config.simultaneous_tasks_count` jobs per
GPU), so 8 independent content x style jobs run one per MI355X with no collective between them
(BASELINE config 5)."""
from __future__ import annotations

import asyncio
from typing import Callable, Optional

import torch

from . import config as _config
from .neural_style_transfer import ContentStylePair, neural_style_transfer


class GpuSlots:
    """Pool of (gpu index) tokens: `per_gpu` tokens for every visible GPU."""

    def __init__(self, per_gpu: Optional[int] = None, n_gpus: Optional[int] = None):
        self.per_gpu = per_gpu if per_gpu is not None else _config.simultaneous_tasks_count
        self.n_gpus = n_gpus if n_gpus is not None else torch.cuda.device_count()
        self._queue: Optional[asyncio.Queue] = None

    def _q(self) -> asyncio.Queue:
        if self._queue is None:
            self._queue = asyncio.Queue()
            # interleave so that consecutive jobs land on different GPUs first
            for _ in range(max(self.per_gpu, 1)):
                for g in range(self.n_gpus):
                    self._queue.put_nowait(g)
        return self._queue

    async def acquire(self) -> int:
        if self.n_gpus < 1:
            raise RuntimeError("no GPU visible: the HIP style-transfer engine has no CPU path")
        return await self._q().get()

    def release(self, gpu: int) -> None:
        self._q().put_nowait(gpu)


slots = GpuSlots()


class Task:
    """One optimisation job; reports every yielded result to the Executor."""

    def __init__(self, content_n_style: ContentStylePair, config, task_id: str, report: Callable,
                 job_done: Callable, gpu_slots: Optional[GpuSlots] = None):
        self.__task_id = task_id
        self.__report = report
        self.__job_done_callback = job_done
        self.__content_n_style = content_n_style
        self.__config = config
        self.__slots = gpu_slots or slots
        self.gpu = None
        self.job = asyncio.create_task(self.__do_job())

    async def __do_job(self):
        cfg = self.__config
        gpu = await self.__slots.acquire()
        self.gpu = gpu
        try:
            async for percent, img in neural_style_transfer(
                    self.__content_n_style, cfg.content_weight, cfg.style_weight, cfg.tv_weight, cfg.optimizer,
                    cfg.model, cfg.init_method, cfg.iters_num, cfg.levels_num, cfg.noise_factor, cfg.noise_levels,
                    cfg.noise_levels_central_amplitude, cfg.noise_levels_peripheral_amplitude,
                    cfg.noise_levels_dispersion, device=torch.device("cuda", gpu)):
                await self.__report(self.__task_id, (percent, img.copy()))
        finally:
            self.__slots.release(gpu)
        await self.__job_done_callback(self.__task_id)


class Executor:
    """Runs the optimisation tasks and keeps the latest (percent, image) of each."""

    def __init__(self, config, report_progress=None, gpu_slots: Optional[GpuSlots] = None):
        self.__tasks = {}
        self.__progress = {}
        self.__config = config
        self.__progress_lock = asyncio.Lock()
        self.__tasks_lock = asyncio.Lock()
        self.__report_progress = report_progress
        self.__slots = gpu_slots
        self.verbose = False

    @staticmethod
    def __copy(value):
        return value[0], (value[1].copy() if value[1] is not None else None)

    async def get_progress(self, key):
        async with self.__progress_lock:
            return self.__copy(self.__progress[key])

    async def progress(self):
        async with self.__progress_lock:
            for item in self.__progress.items():
                yield item

    async def task_ids(self):
        async with self.__progress_lock:
            return list(self.__progress.keys())

    async def set_progress(self, key, value):
        async with self.__progress_lock:
            self.__progress[key] = self.__copy(value)

    async def __report(self, task_id, result):
        await self.set_progress(task_id, result)
        if self.verbose:
            async for tid, p in self.progress():
                print(f"Progress: {tid}, {p[0]}")
        if self.__report_progress is not None:
            await self.__report_progress(task_id, result)

    async def __job_done(self, task_id):
        async with self.__tasks_lock:
            self.__tasks.pop(task_id, None)

    async def add_task(self, task_id: str, content_n_style: ContentStylePair):
        await self.set_progress(task_id, (-1, None))
        async with self.__tasks_lock:
            task = Task(content_n_style, self.__config, task_id=task_id, report=self.__report,
                        job_done=self.__job_done, gpu_slots=self.__slots)
            self.__tasks[task_id] = task
            return task.job

    async def run(self, forever=False):
        """Serve until no task is left, repeatedly when `forever` (the bot front-end's loop). With
        forever=False this returns at once, as the reference's does (task_executor.py:116-118)."""
        while forever:
            while True:
                async with self.__tasks_lock:
                    jobs = [t.job for t in self.__tasks.values()]
                if not jobs:
                    break
                await asyncio.wait(jobs)
            await asyncio.sleep(1)

    async def wait_all(self):
        """Extension: wait until every task added so far has finished.  A job that ended in an exception never calls
        job_done (as in the reference, task_executor.py:28-43, where such a task stays listed for ever): it is taken off
        the list here and its exception re-raised, instead of being waited for again and again."""
        while True:
            async with self.__tasks_lock:
                jobs = {t.job: tid for tid, t in self.__tasks.items()}
            if not jobs:
                return
            done, _ = await asyncio.wait(jobs.keys(), return_when=asyncio.FIRST_EXCEPTION)
            for job in done:
                if job.cancelled() or job.exception() is not None:
                    async with self.__tasks_lock:
                        self.__tasks.pop(jobs[job], None)
                    if not job.cancelled():
                        raise job.exception()
