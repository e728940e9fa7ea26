"""One ingest node: N worker processes, one per GPU, each running ``process_ml_task`` over its share of the jobs.

The reference scales the ml-service by starting more arq workers on the one Redis queue
(``/root/reference/ml-service/src/main_worker.py:118-129``, ``dev/docker-compose.yml:86-131``: one container with
the GPU attached, ``WORKER_MAX_JOBS=1``); a job is one (task_type, video) pair and jobs are independent
(``task_handler.py:22-29``).  On an 8-GPU MI355X node that is 8 workers, each pinned to its GPU **before its first
GPU call** (``HIP_VISIBLE_DEVICES``: the HIP runtime reads it when it initialises, so it has to be in the child's
environment before torch / libeioku_hip are imported) - BASELINE cfg4: 8 x 1 h 1080p videos shard one-per-GPU, no
data-path collective.

Without Redis in the loop the queue is a static plan: longest-processing-time-first over a per-job weight (frames,
bytes or seconds; 1 when unknown), which is what a shared queue converges to when the long jobs are enqueued first.
Results come back in input order.  A job that raises is reported ``failed`` with its message and does not stop the
others - ``max_tries=1`` in the reference (``main_worker.py:126``).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import sys
import queue
import time
from dataclasses import dataclass, field


@dataclass
class Job:
    task_id: str
    task_type: str
    video_id: str
    video_path: str
    config: dict = field(default_factory=dict)
    weight: float = 1.0


def plan(jobs: list[Job], workers: int) -> list[list[int]]:
    """Job indices per worker: longest first, each to the least loaded worker (ties: lower worker, input order)."""
    if workers < 1:
        raise ValueError("workers must be >= 1")
    order = sorted(range(len(jobs)), key=lambda i: (-float(jobs[i].weight), i))
    load = [0.0] * workers
    out: list[list[int]] = [[] for _ in range(workers)]
    for i in order:
        w = min(range(workers), key=lambda j: (load[j], j))
        out[w].append(i)
        load[w] += float(jobs[i].weight)
    return out


def _worker(gpu: int, jobs: list[tuple[int, Job]], results, ctx_factory_path: str | None, extra_env: dict):
    """Child process: pin the GPU, THEN import the GPU stack and run the jobs one after another (max_jobs = 1)."""
    os.environ["HIP_VISIBLE_DEVICES"] = str(gpu)      # physical GPU `gpu` becomes this process' device 0
    os.environ["EIOKU_HIP_DEVICE"] = "0"
    os.environ.update(extra_env)
    from eioku_amd import _lib as _binding  # the ctypes table only: nothing is dlopen()ed by importing it

    assert "torch" not in sys.modules and _binding._lib is None, "GPU stack loaded before the pin"
    import asyncio
    import importlib

    from eioku_amd import task_handler

    ctx = {}
    if ctx_factory_path:  # "package.module:function" returning the arq-style ctx dict (task store, artifact sink ...)
        mod, fn = ctx_factory_path.split(":")
        ctx = getattr(importlib.import_module(mod), fn)(gpu)
    for index, job in jobs:
        t0 = time.perf_counter()
        try:
            r = asyncio.run(task_handler.process_ml_task(ctx, job.task_id, job.task_type, job.video_id, job.video_path,
                                                         job.config))
            r = dict(r, gpu=gpu, seconds=time.perf_counter() - t0)
        except Exception as e:  # noqa: BLE001 - the reference marks the task failed and moves on (max_tries = 1)
            r = {"task_id": job.task_id, "status": "failed", "error": str(e), "gpu": gpu,
                 "seconds": time.perf_counter() - t0}
        results.put((index, r))
    results.put((-1, gpu))  # this worker is done


def run_node(jobs: list[Job], gpus: list[int], ctx_factory: str | None = None, worker=_worker,
             extra_env: dict | None = None, timeout: float | None = None) -> list[dict]:
    """Run ``jobs`` on ``gpus`` (physical device ids), one process per GPU; returns one result dict per job, input order.

    ``worker`` is a seam for tests (a function with ``_worker``'s signature that needs no GPU)."""
    shares = plan(jobs, len(gpus))
    ctx = mp.get_context("spawn")  # never fork a process that may have touched the GPU
    results = ctx.Queue()
    procs = []
    for gpu, share in zip(gpus, shares):
        p = ctx.Process(target=worker, args=(gpu, [(i, jobs[i]) for i in share], results, ctx_factory, dict(extra_env or {})))
        p.start()
        procs.append(p)
    out: list[dict | None] = [None] * len(jobs)
    finished = [False] * len(procs)      # the worker posted its (-1, gpu) sentinel, or it died
    exit_codes: dict[int, int] = {}
    deadline = None if timeout is None else time.monotonic() + timeout
    timed_out = False
    try:
        while not all(finished):
            # a short poll, not a blocking get: a worker that dies (GPU fault, OOM kill, segfault, a failed pin assert) never
            # posts its sentinel, and with timeout=None the parent used to wait for it forever
            try:
                index, r = results.get(timeout=0.2)
            except queue.Empty:
                for i, p in enumerate(procs):
                    if not finished[i] and not p.is_alive():
                        try:  # whatever the dead worker still had in the pipe
                            while True:
                                index, r = results.get(timeout=0.05)
                                if index < 0:
                                    finished[gpus.index(r)] = True
                                else:
                                    out[index] = r
                        except queue.Empty:
                            pass
                        finished[i] = True
                        exit_codes[i] = p.exitcode if p.exitcode is not None else -1
                if deadline is not None and time.monotonic() > deadline:
                    timed_out = True
                    break
                continue
            if index < 0:
                finished[gpus.index(r)] = True
            else:
                out[index] = r
    finally:
        for p in procs:
            p.join(5 if not timed_out else 0.1)
            if p.is_alive():
                p.terminate()
    for w, share in enumerate(shares):
        for i in share:
            if out[i] is None:
                why = (f"worker process for GPU {gpus[w]} died (exit code {exit_codes[w]})" if w in exit_codes
                       else "timed out" if timed_out else "worker process died")
                out[i] = {"task_id": jobs[i].task_id, "status": "failed", "error": why, "gpu": gpus[w]}
    return out  # type: ignore[return-value]
