#!/usr/bin/env python3
"""Hot-path benchmark: frames/s through scene + detect + embed on synthetic frames resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input on every rank (weak
scaling: videos shard one-per-GPU, no data-path collective - SURVEY.md 8e).  Rank 0 prints ONE JSON
line.  `roofline` is the dominant kernel's algorithmic work / its HIP-event duration measured on the
launch stream inside the timed region; `cpu_baseline` is the CPU oracle on a bounded sample (rank 0,
N=1 only).  See DESIGN.md "Measurement" for the per-unit byte/flop figures.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per step per GPU (BASELINE cfg2: 64)")
    ap.add_argument("--height", type=int, default=640)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU oracle sample budget")
    return ap.parse_args()


class ScenePipeline:
    """Stages built so far.  Each step consumes `batch` BGR frames already resident in HBM."""

    def __init__(self, args, device, rank):
        from eioku_amd import synth

        self.args = args
        self.device = device
        self.batch = args.batch
        self.h, self.w = args.height, args.width
        # two alternating batches of one synthetic video (scene changes included), carried prev frame
        self.frames = [synth.frames_bgr(1234 + rank, self.batch, self.h, self.w, device, first_frame=b * self.batch)
                       for b in range(2)]
        self.prev = None
        self.last = None

    stages = ["scene(ContentDetector HSV K2 + luma SAD K1)"]
    missing = ["detect(YOLOv8n)", "embed(MiniLM)"]

    def step(self, i):
        from eioku_amd import scene

        f = self.frames[i & 1]
        sums = scene.hsv_sums(f, self.prev, keep_on_device=True)
        sad = scene.luma_sad(self._luma(i), keep_on_device=True)
        self.prev = f[-1]
        self.last = (sums, sad)

    def _luma(self, i):
        # synthetic Y plane: the decoder would hand this over; here = G channel copy made at init
        if not hasattr(self, "_y"):
            self._y = [fr[..., 1].contiguous() for fr in self.frames]
        return self._y[i & 1]

    def dominant(self):
        from eioku_amd import _lib

        ms, cnt = _lib.prof_read(_lib.PROF_SCENE_HSV)
        alg_bytes = 3.0 * self.h * self.w * self.batch  # 3*W*H bytes per frame
        return {"kernel": "k_hsv_sums", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "alg_per_launch": alg_bytes, "ms_total": ms, "launches": cnt}


def cpu_baseline_scene(args):
    """CPU oracle (numpy port) on a bounded sample of the same workload: frames/s on host cores."""
    import numpy as np
    from oracle import prng, scene as oscene

    n = 3
    frames = prng.synth_frames_bgr(1234, n, args.height, args.width)
    t0 = time.perf_counter()
    done = 0
    while True:
        oscene.content_sums(frames)
        oscene.luma_sad(np.ascontiguousarray(frames[..., 1]))
        done += n - 1  # n frames give n-1 scored transitions
        if time.perf_counter() - t0 > args.cpu_seconds / 3:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"oracle scene stage (HSV + luma SAD) on {n} synthetic {args.height}x{args.width} frames, "
                      f"{done} frame transitions in {dt:.1f}s, numpy single thread"}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from eioku_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    _lib.init(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    pipe = ScenePipeline(args, device, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        pipe.step(i)
    barrier()
    _lib.prof_enable(True)
    _lib.prof_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        pipe.step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    dom = pipe.dominant()
    avg_ms = dom["ms_total"] / max(dom["launches"], 1)
    achieved = dom["alg_per_launch"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    frames_total = args.batch * args.steps * world
    out = {
        "metric": "frames/sec (scene+detect+embed) per node",
        "value": frames_total / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{args.batch}x{args.height}x{args.width} BGR u8 frames/step/GPU resident in HBM; "
                               f"stages run: {', '.join(pipe.stages)}; NOT YET BUILT (so value is not the full metric): "
                               f"{', '.join(pipe.missing)}",
                   "batch": args.batch, "frame": [args.height, args.width], "parallelism": f"shard-by-video x{world}"},
        "roofline": {"bound": dom["bound"], "kernel": dom["kernel"], "achieved": achieved, "peak": dom["peak"],
                     "unit": dom["unit"], "frac": achieved / dom["peak"], "traffic": None,
                     "avg_kernel_ms": avg_ms, "launches": dom["launches"],
                     "algorithmic_per_launch": dom["alg_per_launch"]},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_scene(args)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
