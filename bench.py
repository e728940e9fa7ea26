#!/usr/bin/env python3
"""Hot-path benchmark: frames/s through scene + detect + embed on synthetic frames resident in HBM,
plus kNN QPS@top-10 over 10M x 384 (BASELINE.json's two-part metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input on every rank: 64 BGR frames
(BASELINE cfg2: 640x640) through scene scoring (K2 ContentDetector sums + K1 luma SAD), YOLOv8n
detection (K3 letterbox, K4/K5 network, K6 decode, K7 NMS) and MiniLM embedding of 8 transcript
segments x 128 tokens (K8; one segment per 8 frames is far denser than real speech).  Weak scaling:
videos shard one-per-GPU, no data-path collective.  Rank 0 prints ONE JSON line.
`roofline` is the dominant kernel's algorithmic work / its HIP-event duration measured on the launch
stream inside the timed region; `cpu_baseline` is the CPU oracle on a bounded sample (rank 0, N=1).
The kNN part runs after the timed frames region: rows sharded over ranks, replicated queries, one
RCCL all-gather + local merge per search.  `frames_1080p` repeats the frames measurement on the north star's
own source size (64 x 1080 x 1920 BGR frames per step: bilinear letterbox to 384 x 640) with its own roofline.
`cfg3` / `cfg4` / `cfg5` carry BASELINE.json's remaining configurations in the same driver-run line (MiniLM 512 x 128 +
flat kNN over 1 M x 384; YOLOv8m objects + yolov8n-face on 64 x 1080p; IVF-PQ over a 12.5 M x 384 shard per GPU).
`python bench.py --gpus N` without WORLD_SIZE starts its own N ranks (a child `python -m torch.distributed.run`, spawned
before anything touches the GPU) and relays their line.  See DESIGN.md "Measurement".
"""
from __future__ import annotations

import argparse
import hashlib
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
MFMA_F32_PEAK_TFLOPS = 157.3


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per step per GPU (BASELINE cfg2: 64)")
    ap.add_argument("--height", type=int, default=640)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--model", default="yolov8n.pt")
    ap.add_argument("--conf", type=float, default=0.25)
    ap.add_argument("--segments", type=int, default=8, help="transcript segments embedded per step")
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--embed-batch", type=int, default=0,
                    help="segments per encoder call (eioku_amd.embed.SegmentBatcher; 0 = the step's own segments, every step): "
                         "larger batches cost 2.8x less encoder time per segment (tools/embed_batch_sweep.py) but the 2.4 ms "
                         "burst of 128 x 128-tile GEMMs every 16th step slows the detector beside it by more than that "
                         "(measured 1.64 vs 1.51 ms per step), so the default keeps one call per step")
    ap.add_argument("--stages", default="scene,detect,embed", help="comma list (debug)")
    ap.add_argument("--knn-n", type=int, default=10_000_000, help="0 disables the kNN part")
    ap.add_argument("--knn-nq", type=int, default=1024)
    ap.add_argument("--knn-iters", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=2, help="detector handles alternating over consecutive steps (with --overlap)")
    ap.add_argument("--overlap", type=int, default=1, help="1: scene / detect / embed on three HIP streams; 0: one stream")
    ap.add_argument("--prof-every", type=int, default=-1,
                    help="HIP-event kernel timing on every n-th timed step (0: off, -1: one step in the middle of the region)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU oracle sample budget")
    ap.add_argument("--no-1080p", action="store_true", help="skip the frames_1080p block")
    ap.add_argument("--knn-mode", type=int, default=-1, help="scan_mode of the index: 0 register-tile kernels, 1 scan path (-1: library default)")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the cfg3 block (MiniLM 512 x 128, flat kNN 1 M x 384)")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the cfg4 block (YOLOv8m + yolov8n-face on 64 x 1080p)")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the cfg5 block (IVF-PQ over a 12.5 M x 384 shard per GPU)")
    ap.add_argument("--ivfpq-n", type=int, default=12_500_000, help="rows per GPU of the cfg5 index (100 M / 8)")
    ap.add_argument("--cfg-steps", type=int, default=40, help="timed steps of the cfg4 frame runs")
    return ap.parse_args(argv)


def launcher_decision(gpus: int, env) -> str:
    """How this invocation runs: "ranked" = one rank of a torch.distributed.run job (RANK / WORLD_SIZE in the environment:
    the driver's N > 1 launch), "spawn" = `python bench.py --gpus N>1` on its own: start the N ranks as a child job,
    "inline" = a single GPU in this process."""
    if "WORLD_SIZE" in env or "RANK" in env:
        return "ranked"
    return "spawn" if gpus > 1 else "inline"


def spawn_ranks(gpus: int, argv) -> int:
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <argv>` as a CHILD process (never an
    exec: this must also be safe from a process that has touched the GPU, and this one has not); its stdout - rank 0's
    JSON line - passes straight through.  Returns the child's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def kernel_source_hash() -> str:
    """sha256 over the sources of the conv family and of the graph that launches it (conv.hip / yolo.hip / yolo_ops.hip,
    the headers, the Makefile with its per-file flags): ties a committed PMC traffic measurement of that family to the
    code it was taken on (the GPU box has no .git, so a commit id cannot be read there)."""
    h = hashlib.sha256()
    csrc = ROOT / "eioku_amd" / "csrc"
    for f in sorted(list(csrc.glob("*.h")) + [csrc / n for n in ("conv.hip", "yolo.hip", "yolo_ops.hip", "Makefile")]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


_STREAMS = {}


def lane_streams(device, depth):
    """The detector lanes' high-priority streams, created once per process: a stream's hardware queue is fixed at
    creation, and the streams a later Pipeline would create can share queues with one another (measured: the cfg4
    runs at 41 k instead of 58 k frames/s behind the earlier runs of the same process)."""
    import torch

    key = ("lanes", str(device), depth)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=device, priority=-1) for _ in range(depth)]
    return _STREAMS[key]


def stage_streams(device):
    import torch

    key = ("stages", str(device))
    if key not in _STREAMS:
        _STREAMS[key] = {k: torch.cuda.Stream(device=device) for k in ("scene", "embed")}
    return _STREAMS[key]


class Pipeline:
    """Each step consumes `batch` BGR frames already resident in HBM and leaves its results in HBM."""

    BUILT = ("scene", "detect", "embed")

    def __init__(self, args, device, rank, height=None, width=None):
        import torch

        from eioku_amd import detect, embed, synth

        self.args = args
        self.device = device
        self.batch, self.h, self.w = args.batch, height or args.height, width or args.width
        want = [s for s in args.stages.split(",") if s]
        self.stages = [s for s in want if s in self.BUILT]
        self.missing = [s for s in ("scene", "detect", "embed") if s not in self.stages]
        # two alternating batches of one synthetic video (scene changes included)
        self.frames = [synth.frames_bgr(1234 + rank, self.batch, self.h, self.w, device, first_frame=b * self.batch)
                       for b in range(2)]
        self.luma = [f[..., 1].contiguous() for f in self.frames]  # stands in for the decoder's Y plane
        self.prev = None
        self.det = None
        self.pdet = None
        self.enc = None
        if "detect" in self.stages:
            self.det = detect.Yolov8Detector.from_model_name(args.model, seed=7)  # random-init weights, exact shapes
            # a random net's logit scale is arbitrary: give it a trained detector's candidate density
            self.det.calibrate_random_head(self.frames[0][:8], frac=0.01, conf=args.conf)
            # --depth 2: eioku_amd.detect.PipelinedDetector, the same object ModelManager's frame loop runs on: a
            # second handle (same weights, own activation buffers) on its own stream, consecutive batches overlap
            self.pdet = detect.PipelinedDetector(self.det, depth=max(1, args.depth), device=device,
                                                 streams=lane_streams(device, max(1, args.depth))) if args.overlap else None
        self.batcher = None
        self.embedded = []
        if "embed" in self.stages:
            self.enc = embed.MiniLMEncoder(embed.random_state(embed.MINILM_L6_V2, 11))
            g = torch.Generator(device="cpu").manual_seed(11 + rank)
            # 16 steps' worth of distinct segments, walked cyclically
            self.ids = torch.randint(1000, 30000, (16, args.segments, args.seq_len), generator=g, dtype=torch.int32).to(device)
            self.mask = torch.ones((args.segments, args.seq_len), dtype=torch.uint8, device=device)
            self.batcher = embed.SegmentBatcher(self.enc, args.seq_len, max(args.segments, args.embed_batch), device)
        self.last = None
        # The three stages of a step share no data (the reference runs them as separate tasks): each gets its own
        # HIP stream so that the many small launches of one (20x20 convs, M=1024 GEMMs, each a fraction of the
        # chip) fill the CUs and the launch-to-launch gaps that another leaves idle.
        self.streams = None
        if args.overlap:
            self.streams = stage_streams(device)  # the same streams in every run of this process (see lane_streams)

    def _on(self, name):
        import contextlib

        import torch

        return torch.cuda.stream(self.streams[name]) if self.streams and not self.serial else contextlib.nullcontext()

    def step(self, i, serial=False):
        """One batch through scene + detect + embed.  serial=True (the HIP-event profiled steps): everything on
        the current stream, so that a kernel's event-bracketed duration is its own and not a shared chip's."""
        from eioku_amd import scene

        self.serial = serial

        f = self.frames[i & 1]
        out = []
        if self.det is not None:
            if self.pdet is not None and not serial:
                self.pdet.submit(f, conf=self.args.conf)
                if self.pdet.in_flight() >= self.pdet.depth:
                    out.append(self.pdet.result_on_device())  # results stay in HBM; nothing waits on the host
            else:
                out.append(self.det.detect(f, conf=self.args.conf, keep_on_device=True))
        if self.enc is not None:
            with self._on("embed"):
                self.embedded = self.batcher.add(self.ids[i % 16], self.mask) or self.embedded
        if "scene" in self.stages:
            with self._on("scene"):
                out.append(scene.hsv_sums(f, self.prev, keep_on_device=True))
                out.append(scene.luma_sad(self.luma[i & 1], keep_on_device=True))
                self.prev = f[-1]
        self.last = out

    def finish(self):
        """Inside the timed region, before its closing barrier: every segment handed over so far is encoded."""
        if self.batcher is not None:
            self.serial = False
            with self._on("embed"):
                self.embedded = self.batcher.flush() or self.embedded

    def close(self):
        if self.pdet is not None:
            self.pdet.close()
        elif self.det is not None:
            self.det.close()
        if self.enc is not None:
            self.enc.close()

    def scene_kernels(self):
        """K1 / K2 of the profiled step: algorithmic bytes (W*H and 3*W*H per frame) over their
        HIP-event durations."""
        from eioku_amd import _lib

        out = {}
        for name, tag, bpf in (("k_sad_luma", _lib.PROF_SCENE_SAD, 1.0), ("k_hsv_sums", _lib.PROF_SCENE_HSV, 3.0)):
            ms, cnt = _lib.prof_read(tag)
            if cnt and ms > 0:
                alg = bpf * self.h * self.w * self.batch * cnt
                out[name] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": alg / (ms * 1e-3) / 1e9,
                             "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_kernel_ms": ms / cnt, "launches": cnt,
                             "algorithmic_bytes_per_launch": alg / cnt}
        return out

    def dominant(self):
        from eioku_amd import _lib

        if self.det is not None:
            ms, cnt = _lib.prof_read(_lib.PROF_CONV)
            flops_step = self.det.last_conv_flops()  # algorithmic 2*Cout*Cin*k*k per output pixel, whole batch
            return {"kernel": "YOLOv8 conv family: k_conv_stem_chain / k_conv3x3_chain / k_conv3x3_persist / k_conv3x3_flat / k_conv1x1 (every conv launch of a step)", "bound": "mfma", "unit": "TFLOP/s",
                    "peak": MFMA_F16_PEAK_TFLOPS, "alg_total": flops_step * self.prof_steps, "scale": 1e12,
                    "ms_total": ms, "launches": cnt, "alg_per_step": flops_step}
        ms, cnt = _lib.prof_read(_lib.PROF_SCENE_HSV)
        alg = 3.0 * self.h * self.w * self.batch  # 3*W*H bytes per frame
        return {"kernel": "k_hsv_sums", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "alg_total": alg * cnt, "scale": 1e9, "ms_total": ms, "launches": cnt, "alg_per_step": alg}


def knn_part(args, device, rank, world):
    """kNN QPS@top-10 over knn_n x 384 (rows sharded over ranks, one all-gather per search)."""
    import torch
    import torch.distributed as dist

    from eioku_amd import _lib, search, synth

    d, k = 384, 10
    lo, hi = search.shard_bounds(args.knn_n, world, rank)
    xb = synth.normal_f32(21 + rank, hi - lo, d, device, l2_normalise=True)
    q = synth.normal_f32(22, args.knn_nq, d, device, l2_normalise=True)  # same queries on every rank
    ix = search.IndexFlatL2(d)
    if args.knn_mode >= 0:
        ix.set_param("scan_mode", args.knn_mode)
    mode = args.knn_mode if args.knn_mode >= 0 else 1
    ix.attach(xb)
    sh = search.ShardedFlatL2(ix, lo)
    sh.search(q, k)  # warm-up (allocates workspaces; the wide path builds its bf16 planes of the rows here: index build)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.prof_enable(True, tags=[_lib.PROF_KNN])
    _lib.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.knn_iters):
        D, I = sh.search(q, k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms, cnt = _lib.prof_read(_lib.PROF_KNN)
    per = dt / args.knn_iters
    rows = float(hi - lo)
    kernel_s = max(ms * 1e-3 / max(cnt, 1), 1e-9)
    flops = 2.0 * args.knn_nq * rows * d          # 2 * nq * N * d
    one_pass = rows * d * 4                       # the fp32 rows, once
    scan = rows >= 262144 and mode != 0
    out = {"metric": f"kNN QPS@top-10 over {args.knn_n}x{d}", "value": args.knn_nq / per, "unit": "queries/s",
           "nq": args.knn_nq, "k": k, "ms_per_search": per * 1e3, "n_total": args.knn_n, "rows_per_gpu": hi - lo,
           "collective": "none" if world == 1 else "all_gather_into_tensor (RCCL) of nq*k*16 B per rank"}
    if scan:
        streamed = rows * d * 2  # the bf16 plane of the rows, read exactly once
        out["dtype"] = ("f32 results (sum (q-x)^2 over the fp32 rows of the candidates); candidates from a bf16 one-term filter "
                        "with a rigorous margin on the matrix cores")
        out["roofline"] = {"kernel": "k_l2_scan (row tiles stationary in registers, bf16 MFMA, every query tile streamed past "
                                     "them from L2; one HBM pass per search)",
                           "bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_F16_PEAK_TFLOPS,
                           "achieved": flops / kernel_s / 1e12, "frac": flops / kernel_s / 1e12 / MFMA_F16_PEAK_TFLOPS,
                           "algorithmic_bytes_one_pass": one_pass, "algorithmic_GBps": one_pass / kernel_s / 1e9,
                           "hbm_bytes_streamed_per_launch": streamed, "hbm_GBps_streamed": streamed / kernel_s / 1e9,
                           "note": "achieved = SURVEY 8d's 2*nq*N*d over the scan kernel's HIP-event duration; the sample "
                                   "search, candidate binning / re-rank / selection launches are inside ms_per_search, not in it"}
    else:
        passes = (args.knn_nq + 31) // 32
        wide = args.knn_nq > 64
        streamed = one_pass * ((args.knn_nq + 127) // 128 if wide else passes)
        out["dtype"] = "f32 (split-bf16 products, fp32 accumulate)" if wide else "f32"
        out["roofline"] = {"kernel": "k_flat_l2_bf (query tiles in registers: one DB pass per 128 queries)" if wide
                           else "k_flat_l2 (narrow: one DB pass per 32 queries)", "bound": "hbm", "unit": "GB/s",
                           "peak": HBM_PEAK_GBS, "achieved": one_pass * (1 if wide else passes) / kernel_s / 1e9,
                           "frac": one_pass * (1 if wide else passes) / kernel_s / 1e9 / HBM_PEAK_GBS,
                           "algorithmic_TFLOPs": flops / kernel_s / 1e12,
                           "hbm_bytes_streamed_per_launch": streamed, "hbm_GBps_streamed": streamed / kernel_s / 1e9}
    out["roofline"].update({"avg_kernel_ms": kernel_s * 1e3, "algorithmic_flops_per_launch": flops})
    ix.close()
    del xb
    torch.cuda.empty_cache()
    return out


def cfg3_block(args, device, rank, world):
    """BASELINE cfg3 on this GPU: all-MiniLM-L6-v2 on 512 segments x 128 tokens (bf16-MFMA roofline on SURVEY 8d's
    22.4 MFLOP / token) and flat kNN over 1 M x 384 at nq 1 / 64 / 1024."""
    import torch

    from eioku_amd import _lib, embed, search, synth

    out = {"config": "all-MiniLM-L6-v2 (random-init, fp32 in/out, split-bf16 MFMA) embed batch 512 x seq 128; IndexFlatL2 kNN over 1M x 384, top-10"}
    enc = embed.MiniLMEncoder(embed.random_state(embed.MINILM_L6_V2, 11))
    g = torch.Generator(device="cpu").manual_seed(11 + rank)
    ids = torch.randint(1000, 30000, (512, 128), generator=g, dtype=torch.int32).to(device)
    mask = torch.ones((512, 128), dtype=torch.uint8, device=device)
    for _ in range(2):
        enc.encode_ids(ids, mask)
    torch.cuda.synchronize()
    iters = 5
    t0 = time.perf_counter()
    for _ in range(iters):
        enc.encode_ids(ids, mask)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    _lib.prof_enable(True, tags=[_lib.PROF_GEMM])
    _lib.prof_reset()
    enc.encode_ids(ids, mask)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    gemm_ms, gemm_cnt = _lib.prof_read(_lib.PROF_GEMM)
    alg = 512 * 128 * 22.4e6
    out["minilm"] = {"metric": "segments/s (512 x 128 tokens)", "value": 512 / dt, "ms_per_batch": dt * 1e3, "dtype": "f32 (3-term split-bf16 products, fp32 accumulate)",
                     "roofline": {"kernel": "K8 encoder forward (k_gemm_bf_s GEMMs + attention + add/LN + pooling; whole batch)",
                                  "bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_F16_PEAK_TFLOPS, "achieved": alg / dt / 1e12,
                                  "frac": alg / dt / 1e12 / MFMA_F16_PEAK_TFLOPS, "algorithmic_flops_per_batch": alg,
                                  "executed_flops_per_batch": enc.last_flops(), "gemm_kernel_ms_per_batch": gemm_ms,
                                  "gemm_launches": gemm_cnt,
                                  "note": "achieved = 22.4 MFLOP/token x 65536 tokens over the wall time of one encode; the matrix pipe "
                                          "executes 3 bf16 terms per product (x3 on the pipe)"}}
    enc.close()
    n = 1_000_000
    xb = synth.normal_f32(21 + rank, n, 384, device, l2_normalise=True)
    ix = search.IndexFlatL2(384)
    ix.attach(xb)
    knn = {}
    for nq in (1, 64, 1024):
        q = synth.normal_f32(22, nq, 384, device, l2_normalise=True)
        for _ in range(2):
            ix.search(q, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ix.search(q, 10)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 5
        knn[f"nq{nq}"] = {"ms_per_search": per * 1e3, "qps": nq / per, "algorithmic_GBps_one_pass": n * 384 * 4 / per / 1e9,
                          "algorithmic_TFLOPs": 2.0 * nq * n * 384 / per / 1e12}
    out["knn_1Mx384_top10"] = knn
    ix.close()
    del xb
    return out


def cfg5_block(args, device, rank, world):
    """BASELINE cfg5 on this GPU's share: IndexIVF-PQ build + search over ivfpq_n x 384 rows per GPU (100 M / 8 = 12.5 M),
    nlist 4096, m 48 (8-bit), nprobe 32 (SURVEY 8d fixes m and nprobe).  Rows are generated on the device (20 000
    clusters + noise; queries = rows + a small perturbation, each with a planted neighbour).  world > 1: the quantisers
    are trained on the union of the shards (one integer all-reduce per k-means iteration), every rank adds and scans its
    own rows, one all-gather merges the top-k."""
    import torch
    import torch.distributed as dist

    from eioku_amd import _lib, ivfpq, search, synth

    n, d, nlist, m, nprobe, nq, k = args.ivfpq_n, 384, 4096, 48, 32, 1024, 10
    ncl, sigma = 20000, 0.02
    centres = synth.normal_f32(5, ncl, d, device, l2_normalise=True)
    assign = torch.randint(0, ncl, (n,), device=device, generator=torch.Generator(device=device).manual_seed(6 + rank))
    xb = torch.empty((n, d), dtype=torch.float32, device=device)
    step = 2_500_000
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        xb[lo:hi] = centres[assign[lo:hi]] + sigma * synth.normal_f32(100 + 16 * rank + lo // step, hi - lo, d, device)
    qa = torch.randint(0, n, (nq,), device=device, generator=torch.Generator(device=device).manual_seed(7))
    q = xb[qa] + 0.1 * sigma * synth.normal_f32(9, nq, d, device)
    if world > 1:  # the same queries on every rank: rank 0's
        dist.broadcast(q, src=0)
    torch.cuda.synchronize()
    ix = ivfpq.IndexIVFPQ(d, nlist, m, device=device)
    t0 = time.perf_counter()
    ix.train(xb, group=dist.group.WORLD if world > 1 else None)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    t0 = time.perf_counter()
    for lo in range(0, n, step):
        ix.add(xb[lo:min(n, lo + step)])
    offsets, sizes, _, _ = ix._pack()
    torch.cuda.synchronize()
    t_add = time.perf_counter() - t0
    ix.nprobe = nprobe
    sh = search.ShardedFlatL2(ix, id_base=rank * n)
    for _ in range(2):
        D, I = sh.search(q, k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    iters = 5
    _lib.prof_enable(True, tags=[_lib.PROF_IVFPQ])
    _lib.prof_reset()
    t0 = time.perf_counter()
    for _ in range(iters):
        D, I = sh.search(q, k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / iters
    _lib.prof_enable(False)
    if world > 1:
        t = torch.tensor([per], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per = float(t.item())
    scan_ms, cnt = _lib.prof_read(_lib.PROF_IVFPQ)
    scan_s = max(scan_ms * 1e-3 / max(cnt, 1), 1e-9)
    stats = ix.last_stats.cpu().tolist()
    _, probes = ix._quantizer.search_many(q, nprobe)
    probed = float(sizes.long()[probes.clamp(min=0)].sum().item()) * m  # bytes a query-major scan of THIS index reads
    alg = float(nq) * nprobe * (float(n) / nlist) * m                    # nprobe (N / nlist) m bytes per query
    planted = (I[:, 0] == qa + 0).float().mean().item() if world == 1 else None
    out = {"metric": f"IVF-PQ QPS@top-10, {n} x {d} rows per GPU (nlist {nlist}, m {m}, nprobe {nprobe})", "value": nq * 1.0 / per,
           "unit": "queries/s", "nq": nq, "k": k, "ms_per_search": per * 1e3, "rows_per_gpu": n, "n_total": n * world,
           "train_s": t_train, "add_s": t_add, "add_vectors_per_s": n / t_add, "dtype": "u8 codes; f32 ADC distances (bf16 MFMA filter + exact fp32 re-rank)",
           "planted_neighbour_first": planted, "overflow_flag": stats[0], "work_items": stats[1], "candidates_per_query": stats[3] / nq,
           "largest_candidate_list": stats[2],
           "collective": "none" if world == 1 else "int64 all-reduce per k-means iteration (build); all_gather of nq*k*16 B per rank (search)",
           "roofline": {"kernel": "k_lscan (list-major ADC scan: a list segment's codes decoded once into bf16 MFMA operands, every probing "
                                  "query streamed past them)", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        "achieved": alg / scan_s / 1e9, "frac": alg / scan_s / 1e9 / HBM_PEAK_GBS, "avg_kernel_ms": scan_s * 1e3,
                        "algorithmic_bytes_per_launch": alg, "code_bytes_streamed_per_launch": float(stats[1]) * 512 * m,
                        "code_bytes_of_the_probed_lists_query_major": probed, "code_bytes_in_index": float(n) * m,
                        "note": "achieved = SURVEY 8d's per-query figure (nprobe x N/nlist x m bytes) x nq over the scan kernel's HIP-event "
                                "duration; the list-major scan reads each probed list's codes once per SEARCH (code_bytes_streamed), so "
                                "the algorithmic figure can exceed what crosses HBM; coarse quantiser, bound pass, inversion, "
                                "re-rank are inside ms_per_search, not in it"}}
    ix._quantizer.close()
    del xb
    return out


def cpu_baseline(args, stages):
    """CPU oracle (numpy / torch-CPU port) on a bounded sample of the same workload: frames/s."""
    import numpy as np
    import torch

    from eioku_amd import embed, weights as W
    from oracle import bert as obert, prng, scene as oscene, yolo as oy

    n = 8  # frames per pass: a batch torch-CPU can spread over its threads
    frames = prng.synth_frames_bgr(1234, n, args.height, args.width)
    net = None
    if "detect" in stages:
        variant, nc, _ = W.variant_from_model_name(args.model)
        # fp32, as the reference predicts on the CPU (model_manager.py:270-275: Ultralytics half=False)
        net = oy.Net(W.random_state(variant, nc, 7), *W.YOLO_VARIANTS[variant], nc, fp16=False)
    bert_state = embed.random_state(embed.MINILM_L6_V2, 11) if "embed" in stages else None
    seg_per_frame = args.segments / args.batch
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    done = 0
    seg_debt = 0.0
    while True:
        if "scene" in stages:
            oscene.content_sums(frames)
            oscene.luma_sad(np.ascontiguousarray(frames[..., 1]))
        if net is not None:
            oy.detect(net, frames, args.conf)
        if bert_state is not None:
            seg_debt += seg_per_frame * n
            while seg_debt >= 1.0:  # same segments-per-frame ratio as the GPU step
                ids = rng.integers(1000, 30000, (1, args.seq_len)).astype(np.int32)
                obert.encode(bert_state, embed.MINILM_L6_V2, ids, np.ones_like(ids, dtype=np.uint8), dtype=np.float32)
                seg_debt -= 1.0
        done += n
        if time.perf_counter() - t0 > args.cpu_seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle stages {'+'.join(stages)} (fp32 network, the reference's CPU precision) on {n} synthetic {args.height}x{args.width} frames per pass "
                      f"({seg_per_frame:.3f} segments/frame), {done} frames in {dt:.1f}s; numpy + torch-CPU "
                      f"({torch.get_num_threads()} threads)"}


def main():
    args = parse_args()
    if launcher_decision(args.gpus, os.environ) == "spawn":  # before torch / the library / any HIP call
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    import copy

    import torch
    import torch.distributed as dist

    from eioku_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world  # the job's size is what the launcher made it
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    _lib.init(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP-event hooks bracket every tagged launch on the kernel's own stream; they are sampled (every
    # --prof-every-th step of the timed region, run un-overlapped) because 2 event records per launch x ~130
    # launches per step are themselves ~15% of a step and because overlapped stages stretch each other's kernels
    def profiled(i, args=args):
        if args.prof_every < 0:
            # the middle step.  (Profiling the FIRST timed step instead - the pipeline is empty there anyway, so no second
            # drain / refill - gains 1-2 % on `value` at 20 steps but measures the kernels right behind the idle barrier:
            # roofline.frac 0.117 instead of 0.124 on the same box; the kernel figure is the one to keep honest.)
            return i == args.steps // 2
        return args.prof_every > 0 and i % args.prof_every == 0

    def run_frames(height, width, args=args):
        """warm-up, then EXACTLY args.steps timed steps between barrier + synchronize pairs; max over ranks."""
        pipe = Pipeline(args, device, rank, height, width)
        for i in range(args.warmup):
            pipe.step(i)
        pipe.finish()  # the timed region starts with an empty segment batch
        barrier()
        seg0 = pipe.batcher.encoded if pipe.batcher is not None else 0
        _lib.prof_reset()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if profiled(i, args):
                torch.cuda.synchronize()  # drain the overlapped steps, time this one's kernels alone, drain again
                _lib.prof_enable(True, tags=[_lib.PROF_CONV, _lib.PROF_SCENE_HSV, _lib.PROF_SCENE_SAD])
                pipe.step(i, serial=True)
                torch.cuda.synchronize()
                _lib.prof_enable(False)
            else:
                pipe.step(i)
        pipe.finish()  # the segments of the last steps, still inside the timed region
        barrier()
        elapsed = time.perf_counter() - t0
        if pipe.batcher is not None:
            assert pipe.batcher.encoded - seg0 == args.segments * args.steps, "every step's segments are encoded in the timed region"
        _lib.prof_enable(False)
        pipe.prof_steps = sum(1 for i in range(args.steps) if profiled(i, args))
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        dom = pipe.dominant()
        # HBM traffic of the dominant kernel family from PMC counters (tools/pmc_traffic.sh: separate FETCH_SIZE /
        # WRITE_SIZE passes, gfx950 x2 read correction), measured on this exact workload and committed together with the
        # hash of the kernel sources it was taken on; a measurement of other code is refused, not reported
        traffic, tnote = None, None
        tfile = ROOT / "profiles" / f"traffic_conv_{Path(args.model).stem}_{args.batch}x{height}x{width}.json"
        if pipe.det is not None and tfile.exists():
            t = json.loads(tfile.read_text())
            if t.get("csrc_sha16") == kernel_source_hash():
                traffic = t["hbm_bytes_per_forward"]
                tnote = "HBM bytes per step (all conv launches), PMC FETCH_SIZE x2 + WRITE_SIZE, taken on these kernel sources"
            else:
                tnote = f"{tfile.name} was measured on other kernel sources ({t.get('csrc_sha16')}): refused"
        avg_ms = dom["ms_total"] / max(dom["launches"], 1)
        achieved = dom["alg_total"] / (dom["ms_total"] * 1e-3) / dom["scale"] if dom["ms_total"] > 0 else 0.0
        note = "" if not pipe.missing else f"; stages NOT run (value is not the full metric): {', '.join(pipe.missing)}"
        res = {
            "value": args.batch * args.steps * world / elapsed,
            "ms_per_step": elapsed / args.steps * 1e3,
            "dtype": "f16" if pipe.det is not None else "u8",
            "config": {"workload": f"{args.batch}x{height}x{width} BGR u8 frames/step/GPU resident in HBM: "
                                   f"scene (HSV ContentDetector sums + luma SAD) on every frame, {args.model} (random-init, "
                                   f"fp16) detect on every frame, all-MiniLM-L6-v2 (random-init, fp32) on {args.segments} "
                                   f"segments x {args.seq_len} tokens per step, collected on the device and encoded "
                                   f"{max(args.segments, args.embed_batch)} segments at a time (the rest inside the timed region); "
                                   f"stages run: {', '.join(pipe.stages)}{note}",
                       "batch": args.batch, "frame": [height, width], "parallelism": f"shard-by-video x{world}"},
            "roofline": {"bound": dom["bound"], "kernel": dom["kernel"], "achieved": achieved, "peak": dom["peak"],
                         "unit": dom["unit"], "frac": achieved / dom["peak"], "traffic": traffic, "traffic_note": tnote,
                         "traffic_per_launch": (traffic / dom["launches"] * max(pipe.prof_steps, 1)) if traffic and dom["launches"] else None,
                         "avg_kernel_ms": avg_ms, "launches": dom["launches"],
                         "kernel_ms_per_step": dom["ms_total"] / max(pipe.prof_steps, 1), "profiled_steps": pipe.prof_steps,
                         "algorithmic_per_step": dom["alg_per_step"], "scene_kernels": pipe.scene_kernels()},
        }
        stages = list(pipe.stages)
        pipe.close()
        del pipe
        torch.cuda.empty_cache()
        return res, stages

    head, stages = run_frames(args.height, args.width)
    out = {
        "metric": "frames/sec (scene+detect+embed) per node",
        "value": head["value"],
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": head["dtype"],
        "data": "synthetic",
        "config": head["config"],
        "roofline": head["roofline"],
    }
    if not args.no_1080p and (args.height, args.width) != (1080, 1920):
        # the north star's own source size: 1080p frames, bilinear letterbox to 384 x 640 (the reference's
        # model_manager.py:263-275 on a 1920 x 1080 file); same three stages, same steps / warm-up
        blk, _ = run_frames(1080, 1920)
        out["frames_1080p"] = {"metric": out["metric"], "unit": "frames/s", "steps": args.steps, "warmup": args.warmup, **blk}
    if args.knn_n > 0:
        out["knn"] = knn_part(args, device, rank, world)

    def guarded(name, fn):
        """a side block must not take the headline down with it (its error is part of the line instead)"""
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    if not args.no_cfg3:
        guarded("cfg3", lambda: cfg3_block(args, device, rank, world))
    if not args.no_cfg4:
        def cfg4():
            blk = {"config": "BASELINE cfg4's models on 64 x 1080p sources per step per GPU (scene + detect + embed each); "
                             "random-init weights, head calibrated to ~1 % of anchors above conf"}
            for key, model in (("objects_yolov8m", "yolov8m.pt"), ("faces_yolov8n_face", "yolov8n-face.pt")):
                a4 = copy.copy(args)
                a4.model, a4.steps, a4.prof_every = model, args.cfg_steps, -1
                r, _ = run_frames(1080, 1920, a4)
                blk[key] = {"metric": out["metric"], "unit": "frames/s", "steps": a4.steps, "warmup": a4.warmup, **r}
            return blk
        guarded("cfg4", cfg4)
    if not args.no_cfg5:
        guarded("cfg5", lambda: cfg5_block(args, device, rank, world))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, stages)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
