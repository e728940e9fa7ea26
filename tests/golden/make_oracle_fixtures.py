#!/usr/bin/env python3
"""Small golden vectors produced by the CPU oracle (regression pins for oracle AND kernels).

Run:  python tests/golden/make_oracle_fixtures.py
These are self-generated (the reference holds no vectors for the numeric stages);
they keep the oracle from drifting silently and give the GPU tests committed expected outputs.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

from oracle import prng, scene  # noqa: E402


def scene_fixtures():
    # 32 frames 64x48 with 3 planted cuts (params table switched by hand)
    n, h, w = 32, 48, 64
    params = np.zeros((n, 5), dtype=np.int32)
    bases = [(40, 90, 200, 3, 17), (220, 60, 35, 25, 2), (128, 128, 128, 0, 0), (33, 222, 120, 31, 31)]
    cuts = [7, 16, 25]
    for t in range(n):
        params[t] = bases[sum(t >= c for c in cuts)]
    frames = prng.synth_frames_bgr(99, n, h, w, params=params)
    luma = np.ascontiguousarray(frames[..., 1])  # any plane serves as "Y" for the SAD kernel
    sad = scene.luma_sad(luma)
    mafd, score = scene.ffmpeg_scene_scores(sad, h * w)
    np.savez_compressed(HERE / "scene_luma_64x48.npz", luma=luma, sad=sad, mafd=mafd, score=score,
                        cuts_t03=scene.select_scene_cuts(score, 0.3), planted=np.array(cuts))
    sums = scene.content_sums(frames)
    cs = scene.content_scores(sums, h * w)
    np.savez_compressed(HERE / "scene_hsv_64x48.npz", frames=frames, sums=sums, score=cs,
                        cuts_legacy=np.array(scene.content_cuts(cs, 27.0, 5), dtype=np.int64),
                        cuts_merge=np.array(scene.content_cuts(cs, 27.0, 5, mode="merge"), dtype=np.int64))
    # HSV lattice: every 9th level of each channel + the extremes
    lv = np.unique(np.concatenate([np.arange(0, 256, 9), [255, 254, 1, 128]])).astype(np.uint8)
    b, g, r = np.meshgrid(lv, lv, lv, indexing="ij")
    bgr = np.stack([b, g, r], axis=-1).reshape(-1, 3)
    np.savez_compressed(HERE / "hsv_lattice.npz", bgr=bgr, hsv=scene.bgr2hsv_u8(bgr))


def unit_rows(seed, n, d):
    """Unit-norm rows from the portable PRNG (float64 norm, so the bytes do not depend on BLAS)."""
    x = prng.approx_normal_f32(seed, n * d).reshape(n, d)
    return (x / np.sqrt((x.astype(np.float64) ** 2).sum(1, keepdims=True))).astype(np.float32)


def knn_fixtures():
    from oracle import knn

    xb, xq = unit_rows(21, 4096, 384), unit_rows(22, 16, 384)
    D, I = knn.search(xb, xq, 10)
    # inputs are regenerated from the seeds; a checksum guards against PRNG drift
    np.savez_compressed(HERE / "flatl2_4096x384.npz", seed_db=21, seed_q=22, n=4096, nq=16, d=384, D=D, I=I,
                        xb_sum=np.float64(xb.astype(np.float64).sum()), xq_first=xq[0, :8])


if __name__ == "__main__":
    scene_fixtures()
    knn_fixtures()
    print("wrote oracle fixtures to", HERE)
