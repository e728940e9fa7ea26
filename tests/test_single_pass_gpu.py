"""GPU: ModelManager.analyze_video - scenes + objects + faces from ONE read of the file, frames uploaded once -
returns, per task, exactly what the three separate calls return."""
import asyncio
import json
import sys

import numpy as np
import pytest

from oracle import prng, scene as oscene
from eioku_amd import frames as F, scene
from eioku_amd.model_manager import ModelManager
from test_frames_cv2 import make_cv2

pytestmark = pytest.mark.gpu


def test_luma_sad_bgr_is_opencv_luma_then_k1(gpu):
    import torch

    for n, h, w in ((9, 48, 64), (20, 30, 34), (3, 1080, 1920)):
        f = prng.synth_frames_bgr(7, n + 1, h, w)
        y = F.bgr_to_luma_bt601(f)
        want = oscene.luma_sad(y[1:], y[0])
        got = scene.luma_sad_bgr(f[1:], f[0])
        got_dev = scene.luma_sad_bgr(torch.from_numpy(f[1:]).to(gpu), torch.from_numpy(f[0]).to(gpu))
        assert np.array_equal(got, want) and np.array_equal(got_dev, want)
        assert np.array_equal(scene.luma_sad_bgr(f)[1:], want) and scene.luma_sad_bgr(f)[0] == 0


CONFIGS = {"scene_detection": {"threshold": 0.05}, "object_detection": {"frame_interval": 1, "confidence_threshold": 0.25},
           "face_detection": {"frame_interval": 2, "confidence_threshold": 0.3}}


def _separately(mm, path):
    return {"scene_detection": asyncio.run(mm.detect_scenes(path, CONFIGS["scene_detection"])),
            "object_detection": asyncio.run(mm.detect_objects(path, CONFIGS["object_detection"])),
            "face_detection": asyncio.run(mm.detect_faces(path, CONFIGS["face_detection"]))}


def test_single_pass_equals_three_separate_calls_on_a_raw_clip(gpu, tmp_path):
    frames = prng.synth_frames_bgr(1234, 230, 120, 160)  # scene change at frame 198
    p = tmp_path / "clip.npy"
    np.save(p, frames)
    (tmp_path / "clip.npy.json").write_text(json.dumps({"fps": 29.97}))
    mm = ModelManager(cache_dir=str(tmp_path / "m"), random_init_seed=7, batch_size=48)
    want = _separately(mm, str(p))
    got = asyncio.run(mm.analyze_video(str(p), CONFIGS))
    assert got == want
    assert len(got["scene_detection"]["scenes"]) >= 1 and len(got["object_detection"]["detections"]) > 0
    json.dumps(got)
    # a subset of the tasks, and the ContentDetector flavour of the scene task
    only = asyncio.run(mm.analyze_video(str(p), {"scene_detection": {"detector": "content", "min_scene_len": 15}}))
    assert only == {"scene_detection": asyncio.run(mm.detect_scenes(str(p), {"detector": "content", "min_scene_len": 15}))}
    with pytest.raises(NotImplementedError):
        asyncio.run(mm.analyze_video(str(p), {"transcription": {}}))


def test_single_pass_on_a_cv2_capture_reads_each_frame_once(gpu, monkeypatch, tmp_path):
    bgr = prng.synth_frames_bgr(1234, 230, 48, 64)
    cv2 = make_cv2(bgr, np.ascontiguousarray(bgr[..., 1]), 30.0, honour_convert_rgb=False)
    monkeypatch.setitem(sys.modules, "cv2", cv2)
    mm = ModelManager(cache_dir=str(tmp_path / "m"), random_init_seed=7, batch_size=64)
    want = _separately(mm, "/videos/clip.mp4")
    opened_before = len(cv2.opened)
    got = asyncio.run(mm.analyze_video("/videos/clip.mp4", CONFIGS))
    assert got == want
    mine = cv2.opened[opened_before:]
    # ONE capture decodes the file, every frame read exactly once (+ the one-frame probe that asked the backend for the
    # decoder's planes and was answered with BGR: this stub ignores CAP_PROP_CONVERT_RGB)
    assert len(mine) == 2 and mine[0].pos == 230 and mine[0].released and mine[1].pos == 1 and mine[1].released
    # the three separate calls opened 4 captures (objects, faces, scenes + its luma capture) and decoded 3 x 230 frames
    assert opened_before >= 3
