"""GPU: the drop-in ModelManager end to end on raw clips (HIP kernels underneath, no stubs)."""
import asyncio
import json

import numpy as np
import pytest

from oracle import prng, scene as oscene
from eioku_amd import detect as D
from eioku_amd.model_manager import ModelManager
from eioku_amd import task_handler

pytestmark = pytest.mark.gpu


def _write_y4m(path, luma, fps=(30000, 1001)):
    n, h, w = luma.shape
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F{fps[0]}:{fps[1]} Ip A1:1 C420jpeg\n".encode())
        chroma = np.full(((h + 1) // 2) * ((w + 1) // 2) * 2, 128, np.uint8).tobytes()
        for t in range(n):
            f.write(b"FRAME\n")
            f.write(luma[t].tobytes())
            f.write(chroma)


def test_detect_scenes_ffmpeg_semantics_on_y4m(gpu, tmp_path):
    frames = prng.synth_frames_bgr(1234, 230, 48, 64)  # scene change at frame 198
    luma = np.ascontiguousarray(frames[..., 1])
    p = tmp_path / "clip.y4m"
    _write_y4m(p, luma)
    mm = ModelManager(cache_dir=str(tmp_path / "m"))
    for thr in (0.05, 0.3, 0.7):
        got = asyncio.run(mm.detect_scenes(str(p), {"threshold": thr, "min_scene_length": 2.0}))
        want = oscene.detect_scenes_ffmpeg_like(luma, thr, 1001, 30000, 230 / (30000 / 1001))
        assert got == want, thr
    assert asyncio.run(mm.detect_scenes(str(p), {"threshold": 0.05}))["scenes"][0]["start_ms"] == int(float("%.6g" % (198 * 1001 / 30000)) * 1000)


def test_detect_scenes_content_mode_and_npy_source(gpu, tmp_path):
    frames = prng.synth_frames_bgr(1234, 230, 48, 64)
    p = tmp_path / "clip.npy"
    np.save(p, frames)
    (tmp_path / "clip.npy.json").write_text(json.dumps({"fps": 30.0, "time_base": [1, 30], "duration": 230 / 30}))
    mm = ModelManager(cache_dir=str(tmp_path / "m"))
    got = asyncio.run(mm.detect_scenes(str(p), {"detector": "content", "min_scene_len": 15}))
    cuts = oscene.content_cuts(oscene.content_scores(oscene.content_sums(frames), 48 * 64))
    assert cuts == [198]
    assert [s["start_ms"] for s in got["scenes"]] == [0, 6600] and got["scenes"][-1]["end_ms"] == 7666
    assert [s["scene_index"] for s in got["scenes"]] == [0, 1]
    # reference (ffmpeg) mode on the same clip: BT.601 luma of the raw frames
    got = asyncio.run(mm.detect_scenes(str(p), {"threshold": 0.05}))
    y = mm._open(str(p)).luma_planes(0, 230)
    assert got == oscene.detect_scenes_ffmpeg_like(y, 0.05, 1, 30, 230 / 30)


def test_detect_objects_and_faces_on_raw_clip(gpu, tmp_path):
    frames = prng.synth_frames_bgr(21, 75, 120, 160)
    p = tmp_path / "clip.npy"
    np.save(p, frames)
    (tmp_path / "clip.npy.json").write_text(json.dumps({"fps": 29.97}))
    mm = ModelManager(cache_dir=str(tmp_path / "m"), random_init_seed=7, batch_size=2)
    res = asyncio.run(mm.detect_objects(str(p), {"frame_interval": 1, "confidence_threshold": 0.25}))
    stride = int(29.97 * 1)
    sampled = list(range(0, 75, stride))
    det = D.Yolov8Detector.from_model_name("yolov8n.pt", seed=7)
    dets, counts = det.detect(frames[sampled], conf=0.25)
    want = []
    for fi, row, c in zip(sampled, dets, counts):
        for d in row[:c]:
            want.append({"frame_index": fi, "timestamp_ms": int((fi / 29.97) * 1000), "label": det.names[int(d["cls"])],
                         "confidence": float(d["conf"]),
                         "bbox": {"x": float(d["x1"]), "y": float(d["y1"]), "width": float(np.float32(d["x2"] - d["x1"])),
                                  "height": float(np.float32(d["y2"] - d["y1"]))}})
    assert res == {"detections": want}
    assert all(type(x["confidence"]) is float and type(x["frame_index"]) is int for x in res["detections"])
    json.dumps(res)  # the handler serialises every detection
    faces = asyncio.run(mm.detect_faces(str(p), {"confidence_threshold": 0.3}))
    assert all(f["label"] == "face" and f["cluster_id"] is None and f["confidence"] >= 0.3 for f in faces["detections"])
    assert {f["frame_index"] for f in faces["detections"]} <= {0}  # default 3 s stride at 29.97 fps -> frame 0 only (75 frames)
    det.close()


def test_process_ml_task_end_to_end(gpu, tmp_path, monkeypatch):
    frames = prng.synth_frames_bgr(1234, 230, 48, 64)
    p = tmp_path / "clip.npy"
    np.save(p, frames)
    monkeypatch.setenv("MODEL_CACHE_DIR", str(tmp_path / "models"))
    sink = []
    out = asyncio.run(task_handler.process_ml_task({"artifact_sink": sink.extend}, "task-1", "scene_detection", "vid-1", str(p),
                                                   {"threshold": 0.05}))
    assert out == {"task_id": "task-1", "status": "completed", "artifact_count": len(sink)} and len(sink) >= 1
    assert all(e.artifact_type == "scene" and e.span_start_ms <= e.span_end_ms for e in sink)
    with pytest.raises(RuntimeError, match="Failed to process task task-2"):
        asyncio.run(task_handler.process_ml_task({}, "task-2", "object_detection", "vid-1", str(p), {}))  # no weights on disk
