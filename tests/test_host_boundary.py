"""CPU tests of the drop-in boundary (host logic only, no GPU): ModelManager.detect_objects/faces
against the dicts captured from the reference's own loop, process_ml_task's contract, envelope rules."""
import asyncio
import json

import numpy as np
import pytest

from conftest import GOLDEN
from eioku_amd import task_handler
from eioku_amd.detect import DET_DTYPE
from eioku_amd.frames import FrameSource
from eioku_amd.model_manager import ModelManager

CASES = json.loads((GOLDEN / "ref_detect_loop.json").read_text())


class ScriptedSource(FrameSource):
    """Frames whose pixel (0,0) carries the frame number, like the capture script's fake VideoCapture."""

    def __init__(self, fps, total):
        self.fps, self.total_frames, self.pos = fps, total, 0

    def read(self):
        if self.pos >= self.total_frames:
            return False, None
        f = np.zeros((4, 4, 3), np.uint8)
        f[0, 0] = [self.pos & 255, (self.pos >> 8) & 255, (self.pos >> 16) & 255]
        self.pos += 1
        return True, f

    def grab(self):
        if self.pos >= self.total_frames:
            return False
        self.pos += 1
        return True


class ScriptedDetector:
    """Stands in for the HIP detector: replays the raw boxes the capture script fed the reference,
    applying the predictor's own `conf` filter (strict >, float32)."""

    def __init__(self, case):
        self.by_frame = {c["frame_index"]: c["boxes"] for c in case["detector_calls"]}
        self.names = {int(k): v for k, v in case["names"].items()}
        self.batches = []

    def detect(self, frames, conf):
        self.batches.append(len(frames))
        dets = np.zeros((len(frames), 300), dtype=DET_DTYPE)
        counts = np.zeros(len(frames), np.int32)
        for i, f in enumerate(frames):
            idx = int(f[0, 0, 0]) | (int(f[0, 0, 1]) << 8) | (int(f[0, 0, 2]) << 16)
            k = 0
            for b in self.by_frame[idx]:
                if np.float32(b["conf"]) > np.float32(conf):
                    dets[i, k] = (*[np.float32(v) for v in b["xyxy"]], np.float32(b["conf"]), b["cls"], 0, 0)
                    k += 1
            counts[i] = k
        return dets, counts


@pytest.mark.parametrize("case", CASES, ids=[f"{c['kind']}-{c['fps']}-{c['total_frames']}" for c in CASES])
@pytest.mark.parametrize("batch", [1, 3, 64])
def test_detect_loop_equals_reference_capture(case, batch, tmp_path):
    det = ScriptedDetector(case)
    mm = ModelManager(cache_dir=str(tmp_path), frame_source=lambda p: ScriptedSource(case["fps"], case["total_frames"]),
                      detector_factory=lambda name, cache: det, batch_size=batch)
    fn = mm.detect_objects if case["kind"] == "objects" else mm.detect_faces
    got = asyncio.run(fn("/videos/fake.mp4", dict(case["config"])))
    assert got == case["result"]  # dict equality: ints, strings and every float bit for bit
    assert json.dumps(got) == json.dumps(case["result"])
    assert sum(det.batches) == len(case["detector_calls"])  # same frames sampled, only batched differently
    assert all(b <= batch for b in det.batches)


def test_model_manager_gpu_probe_surface(tmp_path):
    mm = ModelManager(cache_dir=str(tmp_path / "models"))
    assert (tmp_path / "models").is_dir()
    assert mm._get_device() in ("cuda", "cpu") and isinstance(mm.detect_gpu(), bool)
    assert set(mm.get_gpu_info()) == {"gpu_available", "gpu_device_name", "gpu_memory_total_mb", "gpu_memory_used_mb"}
    with pytest.raises(NotImplementedError):
        asyncio.run(mm.transcribe_video("x", {}))


def test_missing_weights_fail_loudly(tmp_path):
    mm = ModelManager(cache_dir=str(tmp_path), frame_source=lambda p: ScriptedSource(30.0, 5))
    with pytest.raises(Exception) as e:
        asyncio.run(mm.detect_objects("/videos/fake.mp4", {}))
    assert "not found" in str(e.value) or "eioku" in str(e.value).lower()


def test_envelope_rules_match_reference_capture():
    from datetime import datetime

    for row in json.loads((GOLDEN / "ref_artifact_spans.json").read_text()):
        try:
            task_handler.ArtifactEnvelope(created_at=datetime(2026, 1, 1), **row["fields"])
            ok, err = True, None
        except Exception as e:  # noqa: BLE001
            ok, err = False, type(e).__name__
        assert (ok, err) == (row["accepted"], row["error_type"]), row["fields"]


class FakeManager:
    def __init__(self, cache_dir="/models", result=None, exc=None):
        self.result, self.exc = result, exc

    async def detect_objects(self, path, config):
        if self.exc:
            raise self.exc
        return self.result

    detect_faces = detect_scenes = detect_objects


class Store:
    def __init__(self):
        self.log = []

    def mark_running(self, t):
        self.log.append(("running", t))

    def mark_completed(self, t):
        self.log.append(("completed", t))

    def mark_failed(self, t, e):
        self.log.append(("failed", t, e))

    def mark_cancelled(self, t):
        self.log.append(("cancelled", t))


def test_process_ml_task_contract():
    case = CASES[1]
    sink, store = [], Store()
    ctx = {"artifact_sink": sink.extend, "task_store": store,
           "model_manager_factory": lambda cache_dir: FakeManager(result=case["result"])}
    out = asyncio.run(task_handler.process_ml_task(ctx, "t1", "object_detection", "vid9", "/v.mp4", {"frame_interval": 3.0}))
    n = len(case["result"]["detections"])
    assert out == {"task_id": "t1", "status": "completed", "artifact_count": n}
    assert store.log == [("running", "t1"), ("completed", "t1")]
    assert len(sink) == n
    e = sink[0]
    d = case["result"]["detections"][0]
    assert e.artifact_type == "object.detection" and e.asset_id == "vid9" and e.schema_version == 1
    assert e.span_start_ms == e.span_end_ms == d["timestamp_ms"]
    assert json.loads(e.payload_json) == d
    assert e.artifact_id.startswith("vid9_object_detection_") and e.artifact_id.endswith("_0")
    assert (e.producer, e.producer_version, e.model_profile, e.config_hash, e.input_hash) == ("ml-service", "1.0.0", "balanced", "", "")
    # scenes use explicit spans; inverted / negative spans are dropped, not fatal (ref :296-308)
    scenes = {"scenes": [{"scene_index": 0, "start_ms": 100, "end_ms": 50, "duration_ms": -50},
                         {"scene_index": 1, "start_ms": 50, "end_ms": 80, "duration_ms": 30}]}
    sink.clear()
    ctx["model_manager_factory"] = lambda cache_dir: FakeManager(result=scenes)
    out = asyncio.run(task_handler.process_ml_task(ctx, "t2", "scene_detection", "vid9", "/v.mp4", None))
    assert out["artifact_count"] == 1 and sink[0].artifact_type == "scene" and (sink[0].span_start_ms, sink[0].span_end_ms) == (50, 80)


def test_process_ml_task_errors():
    store = Store()
    ctx = {"task_store": store, "model_manager_factory": lambda cache_dir: FakeManager(exc=ValueError("boom"))}
    with pytest.raises(RuntimeError, match="Failed to process task t3: boom"):
        asyncio.run(task_handler.process_ml_task(ctx, "t3", "object_detection", "v", "/v.mp4", {}))
    assert store.log[-1] == ("failed", "t3", "boom")
    with pytest.raises(RuntimeError, match="Unknown task type: nonsense"):
        asyncio.run(task_handler.process_ml_task(ctx, "t4", "nonsense", "v", "/v.mp4", {}))
    with pytest.raises(RuntimeError, match="outside the MI355X hot path"):
        asyncio.run(task_handler.process_ml_task(ctx, "t5", "ocr", "v", "/v.mp4", {}))
    ctx["model_manager_factory"] = lambda cache_dir: FakeManager(exc=asyncio.CancelledError())
    with pytest.raises(asyncio.CancelledError):
        asyncio.run(task_handler.process_ml_task(ctx, "t6", "face_detection", "v", "/v.mp4", {}))
    assert store.log[-1] == ("cancelled", "t6")
