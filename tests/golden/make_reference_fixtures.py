#!/usr/bin/env python3
"""Capture golden vectors from the reference's OWN Python code (build container only).

Run:  python tests/golden/make_reference_fixtures.py         (needs /root/reference)

What is pinned here is the reference's orchestration around the third-party calls - the part of
the hot path that exists as code under /root/reference:

  ref_detect_loop.json   ModelManager.detect_objects / detect_faces
                         (ml-service/src/services/model_manager.py:215-407) driven with stub
                         ``cv2`` / ``ultralytics`` modules injected through ``sys.modules``: frame
                         sampling ``max(1, int(fps*sec))``, ``int(idx/fps*1000)`` timestamps, the
                         result dict, float32 -> Python float widening, fp32 width/height
                         subtraction, the face path's extra confidence filter.
  ref_scenes.json        ModelManager.detect_scenes (model_manager.py:715-835) with
                         ``subprocess.run`` replaced by scripted ffmpeg/ffprobe output: pts_time
                         parsing and the scene-list index quirk.
  ref_artifact_spans.json the span / validation rules of the result -> ArtifactEnvelope mapping
                         (ml-service/src/domain/artifacts.py:7-73) for rows produced by the path.
  ref_projection_rows.json ProjectionSyncService.sync_artifact (services/projection_sync_service.py:26-330) run,
                         one artifact at a time as task_handler.py:392-404 does, on an in-memory SQLite with the
                         projection tables of backend/alembic (scene_ranges / object_labels / face_clusters): the
                         envelopes fed in and the table rows that result - what the batched writer must reproduce.

  ref_places_loop.json   ModelManager.classify_places (model_manager.py:560-713) with stub ``cv2`` /
                         ``torchvision`` modules (the real ``torch`` and ``PIL``): frame sampling, timestamps,
                         label-file parsing (and the generic fallback), softmax -> descending sort -> top_k,
                         ``float(probs[j])`` widening.  The stub model returns scripted logits
                         (``places_logits(seed, frame_index)``); the network's arithmetic is NOT pinned by it.

Only inputs and outputs are stored (JSON data); no reference source text is copied.
The stubs stand in for cv2 / ultralytics / ffmpeg *calls*, i.e. the inputs of the orchestration
under test; the arithmetic inside those libraries is NOT pinned by these fixtures.
"""

from __future__ import annotations

import asyncio
import json
import struct
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/ml-service")


def f32_bits(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def script_boxes(seed: int, frame_idx: int) -> list[dict]:
    """Deterministic raw detector output for one frame: fp32 xyxy, conf, cls."""
    rng = np.random.default_rng(seed * 1_000_003 + frame_idx)
    k = int(rng.integers(0, 4))
    out = []
    for _ in range(k):
        x1, y1 = rng.uniform(0, 1500, 2).astype(np.float32)
        w, h = rng.uniform(5, 400, 2).astype(np.float32)
        conf = np.float32(rng.uniform(0.3, 0.99))
        out.append({"xyxy": [float(x1), float(y1), float(np.float32(x1 + w)), float(np.float32(y1 + h))],
                    "conf": float(conf), "cls": int(rng.integers(0, 80))})
    return out


def install_stubs(fps: float, total: int, seed: int, names: dict[int, str]):
    cv2 = types.ModuleType("cv2")
    cv2.CAP_PROP_FPS = 5
    cv2.CAP_PROP_FRAME_COUNT = 7

    class VideoCapture:
        def __init__(self, path):
            self.pos = 0

        def get(self, prop):
            return fps if prop == cv2.CAP_PROP_FPS else float(total)

        def read(self):
            if self.pos >= total:
                return False, None
            frame = np.zeros((4, 4, 3), dtype=np.uint8)
            frame[0, 0, 0] = self.pos & 0xFF
            frame[0, 0, 1] = (self.pos >> 8) & 0xFF
            frame[0, 0, 2] = (self.pos >> 16) & 0xFF
            self.pos += 1
            return True, frame

        def grab(self):
            if self.pos >= total:
                return False
            self.pos += 1
            return True

        def release(self):
            pass

    cv2.VideoCapture = VideoCapture

    ultra = types.ModuleType("ultralytics")
    calls = []

    class _Box:
        def __init__(self, b):
            self.xyxy = torch.tensor([b["xyxy"]], dtype=torch.float32)
            self.conf = torch.tensor([b["conf"]], dtype=torch.float32)
            self.cls = torch.tensor([float(b["cls"])], dtype=torch.float32)

    class _Result:
        def __init__(self, boxes):
            self.boxes = [_Box(b) for b in boxes]
            self.names = names

    class YOLO:
        def __init__(self, path):
            self.path = path

        def to(self, device):
            return self

        def __call__(self, frame, conf=0.25, verbose=False, device=None):
            idx = int(frame[0, 0, 0]) | (int(frame[0, 0, 1]) << 8) | (int(frame[0, 0, 2]) << 16)
            boxes = script_boxes(seed, idx)
            calls.append({"frame_index": idx, "conf": conf, "boxes": boxes})
            # the real predictor drops boxes below `conf` before NMS; keep that contract
            return [_Result([b for b in boxes if np.float32(b["conf"]) > np.float32(conf)])]

    ultra.YOLO = YOLO
    sys.modules["cv2"] = cv2
    sys.modules["ultralytics"] = ultra
    return calls


def capture_detect_loop(ModelManager):
    names = {i: f"class{i}" for i in range(80)}
    cases = []
    specs = [
        ("objects", 30.0, 200, {"frame_interval": 1}),
        ("objects", 29.97, 200, {"frame_interval": 3.0, "confidence_threshold": 0.5, "model_name": "yolov8s.pt"}),
        ("objects", 23.976, 97, {"frame_interval": 0.5, "confidence_threshold": 0.6}),
        ("objects", 0.0, 65, {}),  # CAP_PROP_FPS == 0 -> `or 30`
        ("objects", 60.0, 50, {"frame_interval": 0.001}),  # max(1, int(...)) clamp
        ("faces", 29.97, 200, {}),  # defaults: 3 s, conf 0.7
        ("faces", 25.0, 120, {"frame_interval": 2, "confidence_threshold": 0.9}),
        ("objects", 30.0, 0, {}),  # empty video
    ]
    for seed, (kind, fps, total, config) in enumerate(specs, start=1):
        calls = install_stubs(fps, total, seed, names)
        with tempfile.TemporaryDirectory() as td:
            mm = ModelManager(cache_dir=td)
            fn = mm.detect_objects if kind == "objects" else mm.detect_faces
            result = asyncio.run(fn("/videos/fake.mp4", dict(config)))
        cases.append({"kind": kind, "fps": fps, "total_frames": total, "config": config, "seed": seed,
                      "names": {str(k): v for k, v in names.items()},
                      "detector_calls": calls, "result": result})
    return cases


def places_logits(seed: int, frame_idx: int) -> np.ndarray:
    """Scripted network output of one frame (float32 [365]); tests regenerate it from (seed, frame_index)."""
    rng = np.random.default_rng(seed * 1_000_003 + frame_idx)
    return (3.0 * rng.standard_normal(365)).astype(np.float32)


def install_places_stubs(fps: float, total: int, seed: int):
    install_stubs(fps, total, seed, {})
    cv2 = sys.modules["cv2"]
    cv2.COLOR_BGR2RGB = 4
    cv2.cvtColor = lambda frame, code: np.ascontiguousarray(frame[..., ::-1])
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    transforms = types.ModuleType("torchvision.transforms")
    calls = []

    class _Model:
        def __init__(self):
            self.fc = types.SimpleNamespace(in_features=512)

        def load_state_dict(self, sd):
            calls.append({"load_state_dict": sorted(sd)[:3]})

        def to(self, device):
            return self

        def eval(self):
            return self

        def __call__(self, x):  # x: (1, 3, 4, 4) uint8-valued float tensor carrying the frame index in pixel (0, 0)
            px = x[0, :, 0, 0]
            idx = int(px[2]) | (int(px[1]) << 8) | (int(px[0]) << 16)  # RGB order after the BGR2RGB stub
            calls.append({"frame_index": idx})
            return torch.from_numpy(places_logits(seed, idx))[None]

    models.resnet18 = lambda pretrained=False: _Model()

    class Compose:
        def __init__(self, ts):
            pass

        def __call__(self, img):  # PIL image -> CHW float tensor of its bytes (the real transform is the device's job)
            return torch.from_numpy(np.asarray(img)).permute(2, 0, 1).to(torch.float32)

    transforms.Compose = Compose
    transforms.Resize = lambda size: ("resize", size)
    transforms.ToTensor = lambda: "totensor"
    transforms.Normalize = lambda mean, std: ("normalize", mean, std)
    tv.models, tv.transforms = models, transforms
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models
    sys.modules["torchvision.transforms"] = transforms
    return calls


def capture_places_loop(ModelManager):
    cases = []
    specs = [
        (30.0, 95, {}, True),                                     # defaults: every 30th frame, top_k 5, label file present
        (29.97, 200, {"frame_interval": 3.0, "top_k": 3}, True),
        # (a case without <cache>/places365/categories_places365.txt would not pin the generic place_<i> fallback: the
        # reference then finds the list it ships beside ml-service/src - model_manager.py:583-590 - so every case
        # provides the cache-dir file, the first location looked at)
        (23.976, 61, {"frame_interval": 0.5, "top_k": 1}, True),
        (0.0, 40, {"top_k": 400}, True),                          # fps 0 -> `or 30`; top_k beyond 365 classes
        (25.0, 0, {}, True),                                      # empty video
    ]
    label_lines = [f"/{chr(97 + i % 26)}/label_{i} {i}" for i in range(365)]
    for seed, (fps, total, config, with_labels) in enumerate(specs, start=11):
        calls = install_places_stubs(fps, total, seed)
        with tempfile.TemporaryDirectory() as td:
            if with_labels:
                d = Path(td) / "places365"
                d.mkdir(parents=True)
                (d / "categories_places365.txt").write_text("\n".join(label_lines) + "\n")
            mm = ModelManager(cache_dir=td)
            result = asyncio.run(mm.classify_places("/videos/fake.mp4", dict(config)))
        cases.append({"fps": fps, "total_frames": total, "config": config, "seed": seed, "label_file": with_labels,
                      "label_lines": label_lines if with_labels else None,
                      "model_calls": [c["frame_index"] for c in calls if "frame_index" in c], "result": result})
    return cases


def capture_scenes(ModelManager):
    import subprocess

    cases = []

    def showinfo_line(n, pts_time):
        return (f"[Parsed_showinfo_1 @ 0x5581] n:{n:4d} pts:{n * 512:7d} pts_time:{pts_time:<8} pos:  123 "
                f"fmt:yuv420p sar:1/1 s:854x480 i:P iskey:0 type:P checksum:0A1B2C3D plane_checksum:[0A1B2C3D]")

    specs = [
        {"name": "no_cuts", "pts": [], "duration": "30.033000\n", "config": {}},
        {"name": "one_cut", "pts": ["5.005"], "duration": "30.033000\n", "config": {"threshold": 0.4}},
        {"name": "three_cuts", "pts": ["4.2042", "12.5125", "20.02"], "duration": "30.033000\n",
         "config": {"threshold": 0.7, "min_scene_length": 2.0}},
        {"name": "many_cuts_long", "pts": ["0.0333667", "59.9933", "600.6", "1234.57", "3599.97"],
         "duration": "3600.000000\n", "config": {}},
        {"name": "ffprobe_fails", "pts": ["1.5", "2.75"], "duration": "N/A\n", "config": {}},
        {"name": "malformed_line", "pts": ["1.001", "garbage", "7.5075"], "duration": "10.01\n", "config": {}},
    ]
    real_run = subprocess.run
    for spec in specs:
        lines = ["ffmpeg version 4.4.2 Copyright (c) 2000-2021", "Input #0, mov,mp4, from 'fake.mp4':"]
        for n, p in enumerate(spec["pts"]):
            lines.append(showinfo_line(n, p))
        lines.append("frame=  900 fps=0.0 q=-0.0 Lsize=N/A time=00:00:30.03 bitrate=N/A")
        stderr = "\n".join(lines)
        seen = []

        def fake_run(cmd, capture_output=False, text=False, timeout=None, _spec=spec, _stderr=stderr, _seen=seen):
            _seen.append(list(cmd))
            if cmd[0] == "ffmpeg":
                return types.SimpleNamespace(returncode=0, stdout="", stderr=_stderr)
            return types.SimpleNamespace(returncode=0, stdout=_spec["duration"], stderr="")

        subprocess.run = fake_run
        try:
            with tempfile.TemporaryDirectory() as td:
                mm = ModelManager(cache_dir=td)
                result = asyncio.run(mm.detect_scenes("/videos/fake.mp4", dict(spec["config"])))
        finally:
            subprocess.run = real_run
        cases.append({"name": spec["name"], "config": spec["config"], "ffmpeg_stderr": stderr,
                      "ffprobe_stdout": spec["duration"], "commands": seen, "result": result})
    return cases


def capture_artifact_rules():
    from src.domain.artifacts import ArtifactEnvelope
    from datetime import datetime

    rows = []
    probes = [
        {"span_start_ms": 0, "span_end_ms": 0},
        {"span_start_ms": 1000, "span_end_ms": 1000},
        {"span_start_ms": 5, "span_end_ms": 4},
        {"span_start_ms": -1, "span_end_ms": 4},
        {"span_start_ms": 0, "span_end_ms": 10, "schema_version": 0},
        {"span_start_ms": 0, "span_end_ms": 10, "artifact_id": ""},
        {"span_start_ms": 0, "span_end_ms": 10, "payload_json": ""},
        {"span_start_ms": 0, "span_end_ms": 10, "config_hash": "", "input_hash": ""},
    ]
    base = dict(artifact_id="v_object_detection_r_0", asset_id="v", artifact_type="object.detection",
                schema_version=1, span_start_ms=0, span_end_ms=0, payload_json="{}", producer="ml-service",
                producer_version="1.0.0", model_profile="balanced", config_hash="", input_hash="", run_id="r")
    for p in probes:
        kw = dict(base)
        kw.update(p)
        try:
            ArtifactEnvelope(created_at=datetime(2026, 1, 1), **kw)
            ok, err = True, None
        except Exception as e:  # noqa: BLE001 - record whatever the reference raises
            ok, err = False, type(e).__name__
        rows.append({"fields": kw, "accepted": ok, "error_type": err})
    return rows


def capture_projection_rows():
    """The reference's per-artifact projection upserts on SQLite: inputs (envelope fields) and resulting rows."""
    from datetime import datetime

    from sqlalchemy import create_engine, text
    from sqlalchemy.orm import Session

    from src.domain.artifacts import ArtifactEnvelope
    from src.services.projection_sync_service import ProjectionSyncService

    engine = create_engine("sqlite:///:memory:")
    ddl = {  # columns of backend/alembic/versions/{c6f63e560f88,10a71bc9d989,325d54cd2340}_*.py
        "scene_ranges": "artifact_id TEXT PRIMARY KEY, asset_id TEXT NOT NULL, scene_index INTEGER NOT NULL, start_ms INTEGER NOT NULL, end_ms INTEGER NOT NULL",
        "object_labels": "artifact_id TEXT PRIMARY KEY, asset_id TEXT NOT NULL, label TEXT NOT NULL, confidence REAL NOT NULL, start_ms INTEGER NOT NULL, end_ms INTEGER NOT NULL",
        "face_clusters": "artifact_id TEXT PRIMARY KEY, asset_id TEXT NOT NULL, cluster_id TEXT, confidence REAL NOT NULL, start_ms INTEGER NOT NULL, end_ms INTEGER NOT NULL",
    }
    payloads = [
        ("scene", {"scene_index": 0, "start_ms": 0, "end_ms": 6600, "duration_ms": 6600}, 0, 6600),
        ("scene", {"scene_index": 2, "start_ms": 6600, "end_ms": 7666, "duration_ms": 1066}, 6600, 7666),
        ("scene", {"start_ms": 1, "end_ms": 2}, 1, 2),  # no scene_index -> default 0
        ("object.detection", {"frame_index": 0, "timestamp_ms": 0, "label": "person", "confidence": 0.8999999761581421,
                              "bbox": {"x": 1.5, "y": 2.0, "width": 10.0, "height": 20.0}}, 0, 0),
        ("object.detection", {"frame_index": 89, "timestamp_ms": 2969, "label": "dog", "confidence": 0.5123,
                              "bbox": {"x": 0.0, "y": 0.0, "width": 1.0, "height": 1.0}}, 2969, 2969),
        ("object.detection", {"frame_index": 178, "timestamp_ms": 5939}, 5939, 5939),  # no label / confidence -> "" / 0.0
        ("face.detection", {"frame_index": 0, "timestamp_ms": 0, "label": "face", "confidence": 0.91, "cluster_id": None,
                            "bbox": {"x": 3.0, "y": 4.0, "width": 5.0, "height": 6.0}}, 0, 0),
        ("face.detection", {"frame_index": 90, "timestamp_ms": 3000, "label": "face", "confidence": 0.75, "cluster_id": "c7",
                            "bbox": {"x": 3.0, "y": 4.0, "width": 5.0, "height": 6.0}}, 3000, 3000),
        ("object.detection", {"frame_index": 0, "timestamp_ms": 0, "label": "cat", "confidence": 0.66}, 0, 0),  # same id as row 3: upsert
        ("segment.embedding", {"text": "hello", "embedding": [0.1, 0.2]}, 10, 20),  # no projection table
    ]
    ids = ["vid_scene_r_0", "vid_scene_r_1", "vid_scene_r_2", "vid_object_detection_r_0", "vid_object_detection_r_1",
           "vid_object_detection_r_2", "vid_face_detection_r_0", "vid_face_detection_r_1", "vid_object_detection_r_0", "vid_seg_r_0"]
    envs = []
    with Session(engine) as session:
        for table, cols in ddl.items():
            session.execute(text(f"CREATE TABLE {table} ({cols})"))
        svc = ProjectionSyncService(session)
        for aid, (atype, payload, a, b) in zip(ids, payloads):
            env = ArtifactEnvelope(artifact_id=aid, asset_id="vid", artifact_type=atype, schema_version=1, span_start_ms=a,
                                   span_end_ms=b, payload_json=json.dumps(payload), producer="ml-service", producer_version="1.0.0",
                                   model_profile="balanced", config_hash="", input_hash="", run_id="r", created_at=datetime(2026, 1, 28))
            svc.sync_artifact(env)  # one statement per artifact, as task_handler.py:392-404
            envs.append({"artifact_id": aid, "artifact_type": atype, "span_start_ms": a, "span_end_ms": b, "payload": payload})
        session.flush()
        tables = {t: [list(r) for r in session.execute(text(f"SELECT * FROM {t} ORDER BY artifact_id")).fetchall()] for t in ddl}
    return {"ddl": ddl, "envelopes": envs, "tables": tables}


def main():
    if not REF.exists():
        sys.exit("needs /root/reference (build container only)")
    sys.path.insert(0, str(REF))
    from src.services.model_manager import ModelManager

    (HERE / "ref_detect_loop.json").write_text(json.dumps(capture_detect_loop(ModelManager), indent=1))
    (HERE / "ref_scenes.json").write_text(json.dumps(capture_scenes(ModelManager), indent=1))
    (HERE / "ref_places_loop.json").write_text(json.dumps(capture_places_loop(ModelManager)) + "\n")
    (HERE / "ref_artifact_spans.json").write_text(json.dumps(capture_artifact_rules(), indent=1))
    (HERE / "ref_projection_rows.json").write_text(json.dumps(capture_projection_rows(), indent=1) + "\n")
    print("wrote fixtures to", HERE)


def main_projection_only():
    sys.path.insert(0, str(REF))
    out = capture_projection_rows()
    (HERE / "ref_projection_rows.json").write_text(json.dumps(out, indent=1) + "\n")
    print("wrote ref_projection_rows.json:", {k: len(v) for k, v in out["tables"].items()})


def main_places_only():
    sys.path.insert(0, str(REF))
    from src.services.model_manager import ModelManager

    out = capture_places_loop(ModelManager)
    (HERE / "ref_places_loop.json").write_text(json.dumps(out) + "\n")
    print("wrote ref_places_loop.json:", [len(c["result"]["classifications"]) for c in out])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "places":
        main_places_only()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "projection":
        main_projection_only()
        sys.exit(0)
    main()
