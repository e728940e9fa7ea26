"""``process_ml_task`` for the hot-path task types, with the reference's signature, result and
error behaviour (``/root/reference/ml-service/src/workers/task_handler.py:22-488``).

The reference's handler interleaves three things: (1) SQLAlchemy task-row bookkeeping, (2) the
inference call, (3) result -> ``ArtifactEnvelope`` mapping + bulk insert + projection sync.  (1) and
the SQL half of (3) are persistence plumbing outside this path (out of scope - a
deployment keeps the reference's own handler and swaps only its ``ModelManager`` import, see
INTEGRATION.md).  This module is the database-free mirror of (2) and the mapping half of (3), so the
boundary can be exercised and benchmarked without PostgreSQL: task-row transitions and artifact
storage go through two small hooks carried in the arq ``ctx`` dict.

    ctx["task_store"]   optional object with mark_running / mark_completed / mark_failed /
                        mark_cancelled (task_id[, error])            (ref :68-78, 422-425, 437-469)
    ctx["artifact_sink"] optional callable(list[ArtifactEnvelope])   (ref :344-404)
"""

from __future__ import annotations

import asyncio
import json
import logging
import os
from dataclasses import dataclass
from datetime import datetime
from uuid import uuid4

logger = logging.getLogger(__name__)

TASK_TO_ARTIFACT_TYPE = {"object_detection": "object.detection", "face_detection": "face.detection",
                         "scene_detection": "scene", "segment_embedding": "segment.embedding",
                         "place_detection": "place.classification"}
TASK_TO_RESULT_KEY = {"object_detection": "detections", "face_detection": "detections", "scene_detection": "scenes",
                      "segment_embedding": "embeddings", "place_detection": "classifications"}
# the reference's seven (task_handler.py:92-127) + the one its semantic-search design adds after transcription
# (.kiro/specs/semantic-video-search/tasks.md:297-302): embed the transcript segments, index them
KNOWN_TASK_TYPES = ("object_detection", "face_detection", "transcription", "ocr", "place_detection",
                    "scene_detection", "metadata_extraction", "segment_embedding")


@dataclass
class ArtifactEnvelope:
    """Field set and validation of the reference's envelope (``ml-service/src/domain/artifacts.py:7-73``)."""

    artifact_id: str
    asset_id: str
    artifact_type: str
    schema_version: int
    span_start_ms: int
    span_end_ms: int
    payload_json: str
    producer: str
    producer_version: str
    model_profile: str
    config_hash: str
    input_hash: str
    run_id: str
    created_at: datetime

    def __post_init__(self):
        for field in ("artifact_id", "asset_id", "artifact_type", "schema_version", "span_start_ms", "span_end_ms",
                      "payload_json", "producer", "producer_version", "model_profile", "config_hash", "input_hash",
                      "run_id", "created_at"):
            if getattr(self, field) is None:
                raise ValueError(f"Required field '{field}' cannot be None")
        if self.span_start_ms < 0:
            raise ValueError("span_start_ms must be non-negative")
        if self.span_end_ms < 0:
            raise ValueError("span_end_ms must be non-negative")
        if self.span_start_ms > self.span_end_ms:
            raise ValueError("span_start_ms must be <= span_end_ms")
        if self.schema_version < 1:
            raise ValueError("schema_version must be >= 1")


def result_to_envelopes(result_dict: dict, task_id: str, task_type: str, video_id: str, run_id: str | None = None):
    """The reference's result -> envelope mapping (:142-337) for detection / scene rows."""
    run_id = run_id or str(uuid4())
    config_hash = result_dict.get("config_hash", "")
    input_hash = result_dict.get("input_hash", "")
    producer = result_dict.get("producer", "ml-service")
    producer_version = result_dict.get("producer_version", "1.0.0")
    model_profile = result_dict.get("model_profile", "balanced")
    artifact_type = TASK_TO_ARTIFACT_TYPE[task_type]
    detections = result_dict.get(TASK_TO_RESULT_KEY[task_type], [])
    envelopes = []
    for idx, detection in enumerate(detections):
        try:
            if "start_ms" in detection and "end_ms" in detection:
                span_start_ms = int(detection.get("start_ms", 0))
                span_end_ms = int(detection.get("end_ms", 0))
            elif "timestamp_ms" in detection:
                span_start_ms = span_end_ms = int(detection.get("timestamp_ms", 0))
            else:
                logger.warning(f"⚠️  No time information in detection {idx} for task {task_id}")
                continue
            if span_start_ms < 0 or span_end_ms < 0 or span_start_ms > span_end_ms:
                logger.warning(f"⚠️  Invalid time span for detection {idx}: start={span_start_ms}, end={span_end_ms}")
                continue
            envelopes.append(ArtifactEnvelope(
                artifact_id=f"{video_id}_{task_type}_{run_id}_{idx}", asset_id=video_id, artifact_type=artifact_type,
                schema_version=1, span_start_ms=span_start_ms, span_end_ms=span_end_ms,
                payload_json=json.dumps(detection), producer=producer, producer_version=producer_version,
                model_profile=model_profile, config_hash=config_hash, input_hash=input_hash, run_id=run_id,
                created_at=datetime.utcnow()))
        except (ValueError, KeyError) as e:
            logger.error(f"❌ Error transforming detection {idx} for task {task_id}: {e}")
            continue
    return envelopes


def embed_segments(engine, video_id: str, segments: list[dict]) -> dict:
    """The ``segment_embedding`` task body: K8 over the transcript segments, vectors into the store, one result row per
    segment (span = the segment's, payload = text + 384 floats) for the artifact table."""
    texts = [s["text"] for s in segments]
    emb = engine.generator.generate_batch_embeddings(texts)
    engine.store.delete_by_video_id(video_id)  # a re-run replaces the video's vectors, it does not duplicate them
    meta, rows = [], []
    for i, (s, e) in enumerate(zip(segments, emb)):
        start_ms = int(s["start_ms"]) if "start_ms" in s else int(float(s.get("start", 0.0)) * 1000)
        end_ms = int(s["end_ms"]) if "end_ms" in s else int(float(s.get("end", start_ms / 1000.0)) * 1000)
        meta.append({"video_id": video_id, "start_time": start_ms / 1000.0, "end_time": end_ms / 1000.0, "text": s["text"],
                     "thumbnail_path": s.get("thumbnail_path")})
        rows.append({"start_ms": start_ms, "end_ms": end_ms, "text": s["text"], "embedding": [float(v) for v in e]})
    engine.store.index_segments([f"{video_id}_seg{i}" for i in range(len(segments))], emb, meta)
    return {"embeddings": rows}


async def process_ml_task(ctx, task_id: str, task_type: str, video_id: str, video_path: str,
                          config: dict | None = None) -> dict:
    """Run one hot-path ML task and hand its artifacts to the sink.

    Returns ``{"task_id", "status": "completed", "artifact_count"}``; raises ``RuntimeError(f"Failed to
    process task {task_id}: {e}")`` on any failure and re-raises ``asyncio.CancelledError``, exactly
    like the reference (:431-469).
    """
    ctx = ctx or {}
    store = ctx.get("task_store")
    try:
        logger.info(f"🚀 Dequeued task {task_id} ({task_type}) for video {video_id}")
        if store:
            store.mark_running(task_id)
        from .model_manager import ModelManager

        model_cache_dir = os.getenv("MODEL_CACHE_DIR", "/models")
        factory = ctx.get("model_manager_factory", ModelManager)
        model_manager = factory(cache_dir=model_cache_dir)
        if task_type not in KNOWN_TASK_TYPES:
            raise ValueError(f"Unknown task type: {task_type}")
        logger.info(f"🎬 Starting {task_type} inference on {video_path}")
        if task_type == "object_detection":
            result = await model_manager.detect_objects(video_path, config or {})
        elif task_type == "face_detection":
            result = await model_manager.detect_faces(video_path, config or {})
        elif task_type == "scene_detection":
            result = await model_manager.detect_scenes(video_path, config or {})
        elif task_type == "place_detection":
            result = await model_manager.classify_places(video_path, config or {})
        elif task_type == "segment_embedding":
            # segments: the transcription task's output for this video.  The reference would read them back from its
            # artifact table; without a database they arrive in the job config or through ctx["segment_source"](video_id)
            segments = (config or {}).get("segments")
            if segments is None and ctx.get("segment_source"):
                segments = ctx["segment_source"](video_id)
            if segments is None:
                raise ValueError("segment_embedding needs config['segments'] or ctx['segment_source']")
            engine = ctx.get("search_engine")
            if engine is None:
                raise ValueError("segment_embedding needs ctx['search_engine'] (eioku_amd.semantic.SemanticSearchEngine)")
            result = embed_segments(engine, video_id, segments)
        else:
            raise NotImplementedError(f"task type {task_type} is outside the MI355X hot path; route it to the "
                                      "reference worker")
        envelopes = result_to_envelopes(result, task_id, task_type, video_id)
        sink = ctx.get("artifact_sink")
        if sink and envelopes:
            sink(envelopes)
        if store:
            store.mark_completed(task_id)
        logger.info(f"✅ Task {task_id} ({task_type}) marked as COMPLETED ({len(envelopes)} artifacts persisted)")
        return {"task_id": task_id, "status": "completed", "artifact_count": len(envelopes)}
    except asyncio.CancelledError:
        logger.warning(f"⚠️  Task {task_id} was cancelled via arq")
        if store:
            try:
                store.mark_cancelled(task_id)
            except Exception as e:  # noqa: BLE001
                logger.error(f"❌ Failed to mark task as cancelled: {e}")
        raise
    except Exception as e:
        logger.error(f"❌ Error processing task {task_id}: {e}", exc_info=True)
        if store:
            try:
                store.mark_failed(task_id, str(e))
            except Exception as db_error:  # noqa: BLE001
                logger.error(f"❌ Failed to mark task as failed: {db_error}")
        raise RuntimeError(f"Failed to process task {task_id}: {e}")
