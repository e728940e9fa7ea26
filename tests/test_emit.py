"""CPU: batched artifact emission reproduces the rows the reference's per-artifact
ProjectionSyncService leaves in the projection tables (tests/golden/ref_projection_rows.json: captured by running the
reference's own service on SQLite, tests/golden/make_reference_fixtures.py) - in 1 + (tables touched) statements per
task instead of 1 + (artifacts)."""
import asyncio
import json
import sqlite3
from datetime import datetime

import pytest

from conftest import GOLDEN
from eioku_amd import emit, task_handler

REF = json.loads((GOLDEN / "ref_projection_rows.json").read_text())
ARTIFACTS_DDL = ("artifact_id TEXT, asset_id TEXT, artifact_type TEXT, schema_version INTEGER, span_start_ms INTEGER, span_end_ms INTEGER, "
                 "payload_json TEXT, producer TEXT, producer_version TEXT, model_profile TEXT, config_hash TEXT, input_hash TEXT, run_id TEXT, "
                 "created_at TEXT")


def _db():
    conn = sqlite3.connect(":memory:")
    for table, cols in REF["ddl"].items():
        conn.execute(f"CREATE TABLE {table} ({cols})")
    conn.execute(f"CREATE TABLE artifacts ({ARTIFACTS_DDL})")
    return conn


def _envelopes():
    return [task_handler.ArtifactEnvelope(
        artifact_id=e["artifact_id"], asset_id="vid", artifact_type=e["artifact_type"], schema_version=1,
        span_start_ms=e["span_start_ms"], span_end_ms=e["span_end_ms"], payload_json=json.dumps(e["payload"]), producer="ml-service",
        producer_version="1.0.0", model_profile="balanced", config_hash="", input_hash="", run_id="r", created_at=datetime(2026, 1, 28))
        for e in REF["envelopes"]]


def test_batched_write_leaves_the_reference_rows_in_every_projection_table():
    conn = _db()
    w = emit.ArtifactBatchWriter(conn)
    counts = w.write(_envelopes())
    assert w.statements == 4 and counts["artifacts"] == len(REF["envelopes"])  # artifacts + three projection tables
    for table, want in REF["tables"].items():
        got = [list(r) for r in conn.execute(f"SELECT * FROM {table} ORDER BY artifact_id")]
        assert got == want, table
    # the upsert case (an artifact id seen twice): last write wins, as INSERT OR REPLACE / ON CONFLICT DO UPDATE do
    assert conn.execute("SELECT label FROM object_labels WHERE artifact_id = 'vid_object_detection_r_0'").fetchone() == ("cat",)
    assert conn.execute("SELECT COUNT(*) FROM artifacts").fetchone() == (len(REF["envelopes"]),)
    row = conn.execute("SELECT payload_json, created_at FROM artifacts WHERE artifact_id = 'vid_scene_r_1'").fetchone()
    assert json.loads(row[0])["scene_index"] == 2 and row[1].startswith("2026-01-28")
    assert emit.ArtifactBatchWriter(conn).write([]) == {}


def test_postgresql_statements_are_on_conflict_upserts():
    class Cur:
        def __init__(self, log):
            self.log = log

        def executemany(self, sql, rows):
            self.log.append((sql, list(rows)))

    class Conn:
        def __init__(self):
            self.log = []

        def cursor(self):
            return Cur(self.log)

    conn = Conn()
    emit.ArtifactBatchWriter(conn, "postgresql", payload_as_text=False).write(_envelopes()[:4])
    assert len(conn.log) == 3  # artifacts, scene_ranges, object_labels
    sql = conn.log[1][0]
    assert sql.startswith("INSERT INTO scene_ranges (artifact_id, asset_id, scene_index, start_ms, end_ms) VALUES (%s, %s, %s, %s, %s)")
    assert "ON CONFLICT (artifact_id) DO UPDATE SET asset_id = EXCLUDED.asset_id, scene_index = EXCLUDED.scene_index" in sql
    assert isinstance(conn.log[0][1][0][6], dict)  # JSONB: the payload goes to the driver as a dict
    with pytest.raises(ValueError):
        emit.ArtifactBatchWriter(conn, "mysql")


def test_writer_is_the_artifact_sink_of_process_ml_task(tmp_path):
    """ctx['artifact_sink'] = writer.write: a whole task's artifacts land in 2 statements."""
    class FakeManager:
        def __init__(self, cache_dir):
            pass

        async def detect_scenes(self, path, config):
            return {"scenes": [{"scene_index": i, "start_ms": 1000 * i, "end_ms": 1000 * i + 900, "duration_ms": 900} for i in range(500)]}

    conn = _db()
    w = emit.ArtifactBatchWriter(conn)
    out = asyncio.run(task_handler.process_ml_task({"artifact_sink": w.write, "model_manager_factory": FakeManager}, "t", "scene_detection",
                                                   "vid", "/videos/x.mp4", {}))
    assert out["artifact_count"] == 500 and w.statements == 2
    assert conn.execute("SELECT COUNT(*), MIN(start_ms), MAX(end_ms) FROM scene_ranges").fetchone() == (500, 0, 499900)
