"""Batched artifact emission: the second hot loop of a job once inference is fast.

The reference stores a task's artifacts with one ``session.add_all`` and then runs ``ProjectionSyncService.sync_artifact``
once PER ARTIFACT - one SQL upsert round trip each (``/root/reference/ml-service/src/workers/task_handler.py:344-404``,
``services/projection_sync_service.py:26-330``).  A 1 h video at the live sampling rate yields ~1 200 sampled frames
and several thousand detections per task: thousands of statements for three projection tables.

``ArtifactBatchWriter.write`` does the same work in a fixed number of statements per task: one ``executemany`` into
``artifacts`` and one ``executemany`` upsert per projection table that the batch touches (``scene_ranges``,
``object_labels``, ``face_clusters``) - same tables, same columns, same values (the row builders below restate the
reference's field extraction: ``payload.get("scene_index", 0)``, ``label`` default ``""``, ``confidence`` default 0.0,
``cluster_id`` default NULL), same upsert semantics (``ON CONFLICT (artifact_id) DO UPDATE`` on PostgreSQL,
``INSERT OR REPLACE`` on SQLite).  It talks to any DB-API 2.0 connection (``sqlite3``, ``psycopg``); the caller owns
the transaction, as the reference's handler does (flush, then commit after the task row is marked completed).

Plugged into ``process_ml_task`` as ``ctx["artifact_sink"] = writer.write``.
"""
from __future__ import annotations

import json

ARTIFACT_COLUMNS = ("artifact_id", "asset_id", "artifact_type", "schema_version", "span_start_ms", "span_end_ms",
                    "payload_json", "producer", "producer_version", "model_profile", "config_hash", "input_hash", "run_id",
                    "created_at")

# projection table -> (columns after artifact_id / asset_id, row builder from (payload dict, envelope))
PROJECTIONS = {
    "scene": ("scene_ranges", ("scene_index", "start_ms", "end_ms"),
              lambda p, e: (p.get("scene_index", 0), e.span_start_ms, e.span_end_ms)),
    "object.detection": ("object_labels", ("label", "confidence", "start_ms", "end_ms"),
                         lambda p, e: (p.get("label", ""), p.get("confidence", 0.0), e.span_start_ms, e.span_end_ms)),
    "face.detection": ("face_clusters", ("cluster_id", "confidence", "start_ms", "end_ms"),
                       lambda p, e: (p.get("cluster_id"), p.get("confidence", 0.0), e.span_start_ms, e.span_end_ms)),
}


class ArtifactBatchWriter:
    def __init__(self, connection, dialect: str = "sqlite", payload_as_text: bool = True):
        """``dialect``: "sqlite" (``?`` placeholders, INSERT OR REPLACE) or "postgresql" (``%s``, ON CONFLICT DO UPDATE;
        ``payload_as_text=False`` hands the payload dict to the driver for a JSONB column, as the reference's ORM does)."""
        if dialect not in ("sqlite", "postgresql"):
            raise ValueError(f"unsupported dialect {dialect!r}")
        self.conn, self.dialect, self.payload_as_text = connection, dialect, payload_as_text
        self.statements = 0  # executemany calls issued (what the batching is about)

    def _ph(self, n: int) -> str:
        return ", ".join(["?" if self.dialect == "sqlite" else "%s"] * n)

    def _upsert_sql(self, table: str, cols: tuple[str, ...]) -> str:
        allc = ("artifact_id", "asset_id") + cols
        if self.dialect == "sqlite":
            return f"INSERT OR REPLACE INTO {table} ({', '.join(allc)}) VALUES ({self._ph(len(allc))})"
        sets = ", ".join(f"{c} = EXCLUDED.{c}" for c in allc[1:])
        return (f"INSERT INTO {table} ({', '.join(allc)}) VALUES ({self._ph(len(allc))}) "
                f"ON CONFLICT (artifact_id) DO UPDATE SET {sets}")

    def write(self, envelopes) -> dict:
        """Insert the artifacts and upsert their projection rows; returns ``{table: rows}`` (and counts statements)."""
        envelopes = list(envelopes)
        if not envelopes:
            return {}
        cur = self.conn.cursor()
        rows = []
        by_table: dict[str, list[tuple]] = {}
        for e in envelopes:
            payload = json.loads(e.payload_json)
            created = e.created_at.isoformat(sep=" ") if self.dialect == "sqlite" and hasattr(e.created_at, "isoformat") else e.created_at
            rows.append((e.artifact_id, e.asset_id, e.artifact_type, e.schema_version, e.span_start_ms, e.span_end_ms,
                         e.payload_json if self.payload_as_text else payload, e.producer, e.producer_version, e.model_profile,
                         e.config_hash, e.input_hash, e.run_id, created))
            proj = PROJECTIONS.get(e.artifact_type)
            if proj is not None:
                by_table.setdefault(e.artifact_type, []).append((e.artifact_id, e.asset_id) + tuple(proj[2](payload, e)))
        cur.executemany(f"INSERT INTO artifacts ({', '.join(ARTIFACT_COLUMNS)}) VALUES ({self._ph(len(ARTIFACT_COLUMNS))})", rows)
        self.statements += 1
        out = {"artifacts": len(rows)}
        for atype, prows in by_table.items():
            table, cols, _ = PROJECTIONS[atype]
            cur.executemany(self._upsert_sql(table, cols), prows)
            self.statements += 1
            out[table] = len(prows)
        return out
