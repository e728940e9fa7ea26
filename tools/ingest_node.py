#!/usr/bin/env python3
"""Ingest a list of videos on this node, one worker process per GPU (BASELINE cfg4: 8 x 1 h 1080p, shard by video).

    python tools/ingest_node.py --gpus 8 --tasks scene_detection,object_detection,face_detection videos.txt

videos.txt: one path per line, optionally ``path<TAB>weight`` (frames / bytes / seconds: longest jobs are placed first).
Each (video, task) pair is one job = one ``process_ml_task`` call, as the backend enqueues them
(``/root/reference/backend/src/services/job_producer.py:102-111``).  Prints one JSON line per job, input order.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("videos")
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--tasks", default="scene_detection,object_detection,face_detection")
    ap.add_argument("--config", default="{}", help='JSON: {"object_detection": {"model_name": "yolov8m.pt", ...}, ...}')
    ap.add_argument("--ctx-factory", default=None, help="module:function building the ctx dict (task store / artifact sink)")
    args = ap.parse_args()
    from eioku_amd import ingest  # imports no GPU stack in the parent

    cfg = json.loads(args.config)
    jobs = []
    for n, line in enumerate(open(args.videos)):
        line = line.rstrip("\n")
        if not line:
            continue
        path, _, weight = line.partition("\t")
        for task in args.tasks.split(","):
            jobs.append(ingest.Job(f"{n}_{task}", task, f"video{n}", path, cfg.get(task, {}), float(weight or 1.0)))
    for r in ingest.run_node(jobs, list(range(args.gpus)), args.ctx_factory):
        print(json.dumps(r))


if __name__ == "__main__":
    main()
