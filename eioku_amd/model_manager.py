"""Drop-in ``ModelManager`` for the ml-service worker: same methods, arguments, result dicts and
error behaviour as ``/root/reference/ml-service/src/services/model_manager.py`` for the three task
types on the hot path, with the arithmetic on the MI355X HIP library instead of
ffmpeg / OpenCV / Ultralytics on the CPU.

    detect_objects(video_path, config) -> {"detections": [{frame_index, timestamp_ms, label,
                                           confidence, bbox{x,y,width,height}}]}       (ref :215-306)
    detect_faces(video_path, config)   -> same + "cluster_id": None, label "face"      (ref :308-407)
    detect_scenes(video_path, config)  -> {"scenes": [{scene_index, start_ms, end_ms,
                                           duration_ms}]}                              (ref :715-835)

What changes underneath (and nothing else): sampled frames are detected in batches instead of one
``model(frame)`` call each, and the scene score is computed by K1/K2 instead of an ffmpeg child
process.  Sampling rule, timestamps, label lookup, float widening and the scene-list quirks are the
reference's (pinned by ``tests/golden/ref_detect_loop.json`` / ``ref_scenes.json``).

There is no CPU fallback: without the HIP library / a gfx950 device the calls raise.
"""

from __future__ import annotations

import logging
from pathlib import Path

import numpy as np

logger = logging.getLogger(__name__)

OUT_OF_SCOPE = "is outside the MI355X hot path (SURVEY.md §8): delegate to the reference ModelManager"


class ModelManager:
    """Manages model lifecycle and inference for the hot-path task types."""

    def __init__(self, cache_dir: str = "/models", *, frame_source=None, detector_factory=None, batch_size: int = 64,
                 random_init_seed: int | None = None, place_classifier_factory=None):
        """``cache_dir`` as in the reference (:12-21).  Keyword-only extras are seams for tests and bench:
        ``frame_source(path) -> FrameSource``, ``detector_factory(model_name, cache_dir) -> detector`` with
        ``detect(frames, conf=...) -> (dets, counts)`` and ``names``; ``random_init_seed`` builds random
        weights of the right shapes when no checkpoint can exist (offline benchmarks)."""
        self.cache_dir = Path(cache_dir)
        self.cache_dir.mkdir(parents=True, exist_ok=True)
        self.models = {}
        self._gpu_available = None  # Lazy initialization
        self._frame_source = frame_source
        self._detector_factory = detector_factory
        self._place_classifier_factory = place_classifier_factory  # (cache_dir) -> object with classify(frames, top_k), labels
        self._batch_size = int(batch_size)
        self._lane_streams = {}  # task -> the two HIP streams its detector lanes run on (created once per manager)
        self._seed = random_init_seed

    # ---- GPU probe: identical surface to the reference (:23-42, 168-213) -------------------------
    @property
    def gpu_available(self) -> bool:
        if self._gpu_available is None:
            try:
                import torch

                self._gpu_available = torch.cuda.is_available()
            except Exception as e:  # noqa: BLE001 - same tolerance as the reference
                logger.warning(f"Could not check GPU availability: {e}")
                self._gpu_available = False
        return self._gpu_available

    def _get_device(self) -> str:
        return "cuda" if self.gpu_available else "cpu"  # ROCm torch reports through torch.cuda

    def get_gpu_info(self) -> dict:
        if not self.gpu_available:
            return {"gpu_available": False, "gpu_device_name": None, "gpu_memory_total_mb": None,
                    "gpu_memory_used_mb": None}
        import torch

        return {"gpu_available": True, "gpu_device_name": torch.cuda.get_device_name(0),
                "gpu_memory_total_mb": int(torch.cuda.get_device_properties(0).total_memory / 1e6),
                "gpu_memory_used_mb": int(torch.cuda.memory_allocated(0) / 1e6)}

    def detect_gpu(self) -> bool:
        return self.gpu_available

    def log_gpu_info(self):
        if self.gpu_available:
            info = self.get_gpu_info()
            logger.info(f"GPU device: {info['gpu_device_name']}")
            logger.info(f"GPU memory: {info['gpu_memory_total_mb'] / 1e3:.2f} GB")
        else:
            logger.warning("GPU not available - the HIP hot path cannot run")

    # ---- out-of-scope members: fail loudly, never silently diverge ----------------------------------
    async def download_model(self, model_name: str, model_type: str):
        raise NotImplementedError(f"download_model {OUT_OF_SCOPE} (needs network)")

    async def verify_model(self, model_name: str, model_type: str) -> bool:
        raise NotImplementedError(f"verify_model {OUT_OF_SCOPE}")

    async def transcribe_video(self, video_path: str, config: dict) -> dict:
        raise NotImplementedError(f"transcribe_video {OUT_OF_SCOPE}")

    async def extract_ocr(self, video_path: str, config: dict) -> dict:
        raise NotImplementedError(f"extract_ocr {OUT_OF_SCOPE}")

    async def extract_metadata(self, video_path: str, config: dict) -> dict:
        raise NotImplementedError(f"extract_metadata {OUT_OF_SCOPE}")

    # ---- seams ----------------------------------------------------------------------------------------
    def _open(self, video_path: str):
        if self._frame_source is not None:
            return self._frame_source(video_path)
        from .frames import open_video

        return open_video(video_path)

    def _load_detector(self, model_name: str):
        """``YOLO(cache_dir/ultralytics/model_name); model.to(device)`` (ref :252-254): per job, like the reference."""
        if self._detector_factory is not None:
            return self._detector_factory(model_name, self.cache_dir)
        from .detect import Yolov8Detector

        path = self.cache_dir / "ultralytics" / model_name
        if self._seed is not None and not path.exists():
            return Yolov8Detector.from_model_name(model_name, seed=self._seed)
        return Yolov8Detector.from_model_name(model_name, path=path)

    def _lanes(self, task: str, depth: int = 2):
        """The detector lanes' streams of a task, created on first use and kept: the model is loaded per job like the
        reference's, the streams are not - a HIP stream's hardware queue is fixed when it is created, and a worker that
        made new ones for every job would see its lanes land on whichever queues the earlier jobs left (measured in
        bench.py: the same pipeline at 41 k instead of 58 k frames/s behind three earlier runs)."""
        import torch

        if task not in self._lane_streams:
            self._lane_streams[task] = [torch.cuda.Stream(priority=-1) for _ in range(depth)]
        return self._lane_streams[task]

    # ---- objects / faces: one skeleton, as in the reference --------------------------------------------
    def _detect_loop(self, video_path: str, model_name: str, confidence_threshold: float,
                     frame_interval_seconds: float, face: bool) -> list[dict]:
        cap = self._open(video_path)
        fps = cap.fps or 30
        total_frames = int(cap.total_frames)
        logger.info(f"Video FPS: {fps}, Total frames: {total_frames}")
        frame_interval = max(1, int(fps * frame_interval_seconds))
        frames_to_process = (total_frames + frame_interval - 1) // frame_interval
        logger.info(f"Processing every {frame_interval} frames (every {frame_interval_seconds}s at {fps} FPS, "
                    f"~{frames_to_process} frames to process)")
        detector = self._load_detector(model_name)
        names = detector.names

        detections: list[dict] = []
        pend_frames: list[np.ndarray] = []
        pend_meta: list[tuple[int, int]] = []

        # Two batches in flight on the HIP path (PipelinedDetector: second handle + stream); results are consumed
        # in submission order, so the detection list is the one the synchronous loop produces.
        from .detect import PipelinedDetector, Yolov8Detector

        pipe = (PipelinedDetector(detector, depth=2, streams=self._lanes("face_detection" if face else "object_detection"))
                if isinstance(detector, Yolov8Detector) else None)
        metas: list[list[tuple[int, int]]] = []

        def emit(meta, dets, counts):
            for (frame_idx, timestamp_ms), row, cnt in zip(meta, dets, counts):
                for d in row[: int(cnt)]:
                    x1, y1, x2, y2 = (np.float32(d[k]) for k in ("x1", "y1", "x2", "y2"))
                    confidence = float(np.float32(d["conf"]))  # float32 -> Python float, as float(tensor)
                    if face and confidence < confidence_threshold:
                        continue  # the face path's extra safety filter (ref :375-377)
                    det = {
                        "frame_index": frame_idx,
                        "timestamp_ms": timestamp_ms,
                        "label": "face" if face else names[int(d["cls"])],
                        "confidence": confidence,
                        "bbox": {"x": float(x1), "y": float(y1),
                                 "width": float(np.float32(x2 - x1)), "height": float(np.float32(y2 - y1))},
                    }
                    if face:
                        det["cluster_id"] = None
                    detections.append(det)

        def drain(keep: int):
            while pipe is not None and pipe.in_flight() > keep:
                emit(metas.pop(0), *pipe.result())

        def flush():
            if not pend_frames:
                return
            batch = np.stack(pend_frames)
            if pipe is None:
                emit(list(pend_meta), *detector.detect(batch, conf=confidence_threshold))
            else:
                pipe.submit(batch, conf=confidence_threshold)
                metas.append(list(pend_meta))
                drain(pipe.depth - 1)
            pend_frames.clear()
            pend_meta.clear()

        frame_idx = 0
        try:
            while True:
                if frame_idx % frame_interval == 0:
                    ret, frame = cap.read()
                    if not ret:
                        break
                    pend_frames.append(frame)
                    pend_meta.append((frame_idx, int((frame_idx / fps) * 1000)))
                    if len(pend_frames) >= self._batch_size:
                        flush()
                else:
                    if not cap.grab():
                        break
                frame_idx += 1
            flush()
            drain(0)
        finally:
            cap.release()
            close = pipe.close if pipe is not None else getattr(detector, "close", None)
            if close:
                close()
        return detections

    # ---- places: Places365 ResNet18 (reference :560-713) ---------------------------------------------------
    def _load_place_classifier(self):
        """``models.resnet18`` + 365-way fc + ``<cache>/places365/resnet18_places365.pth.tar`` and the label file
        (ref :579-624), per job like the reference."""
        if self._place_classifier_factory is not None:
            return self._place_classifier_factory(self.cache_dir)
        from .places import Places365Classifier

        return Places365Classifier.from_cache(self.cache_dir, seed=self._seed)

    async def classify_places(self, video_path: str, config: dict) -> dict:
        """Classify places in video frames using Places365 on the HIP path (reference: :560-713): every
        ``max(1, int(fps * frame_interval))``-th frame -> softmax top_k ``{"label", "confidence"}`` lists."""
        try:
            logger.info(f"Place detection: {video_path} (device: {self._get_device()})")
            classifier = self._load_place_classifier()
            classes = classifier.labels
            cap = self._open(video_path)
            fps = cap.fps or 30
            total_frames = int(cap.total_frames)
            frame_interval_seconds = config.get("frame_interval", 1)
            top_k = config.get("top_k", 5)
            frame_interval = max(1, int(fps * frame_interval_seconds))
            frames_to_process = (total_frames + frame_interval - 1) // frame_interval
            logger.info(f"Video FPS: {fps}, Total frames: {total_frames}, Processing every {frame_interval} frames "
                        f"(every {frame_interval_seconds}s, ~{frames_to_process} frames to process)")
            classifications: list[dict] = []
            pend_frames: list[np.ndarray] = []
            pend_meta: list[tuple[int, int]] = []

            def flush():
                if not pend_frames:
                    return
                probs, idx = classifier.classify(np.stack(pend_frames), top_k)
                for (frame_idx, timestamp_ms), p, i in zip(pend_meta, probs, idx):
                    # `for j, i in enumerate(idx[:top_k])`: float(probs[j]) widens the float32 (ref :677-683)
                    classifications.append({"frame_index": frame_idx, "timestamp_ms": timestamp_ms,
                                            "predictions": [{"label": classes[int(c)], "confidence": float(np.float32(v))}
                                                            for v, c in zip(p, i)]})
                pend_frames.clear()
                pend_meta.clear()

            frame_idx = 0
            try:
                while True:
                    if frame_idx % frame_interval == 0:
                        ret, frame = cap.read()
                        if not ret:
                            break
                        pend_frames.append(frame)
                        pend_meta.append((frame_idx, int((frame_idx / fps) * 1000)))
                        if len(pend_frames) >= self._batch_size:
                            flush()
                    else:
                        if not cap.grab():
                            break
                    frame_idx += 1
                flush()
            finally:
                cap.release()
                close = getattr(classifier, "close", None)
                if close:
                    close()
            logger.info(f"✅ Place detection complete: {len(classifications)} classifications")
            return {"classifications": classifications}
        except Exception as e:
            logger.error(f"Place detection failed: {e}", exc_info=True)
            raise

    async def detect_objects(self, video_path: str, config: dict) -> dict:
        """Detect objects in video using YOLOv8 on the HIP path (reference: :215-306)."""
        try:
            model_name = config.get("model_name", "yolov8n.pt")
            confidence_threshold = config.get("confidence_threshold", 0.5)
            frame_interval_seconds = config.get("frame_interval", 1)
            logger.info(f"Object detection: {video_path} (device: {self._get_device()})")
            detections = self._detect_loop(video_path, model_name, confidence_threshold, frame_interval_seconds, False)
            logger.info(f"✅ Object detection complete: {len(detections)} detections")
            return {"detections": detections}
        except Exception as e:
            logger.error(f"Object detection failed: {e}", exc_info=True)
            raise

    async def detect_faces(self, video_path: str, config: dict) -> dict:
        """Detect faces in video using YOLOv8-face on the HIP path (reference: :308-407)."""
        try:
            model_name = config.get("model_name", "yolov8n-face.pt")
            confidence_threshold = config.get("confidence_threshold", 0.7)
            frame_interval_seconds = config.get("frame_interval", 3)
            logger.info(f"Face detection: {video_path} (device: {self._get_device()})")
            detections = self._detect_loop(video_path, model_name, confidence_threshold, frame_interval_seconds, True)
            logger.info(f"✅ Face detection complete: {len(detections)} detections")
            return {"detections": detections}
        except Exception as e:
            logger.error(f"Face detection failed: {e}", exc_info=True)
            raise

    # ---- single pass: one decode, one upload, three stages ------------------------------
    async def analyze_video(self, video_path: str, configs: dict) -> dict:
        """Scenes + objects + faces from ONE read of the file.

        The reference opens and decodes the file once per task (``cv2.VideoCapture`` at :237 and :331, an ffmpeg child at
        :750-755): three decodes of every frame, and the detection loops decode even the frames they skip (``grab()``, :294).
        Here every frame is read once, goes through pinned host memory into HBM once (chunks of ``batch_size`` frames), and
        all three stages read that copy: K1 (``eioku_scene_sad_luma_bgr``: the luma OpenCV derives from these BGR frames)
        on every frame, the two detectors on the frames their own sampling rule picks (``max(1, int(fps * seconds))``,
        :243 / :337).  ``configs``: ``{"scene_detection": {...}, "object_detection": {...}, "face_detection": {...}}`` -
        the per-task config dicts the backend would have put into three jobs; a task type that is absent is not run.

        Returns ``{task_type: result}`` with, per task, exactly the dict the separate call returns on a BGR source
        (``.npy`` clips and cv2 captures whose backend hands back BGR: tests/test_single_pass_gpu.py).  On a capture
        that can expose the decoder's own Y plane the separate ``detect_scenes`` scores THAT plane (bit-exact with
        ffmpeg); this entry scores the BGR-derived luma, which differs by the YUV -> BGR -> Y round trip.
        """
        import torch

        from . import scene
        from .detect import PipelinedDetector

        unknown = set(configs) - {"scene_detection", "object_detection", "face_detection", "decode"}
        if unknown:
            raise NotImplementedError(f"analyze_video covers the hot-path task types only, got {sorted(unknown)}")
        logger.info(f"Single-pass analysis ({', '.join(sorted(configs))}): {video_path} (device: {self._get_device()})")
        src = self._open(video_path)
        fps = src.fps or 30
        total_frames = int(src.total_frames)
        dev = torch.device("cuda", torch.cuda.current_device())
        chunk_n = max(1, self._batch_size)
        # Decoder planes when the source can hand them over (a .y4m clip, a cv2 capture that honours CONVERT_RGB = 0):
        # 1.5 bytes per pixel cross PCIe instead of 3, the scene score is taken on the decoder's own Y plane (what ffmpeg's
        # select filter scores: bit-exact with the separate detect_scenes), and the BGR frames the detectors / the
        # ContentDetector read are produced on the device with OpenCV's integer BT.601 (eioku_yuv420_to_bgr) - the
        # conversion cap.read() would have run on the CPU
        if getattr(src, "yuv_layout", None) is None and hasattr(src, "try_yuv") and configs.get("decode", {}).get("planes", True):
            src.try_yuv()
        yuv = getattr(src, "yuv_layout", None)
        lanes = {}  # task -> detection lane state
        for task, face, dflt_model, dflt_conf, dflt_sec in (("object_detection", False, "yolov8n.pt", 0.5, 1),
                                                            ("face_detection", True, "yolov8n-face.pt", 0.7, 3)):
            if task not in configs:
                continue
            cfg = configs[task] or {}
            detector = self._load_detector(cfg.get("model_name", dflt_model))
            lanes[task] = {"face": face, "conf": cfg.get("confidence_threshold", dflt_conf),
                           "interval": max(1, int(fps * cfg.get("frame_interval", dflt_sec))), "names": detector.names,
                           "pipe": PipelinedDetector(detector, depth=2, streams=self._lanes(task)), "frames": [], "meta": [],
                           "metas": [], "out": []}
        want_scenes = "scene_detection" in configs
        scfg = configs.get("scene_detection") or {}
        content = scfg.get("detector", "ffmpeg") == "content"
        sums, prev_dev, npx = [], None, 1
        pinned = [None, None]
        copied = [None, None]  # events: the chunk's upload has left the pinned buffer

        def emit(lane, meta, dets, counts):
            for (frame_idx, timestamp_ms), row, cnt in zip(meta, dets, counts):
                for d in row[: int(cnt)]:
                    x1, y1, x2, y2 = (np.float32(d[k]) for k in ("x1", "y1", "x2", "y2"))
                    confidence = float(np.float32(d["conf"]))
                    if lane["face"] and confidence < lane["conf"]:
                        continue
                    det = {"frame_index": frame_idx, "timestamp_ms": timestamp_ms,
                           "label": "face" if lane["face"] else lane["names"][int(d["cls"])], "confidence": confidence,
                           "bbox": {"x": float(x1), "y": float(y1), "width": float(np.float32(x2 - x1)),
                                    "height": float(np.float32(y2 - y1))}}
                    if lane["face"]:
                        det["cluster_id"] = None
                    lane["out"].append(det)

        def drain(lane, keep):
            while lane["pipe"].in_flight() > keep:
                emit(lane, lane["metas"].pop(0), *lane["pipe"].result())

        def flush(lane):
            if not lane["frames"]:
                return
            batch = torch.stack(lane["frames"]) if len(lane["frames"]) > 1 else lane["frames"][0][None]
            lane["pipe"].submit(batch.contiguous(), conf=lane["conf"])
            lane["metas"].append(list(lane["meta"]))
            lane["frames"].clear()
            lane["meta"].clear()
            drain(lane, lane["pipe"].depth - 1)

        frame_idx, slot = 0, 0
        try:
            done = False
            while not done:
                host = []
                while len(host) < chunk_n:
                    ret, frame = src.read_yuv() if yuv else src.read()
                    if not ret:
                        done = True
                        break
                    host.append(frame)
                if not host:
                    break
                n = len(host)
                h, w = host[0].shape[:2]
                fshape = (h, w) if yuv else (h, w, 3)
                if yuv:
                    h = h * 2 // 3
                npx = h * w
                if pinned[slot] is None or tuple(pinned[slot].shape[1:]) != fshape:
                    pinned = [torch.empty((chunk_n, *fshape), dtype=torch.uint8).pin_memory() for _ in range(2)]
                    copied = [None, None]
                if copied[slot] is not None:
                    copied[slot].synchronize()  # the upload that last used this pinned buffer has finished
                buf = pinned[slot][:n]
                view = buf.numpy()  # the pinned buffer as an ndarray: frames are copied in without wrapping (possibly
                for i, f in enumerate(host):  # read-only, memory-mapped) source arrays as tensors
                    view[i] = f
                chunk = buf.to(dev, non_blocking=True)  # the one trip over PCIe
                ev = torch.cuda.Event()
                ev.record()
                copied[slot] = ev
                slot ^= 1
                planes = None
                if yuv:
                    planes = chunk
                    chunk = scene.yuv420_to_bgr(planes, h, w, yuv) if (lanes or (want_scenes and content)) else None
                if want_scenes and planes is not None and not content:
                    # K1 straight on the Y planes of the uploaded frames: rows 0 .. h-1 of every (3h/2, w) frame
                    sums.append(scene.luma_sad(planes, prev_dev, shape=(n, h, w), row_stride=w, frame_stride=h * 3 // 2 * w,
                                               keep_on_device=True))
                    prev_dev = planes[n - 1]
                elif want_scenes:
                    if content:
                        sums.append(scene.hsv_sums(chunk, prev_dev, keep_on_device=True))
                    elif npx % 4 == 0:
                        sums.append(scene.luma_sad_bgr(chunk, prev_dev, keep_on_device=True))
                    else:  # odd pixel counts: luma plane on the device, then K1
                        c = chunk.to(torch.int64)
                        y = ((269484 * c[..., 2] + 528482 * c[..., 1] + 102760 * c[..., 0] + (16 << 20) + (1 << 19)) >> 20).to(torch.uint8)
                        py = None
                        if prev_dev is not None:
                            p = prev_dev.to(torch.int64)
                            py = ((269484 * p[..., 2] + 528482 * p[..., 1] + 102760 * p[..., 0] + (16 << 20) + (1 << 19)) >> 20).to(torch.uint8)
                        sums.append(scene.luma_sad(y.contiguous(), py.contiguous() if py is not None else None, keep_on_device=True))
                    prev_dev = chunk[n - 1]
                for lane in lanes.values():
                    for i in range(n):
                        idx = frame_idx + i
                        if idx % lane["interval"] == 0:
                            lane["frames"].append(chunk[i])
                            lane["meta"].append((idx, int((idx / fps) * 1000)))
                            if len(lane["frames"]) >= self._batch_size:
                                flush(lane)
                frame_idx += n
            for lane in lanes.values():
                flush(lane)
                drain(lane, 0)
        finally:
            src.release()
            for lane in lanes.values():
                lane["pipe"].close()
        out = {}
        for task, lane in lanes.items():
            out[task] = {"detections": lane["out"]}
        if want_scenes:
            tb_num, tb_den = src.time_base
            duration_ms = None if src.duration_s is None else int(float(src.duration_s) * 1000)
            s_all = torch.cat(sums).cpu().numpy() if sums else np.zeros((0, 3) if content else (0,), np.uint64)
            if content:
                scores = scene.content_scores(s_all, npx)
                cuts = scene.content_cuts(scores, float(scfg.get("content_threshold", 27.0)), int(scfg.get("min_scene_len", 15)),
                                          scfg.get("filter_mode", "legacy"))
                ts = [0] + [int(float(scene.pts_time_string(c, tb_num, tb_den)) * 1000) for c in cuts]
                end = duration_ms if duration_ms is not None else (ts[-1] + 1000)
                out["scene_detection"] = {"scenes": [{"scene_index": i, "start_ms": a, "end_ms": b, "duration_ms": b - a}
                                                     for i, (a, b) in enumerate(zip(ts, ts[1:] + [end]))]}
            else:
                _, score = scene.ffmpeg_scene_scores(s_all, npx)
                cut_ms = [int(float(scene.pts_time_string(int(c), tb_num, tb_den)) * 1000)
                          for c in np.nonzero(score > float(scfg.get("threshold", 0.7)))[0]]
                out["scene_detection"] = {"scenes": scene.build_scenes(cut_ms, duration_ms)}
        logger.info(f"✅ Single-pass analysis complete: {frame_idx} frames (header: {total_frames}), "
                    + ", ".join(f"{k}: {len(v.get('detections', v.get('scenes', [])))}" for k, v in out.items()))
        return out

    # ---- scenes ----------------------------------------------------------------------------------------
    async def detect_scenes(self, video_path: str, config: dict) -> dict:
        """Scene boundaries.  Default = what the reference observes from
        ``ffmpeg -vf select='gt(scene\\,T)',showinfo`` (:736-828): luma SAD score (K1), ``pts_time`` with
        six significant digits, the scene list with its index quirk.  ``config["detector"] = "content"``
        selects the PySceneDetect ContentDetector of BASELINE.json's north_star (K2) instead."""
        try:
            from . import scene

            logger.info(f"Scene detection: {video_path}")
            threshold = config.get("threshold", 0.7)
            src = self._open(video_path)
            try:
                n = int(src.total_frames)
                tb_num, tb_den = src.time_base
                duration_ms = None if src.duration_s is None else int(float(src.duration_s) * 1000)
                chunk = 64
                if config.get("detector", "ffmpeg") == "content":
                    sums = []
                    prev = None
                    for lo in range(0, n, chunk):
                        frames = self._bgr_chunk(src, lo, min(chunk, n - lo))
                        sums.append(scene.hsv_sums(frames, prev))
                        prev = frames[-1]
                    sums = np.concatenate(sums) if sums else np.zeros((0, 3), np.uint64)
                    h, w = (frames.shape[1], frames.shape[2]) if n else (1, 1)
                    scores = scene.content_scores(sums, h * w)
                    cuts = scene.content_cuts(scores, float(config.get("content_threshold", 27.0)),
                                              int(config.get("min_scene_len", 15)), config.get("filter_mode", "legacy"))
                    ts = [0] + [int(float(scene.pts_time_string(c, tb_num, tb_den)) * 1000) for c in cuts]
                    end = duration_ms if duration_ms is not None else (ts[-1] + 1000)
                    scenes = [{"scene_index": i, "start_ms": a, "end_ms": b, "duration_ms": b - a}
                              for i, (a, b) in enumerate(zip(ts, ts[1:] + [end]))]
                    return {"scenes": scenes}
                from .frames import EndOfStream

                sad = []
                prev = None
                y = None
                lo = 0
                while lo < n:
                    want = min(chunk, n - lo)
                    try:
                        y = np.ascontiguousarray(src.luma_planes(lo, want))
                    except EndOfStream:
                        break  # the header's frame count was an estimate: the stream ended early
                    sad.append(scene.luma_sad(y, prev))
                    prev = y[-1]
                    lo += len(y)
                    if len(y) < want:
                        break
                sad = np.concatenate(sad) if sad else np.zeros(0, np.uint64)
                count = int(y.shape[1] * y.shape[2]) if y is not None else 1
                _, score = scene.ffmpeg_scene_scores(sad, count)
                cut_ms = [int(float(scene.pts_time_string(int(c), tb_num, tb_den)) * 1000)
                          for c in np.nonzero(score > float(threshold))[0]]
                scenes = scene.build_scenes(cut_ms, duration_ms)
                if len(cut_ms) == 0:
                    logger.info(f"No scene cuts detected. Created single scene for entire video ({scenes[0]['end_ms']}ms)")
                logger.info(f"✅ Scene detection complete: {len(scenes)} scenes")
                return {"scenes": scenes}
            finally:
                src.release()
        except Exception as e:
            logger.error(f"Scene detection failed: {e}", exc_info=True)
            raise

    @staticmethod
    def _bgr_chunk(src, lo, count):
        frames = getattr(src, "frames", None)
        if frames is not None:
            return np.ascontiguousarray(frames[lo:lo + count])
        out = []
        for _ in range(count):
            ok, f = src.read()
            if not ok:
                break
            out.append(f)
        return np.stack(out)
