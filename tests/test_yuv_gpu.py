"""GPU: decoder planes in, BGR on the device (csrc/yuv.hip) - bit-exact against oracle/yuv.py (OpenCV's integer BT.601)
on a full colour lattice and on ragged frames, both layouts, host and device buffers; and the single-pass ingest on a
4:2:0 clip against ORACLE-derived expectations (not only against the three separate calls)."""
import asyncio
import json

import numpy as np
import pytest

from oracle import prng, scene as oscene, yuv as oyuv
from eioku_amd import scene
from eioku_amd.model_manager import ModelManager

pytestmark = pytest.mark.gpu


def test_yuv420_to_bgr_bit_exact_on_the_colour_lattice(gpu):
    import torch

    # every Y value against a 64 x 64 lattice of (U, V): one 2 x 2 block per chroma pair, Y varying along the frames
    us, vs = np.meshgrid(np.arange(0, 256, 4), np.arange(0, 256, 4), indexing="ij")
    h, w = 2 * 64, 2 * 64
    frames = np.empty((256, h * 3 // 2, w), np.uint8)
    q = (h // 2) * (w // 2)
    for y in range(256):
        frames[y, :h] = y
        flat = frames[y, h:].reshape(-1)
        flat[:q] = us.reshape(-1)
        flat[q:2 * q] = vs.reshape(-1)
    want = oyuv.yuv420_to_bgr(frames, h, w)
    got = scene.yuv420_to_bgr(torch.from_numpy(frames).to(gpu), h, w).cpu().numpy()
    assert np.array_equal(got, want)
    assert got.min() == 0 and got.max() == 255 and len(np.unique(got)) == 256


@pytest.mark.parametrize("h,w", [(48, 64), (1080, 1920), (30, 36), (2, 4)])
def test_yuv420_to_bgr_layouts_and_sides(gpu, h, w):
    import torch

    rng = np.random.default_rng(h * 31 + w)
    p = rng.integers(0, 256, (3, h * 3 // 2, w), dtype=np.uint8)
    want = oyuv.yuv420_to_bgr(p, h, w, "i420")
    nv = oyuv.i420_to_nv12(p, h, w)
    for layout, src in (("i420", p), ("nv12", nv)):
        assert np.array_equal(scene.yuv420_to_bgr(src, h, w, layout), want)                                   # host: staged
        assert np.array_equal(scene.yuv420_to_bgr(torch.from_numpy(src).to(gpu), h, w, layout).cpu().numpy(), want)
    with pytest.raises(Exception):
        scene.yuv420_to_bgr(p[:, :, : w - 2], h, w - 2)  # w must be a multiple of 4


def write_y4m(path, planar, h, w, fps=(30, 1)):
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F{fps[0]}:{fps[1]} Ip A1:1 C420jpeg\n".encode())
        for fr in planar:
            f.write(b"FRAME\n")
            f.write(fr.tobytes())


CONFIGS = {"scene_detection": {"threshold": 0.05}, "object_detection": {"frame_interval": 1, "confidence_threshold": 0.25},
           "face_detection": {"frame_interval": 2, "confidence_threshold": 0.3}}


def test_single_pass_on_decoder_planes_matches_oracle_derived_expectations(gpu, tmp_path):
    """A 4:2:0 clip goes through analyze_video as PLANES (1.5 B / pixel uploaded).  Expected, all from the oracle side:
    scenes = the ffmpeg-semantics scene list of the clip's own Y planes (oracle/scene.py), detections = what the detector
    returns on the BGR frames oracle/yuv.py converts those planes to (fed as a raw .npy clip: the BGR route)."""
    h, w, n = 120, 160, 230
    bgr0 = prng.synth_frames_bgr(1234, n, h, w)  # scene change at frame 198
    planar = oyuv.bgr_to_i420(bgr0)
    y4m = tmp_path / "clip.y4m"
    write_y4m(y4m, planar, h, w)
    mm = ModelManager(cache_dir=str(tmp_path / "m"), random_init_seed=7, batch_size=48)
    got = asyncio.run(mm.analyze_video(str(y4m), CONFIGS))
    # scenes: oracle on the Y planes, and the separate call (which reads the same planes from the file)
    ys = np.ascontiguousarray(planar[:, :h])
    want_scenes = oscene.detect_scenes_ffmpeg_like(ys, 0.05, 1, 30, n / 30.0)
    assert got["scene_detection"] == want_scenes == asyncio.run(mm.detect_scenes(str(y4m), CONFIGS["scene_detection"]))
    assert len(want_scenes["scenes"]) >= 1
    # detections: the BGR route on the oracle's conversion of the planes
    npy = tmp_path / "clip.npy"
    np.save(npy, oyuv.yuv420_to_bgr(planar, h, w))
    (tmp_path / "clip.npy.json").write_text(json.dumps({"fps": 30.0}))
    assert got["object_detection"] == asyncio.run(mm.detect_objects(str(npy), CONFIGS["object_detection"]))
    assert got["face_detection"] == asyncio.run(mm.detect_faces(str(npy), CONFIGS["face_detection"]))
    assert len(got["object_detection"]["detections"]) > 0
    # ContentDetector flavour: K2 on the device-converted frames == the oracle's sums on the oracle's frames
    only = asyncio.run(mm.analyze_video(str(y4m), {"scene_detection": {"detector": "content", "min_scene_len": 15}}))
    sc = oscene.content_scores(oscene.content_sums(oyuv.yuv420_to_bgr(planar, h, w)), h * w)
    cuts = oscene.content_cuts(sc, 27.0, 15)
    assert [s["start_ms"] for s in only["scene_detection"]["scenes"]] == [0] + [int(float("%.6g" % (c / 30.0)) * 1000) for c in cuts]
    assert len(cuts) >= 1


def test_cv2_capture_hands_over_planes_when_the_backend_honours_convert_rgb(gpu, monkeypatch, tmp_path):
    import sys

    from test_frames_cv2 import make_cv2

    bgr = prng.synth_frames_bgr(1234, 230, 48, 64)
    luma = np.ascontiguousarray(bgr[..., 1])
    cv2 = make_cv2(bgr, luma, 30.0, honour_convert_rgb=True, planar=True)  # raw frames: Y plane + grey chroma (I420)
    monkeypatch.setitem(sys.modules, "cv2", cv2)
    mm = ModelManager(cache_dir=str(tmp_path / "m"), random_init_seed=7, batch_size=64)
    before = len(cv2.opened)
    got = asyncio.run(mm.analyze_video("/videos/clip.mp4", {"scene_detection": {"threshold": 0.05}, "object_detection": {"frame_interval": 5}}))
    mine = cv2.opened[before:]
    assert len(mine) == 3 and mine[-1].raw and mine[-1].pos == 230  # probe, the BGR capture it replaced, the raw capture: read once
    assert got["scene_detection"] == asyncio.run(mm.detect_scenes("/videos/clip.mp4", {"threshold": 0.05}))
    # the planes converted by the oracle, through the BGR route
    planar = np.concatenate([luma, np.full((230, 24, 64), 128, np.uint8)], 1)
    npy = tmp_path / "c.npy"
    np.save(npy, oyuv.yuv420_to_bgr(planar, 48, 64))
    (tmp_path / "c.npy.json").write_text(json.dumps({"fps": 30.0}))
    assert got["object_detection"] == asyncio.run(mm.detect_objects(str(npy), {"frame_interval": 5}))
