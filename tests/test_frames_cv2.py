"""The ``cv2.VideoCapture`` seam (``/root/reference/ml-service/src/services/model_manager.py:237-299``) on a
stub ``cv2`` module: container files (``.mp4`` ...) reach ``Cv2FrameSource``, whose ``luma_planes`` used to
raise - the documented ModelManager swap failed scene_detection on every real video.

The stub serves a scripted clip the two ways OpenCV's FFmpeg backend can answer ``CAP_PROP_CONVERT_RGB = 0``:
planar I420 (the decoder's own Y plane: bit-exact with what ffmpeg's ``select`` filter scores) or, when a backend
ignores the property, BGR (luma recomputed with OpenCV's COLOR_BGR2YUV_I420 integer formula)."""
import asyncio
import sys
import types

import numpy as np
import pytest

from eioku_amd import frames as F
from oracle import prng, scene as oscene


def make_cv2(clip_bgr, clip_luma, fps, honour_convert_rgb=True, planar=True, frame_count=None):
    cv2 = types.ModuleType("cv2")
    cv2.CAP_PROP_FPS, cv2.CAP_PROP_FRAME_COUNT, cv2.CAP_PROP_CONVERT_RGB, cv2.CAP_PROP_FRAME_HEIGHT = 5, 7, 16, 4
    cv2.opened = []

    class VideoCapture:
        def __init__(self, path):
            self.path, self.pos, self.raw, self.released = path, 0, False, False
            cv2.opened.append(self)

        def get(self, prop):
            return {5: fps, 7: float(frame_count if frame_count is not None else len(clip_bgr)),
                    4: float(clip_bgr.shape[1])}.get(prop, 0.0)

        def set(self, prop, value):
            if prop == 16 and honour_convert_rgb:
                self.raw = not value
            return honour_convert_rgb

        def grab(self):
            if self.pos >= len(clip_bgr):
                return False
            self.pos += 1
            return True

        def read(self):
            if self.pos >= len(clip_bgr):
                return False, None
            i = self.pos
            self.pos += 1
            if not self.raw:
                return True, clip_bgr[i].copy()
            if not planar:
                return True, clip_luma[i].copy()
            h, w = clip_luma[i].shape
            chroma = np.full((h // 2, w), 128, np.uint8)  # U and V planes, packed under Y as OpenCV returns I420
            return True, np.concatenate([clip_luma[i], chroma], 0)

        def release(self):
            self.released = True

    cv2.VideoCapture = VideoCapture
    return cv2


@pytest.fixture
def clip():
    bgr = prng.synth_frames_bgr(1234, 230, 48, 64)  # scene change at frame 198
    luma = np.ascontiguousarray(bgr[..., 1])         # stands in for the decoder's Y plane
    return bgr, luma


@pytest.mark.parametrize("planar", [True, False])
def test_luma_planes_is_the_decoder_plane_when_convert_rgb_is_honoured(monkeypatch, clip, planar):
    bgr, luma = clip
    monkeypatch.setitem(sys.modules, "cv2", make_cv2(bgr, luma, 29.97, True, planar))
    src = F.open_video("/videos/a.mp4")
    assert isinstance(src, F.Cv2FrameSource) and src.total_frames == 230 and src.fps == 29.97
    assert src.time_base == (1000, 29970) or src.time_base[1] / src.time_base[0] == pytest.approx(29.97)
    a = src.luma_planes(0, 64)
    b = src.luma_planes(64, 64)
    assert a.dtype == np.uint8 and a.shape == (64, 48, 64) and src.luma_exact is True
    assert np.array_equal(a, luma[:64]) and np.array_equal(b, luma[64:128])
    assert np.array_equal(src.luma_planes(200, 30), luma[200:230])   # forward seek = grab()
    assert np.array_equal(src.luma_planes(10, 5), luma[10:15])       # backward = reopen
    ok, frame = src.read()                                           # the BGR capture is untouched by all this
    assert ok and np.array_equal(frame, bgr[0])
    src.release()
    assert all(c.released for c in sys.modules["cv2"].opened)


def test_luma_planes_falls_back_to_opencv_bt601_when_the_backend_returns_bgr(monkeypatch, clip):
    bgr, luma = clip
    monkeypatch.setitem(sys.modules, "cv2", make_cv2(bgr, luma, 30.0, honour_convert_rgb=False))
    src = F.open_video("/videos/a.mkv")
    y = src.luma_planes(0, 8)
    assert src.luma_exact is False and y.shape == (8, 48, 64)
    assert np.array_equal(y, F.bgr_to_luma_bt601(bgr[:8]))


def test_opencv_bt601_luma_known_answers():
    """COLOR_BGR2YUV_I420 luma of the primaries [PUBLIC-LIB: OpenCV imgproc color_yuv, ITUR_BT_601_* constants]:
    black 16, white 235, pure R 82, G 145, B 41, mid grey 126 (the studio-range BT.601 table values)."""
    px = np.array([[[0, 0, 0], [255, 255, 255], [0, 0, 255], [0, 255, 0], [255, 0, 0], [128, 128, 128]]], np.uint8)
    assert F.bgr_to_luma_bt601(px).tolist() == [[16, 235, 82, 145, 41, 126]]


def test_short_stream_ends_the_scan(monkeypatch, clip):
    bgr, luma = clip
    monkeypatch.setitem(sys.modules, "cv2", make_cv2(bgr[:100], luma[:100], 30.0, frame_count=130))  # header lies
    src = F.open_video("/videos/a.avi")
    assert src.luma_planes(64, 64).shape == (36, 48, 64)
    with pytest.raises(RuntimeError, match="no frame"):
        src.luma_planes(100, 30)


@pytest.mark.gpu
@pytest.mark.parametrize("honour", [True, False])
def test_model_manager_scene_detection_on_a_container_path(gpu, monkeypatch, tmp_path, clip, honour):
    """INTEGRATION.md routes scene_detection to the HIP ModelManager: through open_video() on an .mp4 path."""
    from eioku_amd.model_manager import ModelManager

    bgr, luma = clip
    monkeypatch.setitem(sys.modules, "cv2", make_cv2(bgr, luma, 30.0, honour_convert_rgb=honour, frame_count=236))
    mm = ModelManager(cache_dir=str(tmp_path / "m"))
    y = luma if honour else F.bgr_to_luma_bt601(bgr)
    for thr in (0.05, 0.7):
        got = asyncio.run(mm.detect_scenes("/videos/clip.mp4", {"threshold": thr}))
        # the header promised 236 frames, the stream held 230: duration follows the header like ffprobe's would
        want = oscene.detect_scenes_ffmpeg_like(y, thr, 1, 30, 236 / 30.0)
        assert got == want, (honour, thr)
