"""Scene-cut stage oracle (rows a1 and a1' of SURVEY.md §8).

Test infrastructure (see ``oracle/__init__.py``).  **Parity unpinned** for the
arithmetic (no ffmpeg / cv2 / scenedetect in the build container and the reference
tests hold no vectors); the scene-list construction is pinned by
``tests/golden/ref_scenes_*.json`` captured from the reference's own code.

a1  - what the reference really runs: ``ffmpeg -vf select='gt(scene\\,T)',showinfo``
      (``ml-service/src/services/model_manager.py:736-745``), then parses ``pts_time:``
      and builds the scene dicts (``model_manager.py:758-828``).
a1' - PySceneDetect ``ContentDetector`` named by BASELINE.json north_star; the
      reference has no code for it (intent only: ``.kiro/specs/semantic-video-search/
      design.md:59-61``).
"""

from __future__ import annotations

import numpy as np

# ---------------------------------------------------------------------------
# a1: libavfilter f_select.c ``get_scene_score``  [PUBLIC-LIB: ffmpeg 4.3+]
# ---------------------------------------------------------------------------


def luma_sad(y_frames: np.ndarray, prev: np.ndarray | None = None) -> np.ndarray:
    """``sad[t] = sum |Y_t - Y_{t-1}|`` over plane 0, uint64; ``sad[0]`` uses ``prev`` (0 if None).

    For planar YUV only the luma plane enters the score (``nb_planes = is_yuv ? 1 : ...``).
    """
    y = np.asarray(y_frames)
    assert y.dtype == np.uint8 and y.ndim == 3
    n = y.shape[0]
    sad = np.zeros(n, dtype=np.uint64)
    for t in range(n):
        p = y[t - 1] if t > 0 else prev
        if p is None:
            continue
        sad[t] = np.abs(y[t].astype(np.int16) - p.astype(np.int16)).sum(dtype=np.uint64)
    return sad


def ffmpeg_scene_scores(sad: np.ndarray, count: int, bitdepth: int = 8, prev_mafd: float = 0.0,
                        first_has_prev: bool = False):
    """Scores exactly as f_select.c computes them, frame by frame.

    ``mafd = (double)sad / count / (1 << (bitdepth-8))``; ``diff = fabs(mafd - prev_mafd)``;
    ``ret = av_clipf(FFMIN(mafd, diff) / 100., 0, 1)``.  ``av_clipf`` takes and returns
    *float*, so the double quotient is rounded to float32 before it is compared with the
    threshold (as a double) by the ``gt(scene,T)`` expression.  The first frame has no
    previous picture and scores 0 (prev_mafd stays at its zero initialisation).

    Returns ``(mafd[n] float64, score[n] float64)``.
    """
    n = len(sad)
    mafd = np.zeros(n, dtype=np.float64)
    score = np.zeros(n, dtype=np.float64)
    pm = float(prev_mafd)
    for t in range(n):
        if t == 0 and not first_has_prev:
            continue
        m = float(sad[t]) / float(count) / float(1 << (bitdepth - 8))
        d = abs(m - pm)
        q = min(m, d) / 100.0
        f = np.float32(q)  # double -> float at the av_clipf call
        f = np.float32(0.0) if f < 0 else (np.float32(1.0) if f > 1 else f)
        mafd[t] = m
        score[t] = float(f)
        pm = m
    return mafd, score


def showinfo_pts_time(frame_index: int, tb_num: int, tb_den: int, pts_per_frame: int = 1) -> str:
    """``pts_time:`` field printed by vf_showinfo: ``av_ts2timestr`` = ``"%.6g" % (av_q2d(tb) * pts)``."""
    t = (tb_num / float(tb_den)) * (frame_index * pts_per_frame)
    return "%.6g" % t


def select_scene_cuts(score: np.ndarray, threshold: float) -> np.ndarray:
    """Frames the ``select='gt(scene,T)'`` expression lets through (strict ``>``)."""
    return np.nonzero(np.asarray(score, dtype=np.float64) > float(threshold))[0].astype(np.int64)


def build_scenes_like_reference(cut_timestamps_ms: list[int], duration_ms: int | None) -> list[dict]:
    """Scene dicts with the reference's index quirk (``model_manager.py:758-828``).

    With cuts t1<...<tn the reference emits ``{idx 0: t1->t2} ... {idx n-2: t(n-1)->tn}`` and a
    final ``{idx n: tn->dur}``: the leading ``0->t1`` scene is never emitted and index ``n-1`` is
    skipped.  With no cuts: one ``{0, 0->dur}`` scene.  ``duration_ms`` None models an ffprobe
    failure (``prev + 1000`` fallback, ``:802-805``).
    """
    scenes = []
    scene_idx = 0
    prev = 0
    for ts in cut_timestamps_ms:
        if scene_idx > 0:
            scenes.append({"scene_index": scene_idx - 1, "start_ms": prev, "end_ms": ts,
                           "duration_ms": ts - prev})
        prev = ts
        scene_idx += 1
    if duration_ms is None:
        duration_ms = prev + 1000
    if scene_idx > 0:
        scenes.append({"scene_index": scene_idx, "start_ms": prev, "end_ms": duration_ms,
                       "duration_ms": duration_ms - prev})
    else:
        scenes.append({"scene_index": 0, "start_ms": 0, "end_ms": duration_ms,
                       "duration_ms": duration_ms})
    return scenes


def detect_scenes_ffmpeg_like(y_frames: np.ndarray, threshold: float, tb_num: int, tb_den: int,
                              duration_s: float | None) -> dict:
    """End-to-end a1 oracle on raw luma planes: what ``detect_scenes`` would return."""
    n, h, w = y_frames.shape
    sad = luma_sad(y_frames)
    _, score = ffmpeg_scene_scores(sad, h * w)
    cuts = select_scene_cuts(score, threshold)
    ts_ms = [int(float(showinfo_pts_time(int(c), tb_num, tb_den)) * 1000) for c in cuts]
    dur_ms = None if duration_s is None else int(float(duration_s) * 1000)
    return {"scenes": build_scenes_like_reference(ts_ms, dur_ms)}


# ---------------------------------------------------------------------------
# a1': OpenCV 8-bit BGR->HSV + PySceneDetect ContentDetector  [PUBLIC-LIB]
# ---------------------------------------------------------------------------

HSV_SHIFT = 12


def _cv_round_div(num: int, den: float) -> int:
    # saturate_cast<int>(double) == cvRound == lrint (round half to even)
    return int(np.rint(num / den))


def hsv_tables():
    """``sdiv_table`` / ``hdiv_table180`` of OpenCV's RGB2HSV_b (imgproc/color_hsv)."""
    sdiv = np.zeros(256, dtype=np.int32)
    hdiv = np.zeros(256, dtype=np.int32)
    for i in range(1, 256):
        sdiv[i] = _cv_round_div(255 << HSV_SHIFT, 1.0 * i)
        hdiv[i] = _cv_round_div(180 << HSV_SHIFT, 6.0 * i)
    return sdiv, hdiv


_SDIV, _HDIV = hsv_tables()


def bgr2hsv_u8(bgr: np.ndarray) -> np.ndarray:
    """``cv2.cvtColor(frame, cv2.COLOR_BGR2HSV)`` for uint8 input (H in [0,180))."""
    a = np.asarray(bgr)
    assert a.dtype == np.uint8 and a.shape[-1] == 3
    b = a[..., 0].astype(np.int32)
    g = a[..., 1].astype(np.int32)
    r = a[..., 2].astype(np.int32)
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    s = (diff * _SDIV[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT  # arithmetic shift (floor)
    h = h + np.where(h < 0, 180, 0)
    out = np.empty(a.shape, dtype=np.uint8)
    out[..., 0] = np.clip(h, 0, 255)
    out[..., 1] = s
    out[..., 2] = v
    return out


def content_sums(bgr_frames: np.ndarray, prev: np.ndarray | None = None) -> np.ndarray:
    """Per-frame integer sums ``sum|c_t - c_{t-1}|`` for c in (hue, sat, lum); uint64 (n,3).

    Plain ``|.|`` on hue (not circular), as ``ContentDetector._mean_pixel_distance`` does.
    """
    f = np.asarray(bgr_frames)
    n = f.shape[0]
    sums = np.zeros((n, 3), dtype=np.uint64)
    last = None if prev is None else bgr2hsv_u8(prev).astype(np.int32)
    for t in range(n):
        cur = bgr2hsv_u8(f[t]).astype(np.int32)
        if last is not None:
            sums[t] = np.abs(cur - last).reshape(-1, 3).sum(axis=0, dtype=np.uint64)
        last = cur
    return sums


def content_scores(sums: np.ndarray, num_pixels: int, first_has_prev: bool = False) -> np.ndarray:
    """``frame_score = (1*dh + 1*ds + 1*dl + 0*de) / 3`` in float64; first frame scores 0.

    Each delta is ``np.sum(abs(...)) / float(num_pixels)``; the weighted sum is Python's
    left-to-right ``sum`` starting from int 0; divided by ``sum(abs(w)) == 3.0``.
    """
    n = sums.shape[0]
    out = np.zeros(n, dtype=np.float64)
    npx = float(num_pixels)
    for t in range(n):
        if t == 0 and not first_has_prev:
            continue
        dh = float(sums[t, 0]) / npx
        ds = float(sums[t, 1]) / npx
        dl = float(sums[t, 2]) / npx
        acc = 0
        for comp, wgt in ((dh, 1.0), (ds, 1.0), (dl, 1.0), (0.0, 0.0)):
            acc = acc + comp * wgt
        out[t] = acc / 3.0
    return out


def content_cuts(scores: np.ndarray, threshold: float = 27.0, min_scene_len: int = 15,
                 mode: str = "legacy") -> list[int]:
    """Cut frame numbers from the per-frame score series.

    ``legacy`` (PySceneDetect 0.6.0-0.6.3 ``process_frame``): ``last_cut`` starts at the first
    frame; frame n is a cut iff ``score >= threshold and n - last_cut >= min_scene_len``.
    ``merge`` / ``suppress``: the 0.6.4+ ``FlashFilter`` modes (MERGE is that version's default).
    The first frame has no score and can never cut.
    """
    n = len(scores)
    cuts: list[int] = []
    if mode == "legacy" or mode == "suppress":
        last = 0
        for t in range(1, n):
            if scores[t] >= threshold and (t - last) >= min_scene_len:
                cuts.append(t)
                last = t
        return cuts
    if mode != "merge":
        raise ValueError(f"unknown mode {mode!r}")
    last_above = None
    merge_enabled = False
    merge_triggered = False
    merge_start = None
    for t in range(n):  # the filter also sees the first frame (score 0.0), which seeds last_above
        above = bool(t > 0 and scores[t] >= threshold)
        if not min_scene_len > 0:
            if above:
                cuts.append(t)
            continue
        if last_above is None:
            last_above = t
        min_length_met = (t - last_above) >= min_scene_len
        if above:
            last_above = t
        if merge_triggered:
            num_merged = last_above - merge_start
            if min_length_met and not above and num_merged >= min_scene_len:
                merge_triggered = False
                cuts.append(last_above)
            continue
        if not above:
            continue
        if min_length_met:
            merge_enabled = True
            cuts.append(t)
            continue
        if merge_enabled:
            merge_triggered = True
            merge_start = t
    return cuts
