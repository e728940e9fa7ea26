"""CPU tests: scene oracle vs known answers, golden vectors and the reference-captured fixtures;
product host arithmetic (eioku_amd.scene) vs the oracle."""
import json

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import prng, scene as oscene
from eioku_amd import scene as pscene, synth


# ---- OpenCV BGR->HSV known answers (documented cv2 outputs for primaries / greys) ----------
@pytest.mark.parametrize("bgr,hsv", [
    ((0, 0, 255), (0, 255, 255)),      # red
    ((0, 255, 0), (60, 255, 255)),     # green
    ((255, 0, 0), (120, 255, 255)),    # blue
    ((0, 255, 255), (30, 255, 255)),   # yellow
    ((255, 255, 0), (90, 255, 255)),   # cyan
    ((255, 0, 255), (150, 255, 255)),  # magenta
    ((255, 255, 255), (0, 0, 255)),    # white
    ((128, 128, 128), (0, 0, 128)),    # grey
    ((0, 0, 0), (0, 0, 0)),            # black
    ((0, 0, 128), (0, 255, 128)),      # dark red
    ((128, 128, 255), (0, 127, 255)),  # light red: s = (127*4096+2048)>>12
])
def test_hsv_known_answers(bgr, hsv):
    out = oscene.bgr2hsv_u8(np.array([bgr], dtype=np.uint8))[0]
    assert tuple(int(v) for v in out) == hsv


def test_hsv_tables_match_opencv_constants():
    sdiv, hdiv = oscene.hsv_tables()
    assert sdiv[0] == 0 and hdiv[0] == 0
    assert sdiv[255] == 4096 and sdiv[1] == 255 << 12
    assert hdiv[255] == round((180 << 12) / (6.0 * 255)) and hdiv[1] == 122880


def test_hsv_hue_range_and_lattice_golden():
    g = np.load(GOLDEN / "hsv_lattice.npz")
    out = oscene.bgr2hsv_u8(g["bgr"])
    assert np.array_equal(out, g["hsv"])
    assert out[:, 0].max() < 180


def test_luma_golden_and_planted_cuts():
    g = np.load(GOLDEN / "scene_luma_64x48.npz")
    sad = oscene.luma_sad(g["luma"])
    assert np.array_equal(sad, g["sad"])
    mafd, score = oscene.ffmpeg_scene_scores(sad, 48 * 64)
    assert np.array_equal(mafd, g["mafd"]) and np.array_equal(score, g["score"])
    assert list(oscene.select_scene_cuts(score, 0.3)) == list(g["cuts_t03"])
    assert set(g["cuts_t03"]) <= set(g["planted"])  # a cut is only ever reported at a planted change
    assert score[0] == 0.0


def test_content_golden_and_planted_cuts():
    g = np.load(GOLDEN / "scene_hsv_64x48.npz")
    sums = oscene.content_sums(g["frames"])
    assert np.array_equal(sums, g["sums"])
    sc = oscene.content_scores(sums, 48 * 64)
    assert np.array_equal(sc, g["score"])
    assert oscene.content_cuts(sc, 27.0, 5) == list(g["cuts_legacy"]) == [7, 16, 25]


def test_score_is_float32_rounded_like_av_clipf():
    # 70.00000001/100 rounds to float32(0.7) < 0.7 (double): not selected, as in ffmpeg
    sad = np.array([0, 7000000001], dtype=np.uint64)
    _, score = oscene.ffmpeg_scene_scores(sad, 100000000)
    assert score[1] == float(np.float32(0.7)) and not (score[1] > 0.7)


def test_pts_time_six_significant_digits():
    assert oscene.showinfo_pts_time(37037, 1001, 30000) == "1235.8"
    assert oscene.showinfo_pts_time(1, 1001, 30000) == "0.0333667"
    assert oscene.showinfo_pts_time(150, 1, 30) == "5"


# ---- fixtures captured from the reference's own detect_scenes -------------------------------
def _parse_like_reference(stderr):
    out = []
    for line in stderr.split("\n"):
        if "showinfo" in line and "pts_time:" in line:
            try:
                out.append(int(float(line.split("pts_time:")[1].split()[0]) * 1000))
            except (ValueError, IndexError):
                continue
    return out


def _duration(stdout):
    try:
        return int(float(stdout.strip()) * 1000)
    except ValueError:
        return None


def test_scene_list_matches_reference_capture():
    cases = json.loads((GOLDEN / "ref_scenes.json").read_text())
    assert len(cases) >= 6
    for c in cases:
        ts = _parse_like_reference(c["ffmpeg_stderr"])
        dur = _duration(c["ffprobe_stdout"])
        assert oscene.build_scenes_like_reference(ts, dur) == c["result"]["scenes"], c["name"]
        assert pscene.build_scenes(ts, dur) == c["result"]["scenes"], c["name"]


# ---- product host arithmetic vs oracle -------------------------------------------------------
def test_host_ffmpeg_scores_match_oracle():
    rng = np.random.default_rng(5)
    for _ in range(20):
        n = int(rng.integers(1, 40))
        count = int(rng.integers(1, 5000))
        sad = rng.integers(0, 255 * count, n).astype(np.uint64)
        sad[0] = 0
        m0, s0 = oscene.ffmpeg_scene_scores(sad, count)
        m1, s1 = pscene.ffmpeg_scene_scores(sad, count)
        assert np.array_equal(m0, m1) and np.array_equal(s0, s1)
        # streaming continuation
        m0, s0 = oscene.ffmpeg_scene_scores(sad, count, prev_mafd=3.25, first_has_prev=True)
        m1, s1 = pscene.ffmpeg_scene_scores(sad, count, prev_mafd=3.25, first_has_prev=True)
        assert np.array_equal(m0, m1) and np.array_equal(s0, s1)


def test_host_content_scores_and_cuts_match_oracle():
    rng = np.random.default_rng(6)
    for trial in range(30):
        n = int(rng.integers(1, 200))
        npx = int(rng.integers(1, 10000))
        sums = rng.integers(0, 255 * npx, (n, 3)).astype(np.uint64)
        a = oscene.content_scores(sums, npx)
        b = pscene.content_scores(sums, npx)
        assert np.array_equal(a, b)
        sc = rng.uniform(0, 60, n)
        sc[0] = 0
        for mode in ("legacy", "merge", "suppress"):
            for msl in (0, 1, 5, 15):
                assert oscene.content_cuts(sc, 27.0, msl, mode) == pscene.content_cuts(sc, 27.0, msl, mode), (trial, mode, msl)


def test_pts_string_and_frame_params_agree():
    for k in (0, 1, 89, 37037, 107999):
        assert oscene.showinfo_pts_time(k, 1001, 30000) == pscene.pts_time_string(k, 1001, 30000)
    a = synth.frame_params(1234, 500, first_frame=7)
    b = prng.scene_params(1234, prng.scene_schedule(1234, 507))[7:]
    assert np.array_equal(a, b)
    assert synth.splitmix64_scalar(42, 5) == int(prng.splitmix64(42, 1, offset=5)[0])


def test_end_to_end_oracle_on_synthetic_clip():
    fr = prng.synth_frames_bgr(1234, 220, 24, 32)
    y = np.ascontiguousarray(fr[..., 1])
    r = oscene.detect_scenes_ffmpeg_like(y, 0.05, 1, 30, 220 / 30)
    assert r["scenes"][-1]["end_ms"] == int(220 / 30 * 1000)
    sums = oscene.content_sums(fr)
    cuts = oscene.content_cuts(oscene.content_scores(sums, 24 * 32))
    assert cuts == [198]
