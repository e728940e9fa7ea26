"""GPU parity: K1 luma SAD and K2 HSV sums (HIP, through the C ABI) vs the CPU oracle. Bit-exact."""
import numpy as np
import pytest

from conftest import GOLDEN
from oracle import prng, scene as oscene
from eioku_amd import scene, synth

pytestmark = pytest.mark.gpu


def _to_dev(a, dev):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_synth_frames_device_equals_oracle(gpu):
    for (n, h, w, first) in [(3, 17, 23, 0), (5, 48, 64, 148), (2, 1, 1, 9)]:
        d = synth.frames_bgr(77, n, h, w, gpu, first_frame=first)
        o = prng.synth_frames_bgr(77, n, h, w, first_frame=first)
        assert np.array_equal(d.cpu().numpy(), o)


def test_synth_streams_device_equal_oracle(gpu):
    assert np.array_equal(synth.u64(5, 1000, gpu, offset=12345).cpu().numpy().view(np.uint64),
                          prng.splitmix64(5, 1000, offset=12345))
    assert np.array_equal(synth.bytes_u8(9, 1003, gpu).cpu().numpy(), prng.uniform_bytes(9, 1003))
    x = synth.normal_f32(3, 64, 384, gpu).cpu().numpy()
    assert np.array_equal(x.reshape(-1), prng.approx_normal_f32(3, 64 * 384))
    assert abs(float(x.std()) - 1.0) < 0.02
    u = synth.normal_f32(3, 64, 384, gpu, l2_normalise=True).cpu().numpy()
    assert np.allclose(np.linalg.norm(u, axis=1), 1.0, atol=1e-6)


def test_bgr2hsv_lattice_golden(gpu):
    g = np.load(GOLDEN / "hsv_lattice.npz")
    out_dev = scene.bgr2hsv(_to_dev(g["bgr"], gpu)).cpu().numpy()
    assert np.array_equal(out_dev, g["hsv"])
    out_host = scene.bgr2hsv(np.ascontiguousarray(g["bgr"]))  # EIOKU_MEM_HOST staging path
    assert np.array_equal(out_host, g["hsv"])


def test_bgr2hsv_exhaustive_slice(gpu):
    """All 2^24 colours would be 48 MB; check every (b,g) pair for 16 spread r levels."""
    lv = np.arange(256, dtype=np.uint8)
    rs = np.array([0, 1, 2, 17, 63, 64, 100, 127, 128, 129, 200, 250, 253, 254, 255, 31], dtype=np.uint8)
    b, g, r = np.meshgrid(lv, lv, rs, indexing="ij")
    bgr = np.ascontiguousarray(np.stack([b, g, r], -1).reshape(-1, 3))
    assert np.array_equal(scene.bgr2hsv(_to_dev(bgr, gpu)).cpu().numpy(), oscene.bgr2hsv_u8(bgr))


def test_luma_sad_golden(gpu):
    g = np.load(GOLDEN / "scene_luma_64x48.npz")
    assert np.array_equal(scene.luma_sad(_to_dev(g["luma"], gpu)), g["sad"])
    assert np.array_equal(scene.luma_sad(np.ascontiguousarray(g["luma"])), g["sad"])  # host pointers


def test_hsv_sums_golden(gpu):
    g = np.load(GOLDEN / "scene_hsv_64x48.npz")
    sums = scene.hsv_sums(_to_dev(g["frames"], gpu))
    assert np.array_equal(sums, g["sums"])
    assert np.array_equal(scene.hsv_sums(np.ascontiguousarray(g["frames"])), g["sums"])
    sc = scene.content_scores(sums, 48 * 64)
    assert np.array_equal(sc, g["score"])
    assert scene.content_cuts(sc, 27.0, 5) == list(g["cuts_legacy"])
    assert scene.content_cuts(sc, 27.0, 5, mode="merge") == list(g["cuts_merge"])


# ragged shapes: plane sizes that are not multiples of 16 bytes / 4 pixels, single frames,
# frame counts around the in-register group (8) and run boundaries
@pytest.mark.parametrize("n,h,w", [(1, 5, 7), (2, 1, 1), (9, 3, 5), (17, 31, 33), (8, 16, 16), (40, 37, 41),
                                   (3, 64, 65), (33, 2, 8), (70, 9, 11)])
def test_ragged_shapes_random_bytes(gpu, n, h, w):
    rng = np.random.default_rng(n * 1000 + h * 10 + w)
    y = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    assert np.array_equal(scene.luma_sad(_to_dev(y, gpu)), oscene.luma_sad(y))
    f = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    assert np.array_equal(scene.hsv_sums(_to_dev(f, gpu)), oscene.content_sums(f))


def test_prev_frame_streaming_equals_whole(gpu):
    """Chunked processing with the carried previous frame equals one pass (streaming contract)."""
    rng = np.random.default_rng(3)
    y = rng.integers(0, 256, (21, 19, 35), dtype=np.uint8)
    f = rng.integers(0, 256, (21, 19, 35, 3), dtype=np.uint8)
    whole_y = oscene.luma_sad(y)
    whole_f = oscene.content_sums(f)
    got_y, got_f = [], []
    for a, b in [(0, 5), (5, 6), (6, 21)]:
        py = _to_dev(y[a - 1], gpu) if a else None
        pf = _to_dev(f[a - 1], gpu) if a else None
        got_y.append(scene.luma_sad(_to_dev(y[a:b], gpu), prev=py))
        got_f.append(scene.hsv_sums(_to_dev(f[a:b], gpu), prev=pf))
    assert np.array_equal(np.concatenate(got_y), whole_y)
    assert np.array_equal(np.concatenate(got_f), whole_f)


def test_luma_row_strided_planes(gpu):
    """Decoder-style padded planes (linesize > width) take the strided kernel."""
    import torch

    rng = np.random.default_rng(4)
    n, h, w, ls = 6, 13, 37, 64
    buf = rng.integers(0, 256, (n, h + 2, ls), dtype=np.uint8)  # 2 rows of padding between planes
    y = np.ascontiguousarray(buf[:, :h, :w])
    d = torch.from_numpy(buf).to(gpu)
    got = scene.luma_sad(d, row_stride=ls, frame_stride=(h + 2) * ls, shape=(n, h, w))
    assert np.array_equal(got, oscene.luma_sad(y))


def test_extremes_and_identical_frames(gpu):
    z = np.zeros((4, 32, 32), dtype=np.uint8)
    o = np.full((4, 32, 32), 255, dtype=np.uint8)
    alt = np.stack([z[0], o[0], z[0], o[0]])
    assert list(scene.luma_sad(_to_dev(alt, gpu))) == [0, 255 * 1024, 255 * 1024, 255 * 1024]
    assert not scene.luma_sad(_to_dev(z, gpu)).any()
    f = np.zeros((3, 8, 8, 3), dtype=np.uint8)
    f[1] = 255  # black -> white: dH=0, dS=0, dV=255 per pixel
    s = scene.hsv_sums(_to_dev(f, gpu))
    assert s.tolist() == [[0, 0, 0], [0, 0, 255 * 64], [0, 0, 255 * 64]]


def test_empty_batch_and_bad_args(gpu, built_lib):
    import torch
    from eioku_amd._lib import EiokuHipError

    e = torch.empty((0, 8, 8), dtype=torch.uint8, device=gpu)
    assert scene.luma_sad(e).shape == (0,)
    e3 = torch.empty((0, 8, 8, 3), dtype=torch.uint8, device=gpu)
    assert scene.hsv_sums(e3).shape == (0, 3)
    with pytest.raises(EiokuHipError):
        scene.luma_sad(torch.zeros((2, 4, 4), dtype=torch.uint8, device=gpu), row_stride=2, shape=(2, 4, 4))


def test_full_size_1080p_properties(gpu):
    """BASELINE size (64 x 1080p): size-independent properties instead of a slow oracle pass.
    (1) device-synth frames: sums over a run equal chunked sums with carried prev (checksum of
    checksums); (2) an oracle spot-check on 2 frames; (3) planted scene change is the only cut."""
    n, h, w = 64, 1080, 1920
    frames = synth.frames_bgr(1234, n, h, w, gpu, first_frame=160)  # scene change at absolute frame 198
    whole = scene.hsv_sums(frames)
    parts = [scene.hsv_sums(frames[:20]), scene.hsv_sums(frames[20:41], prev=frames[19]),
             scene.hsv_sums(frames[41:], prev=frames[40])]
    assert np.array_equal(np.concatenate(parts), whole)
    two = frames[37:39].cpu().numpy()
    assert np.array_equal(oscene.content_sums(two)[1], whole[38])
    cuts = scene.content_cuts(scene.content_scores(whole, h * w))
    assert cuts == [198 - 160]
    y = frames[..., 1].contiguous()
    sad = scene.luma_sad(y)
    assert np.array_equal(sad[38], oscene.luma_sad(two[..., 1])[1])
    assert np.array_equal(np.concatenate([scene.luma_sad(y[:33]), scene.luma_sad(y[33:], prev=y[32])]), sad)


def test_scores_and_cuts_behind_the_c_abi(gpu):
    """libavfilter's score (double -> float32 clip) and the PySceneDetect cut filter live behind the C
    ABI (eioku_scene_scores_luma / eioku_scene_content): host and device inputs, with and without a carried previous
    frame, against the oracle - bit for bit (the scores are float64 images of exact integer sums)."""
    import torch

    n, h, w = 40, 72, 96
    host = prng.synth_frames_bgr(1234, n, h, w, first_frame=185)
    y = np.ascontiguousarray(host[..., 1])
    sad = oscene.luma_sad(y)
    mafd_o, score_o = oscene.ffmpeg_scene_scores(sad, h * w)
    for src in (y, torch.from_numpy(y).to(gpu)):
        mafd, score = scene.scene_scores_luma(src)
        assert np.array_equal(mafd, mafd_o) and np.array_equal(score, score_o)
    # streaming: the second half scored with the first half's last plane / mafd carried over
    k = 17
    m1, s1 = scene.scene_scores_luma(torch.from_numpy(y[:k]).to(gpu))
    m2, s2 = scene.scene_scores_luma(torch.from_numpy(y[k:]).to(gpu), prev=torch.from_numpy(y[k - 1]).to(gpu), prev_mafd=float(m1[-1]))
    assert np.array_equal(np.concatenate([m1, m2]), mafd_o) and np.array_equal(np.concatenate([s1, s2]), score_o)
    sums = oscene.content_sums(host)
    sc_o = oscene.content_scores(sums, h * w)
    for mode in ("legacy", "merge"):
        for msl in (0, 3, 15):
            want = oscene.content_cuts(sc_o, 27.0, msl, mode)
            for src in (host, torch.from_numpy(host).to(gpu)):
                cuts, sc = scene.content_detect(src, threshold=27.0, min_scene_len=msl, mode=mode)
                assert cuts == want and np.array_equal(sc, sc_o), (mode, msl)
    assert len(oscene.content_cuts(sc_o, 27.0, 3)) >= 1  # the synthetic video does hold cuts
