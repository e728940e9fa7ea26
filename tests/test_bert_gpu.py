"""GPU parity: K8 MiniLM encoder (fp32 MFMA GEMMs) vs the float64 numpy oracle.
Bar (BASELINE.json): embeddings within 1e-4 relative; unit-norm vectors, so elementwise
|got - want| <= 1e-4 * max|want| (and far below in practice: fp32 end to end)."""
import numpy as np
import pytest

from conftest import GOLDEN
from oracle import bert as obert
from eioku_amd import embed

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _inputs(vocab, B, S, seed, ragged=True):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1, vocab, (B, S)).astype(np.int32)
    mask = np.ones((B, S), dtype=np.uint8)
    if ragged:
        for b in range(B):
            n = int(rng.integers(1, S + 1))
            mask[b, n:] = 0
            ids[b, n:] = 0
    return ids, mask


def _close(got, want):
    return np.abs(got - want).max() <= RTOL * np.abs(want).max()


def test_golden_minilm(gpu):
    g = np.load(GOLDEN / "minilm_seed11.npz")
    cfg = dict(embed.MINILM_L6_V2, vocab=int(g["vocab"]))
    enc = embed.MiniLMEncoder(embed.random_state(cfg, int(g["seed"])), cfg)
    out = enc.encode_ids(g["ids"], g["mask"])
    assert _close(out, g["out"]), float(np.abs(out - g["out"]).max())
    enc.close()


@pytest.mark.parametrize("B,S,ragged", [(8, 128, True), (1, 7, False), (3, 33, True), (5, 256, True), (130, 16, True)])
def test_encoder_matches_oracle(gpu, B, S, ragged):
    import torch

    cfg = dict(embed.MINILM_L6_V2, vocab=3000)
    state = embed.random_state(cfg, 3)
    enc = embed.MiniLMEncoder(state, cfg)
    ids, mask = _inputs(cfg["vocab"], B, S, B * 100 + S, ragged)
    want = obert.encode(state, cfg, ids, mask)
    got = enc.encode_ids(ids, mask)
    assert got.shape == (B, 384) and _close(got, want), float(np.abs(got - want).max())
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    # device-resident inputs: same bytes
    got_d = enc.encode_ids(torch.from_numpy(ids).to(gpu), torch.from_numpy(mask).to(gpu)).cpu().numpy()
    assert np.array_equal(got_d, got)
    assert enc.last_flops() > 0
    enc.close()


def test_empty_segment_and_errors(gpu):
    from eioku_amd._lib import EiokuHipError

    cfg = dict(embed.MINILM_L6_V2, vocab=3000)
    enc = embed.MiniLMEncoder(embed.random_state(cfg, 3), cfg)
    ids = np.zeros((2, 8), np.int32)
    mask = np.zeros((2, 8), np.uint8)
    ids[1, :3] = [5, 6, 7]
    mask[1, :3] = 1
    out = enc.encode_ids(ids, mask)
    assert not out[0].any()  # fully masked segment pools to the zero vector, as sentence-transformers does
    assert abs(np.linalg.norm(out[1]) - 1) < 1e-5
    assert enc.encode_ids(np.zeros((0, 8), np.int32), np.zeros((0, 8), np.uint8)).shape == (0, 384)
    bad = ids.copy()
    bad[1, 0] = 999999
    with pytest.raises(EiokuHipError):
        enc.encode_ids(bad, mask)
    with pytest.raises(EiokuHipError):
        enc.encode_ids(np.zeros((1, 600), np.int32), np.ones((1, 600), np.uint8))  # > 512 positions
    with pytest.raises(RuntimeError):
        enc.encode(["no tokenizer configured"])
    enc.close()
    bare = embed.MiniLMEncoder(None, cfg)
    with pytest.raises(EiokuHipError):
        bare.encode_ids(ids, mask)
    bare.close()


def test_batch_512x128_properties(gpu):
    """cfg3 size: (1) rows of a big batch equal the same rows encoded alone (batch independence);
    (2) permuting segments permutes outputs; (3) an oracle spot check on 2 of the 512 segments."""
    cfg = dict(embed.MINILM_L6_V2)
    state = embed.random_state(cfg, 7)
    enc = embed.MiniLMEncoder(state, cfg)
    ids, mask = _inputs(cfg["vocab"], 512, 128, 99, ragged=True)
    out = enc.encode_ids(ids, mask)
    sub = enc.encode_ids(ids[[3, 200, 511]], mask[[3, 200, 511]])
    assert np.allclose(sub, out[[3, 200, 511]], rtol=0, atol=2e-6)
    perm = np.random.default_rng(0).permutation(512)
    assert np.allclose(enc.encode_ids(ids[perm], mask[perm]), out[perm], rtol=0, atol=2e-6)
    want = obert.encode(state, cfg, ids[[3, 200]], mask[[3, 200]])
    assert _close(out[[3, 200]], want)
    enc.close()


def test_full_size_batch_invariance(gpu):
    """BASELINE cfg3 size (512 x 128): a segment's embedding does not depend on its batch.  Bit-exact for
    batches that take the same GEMM configuration (tile size and split-K are chosen from the token count),
    within the parity tolerance across configurations."""
    cfg = dict(embed.MINILM_L6_V2, vocab=3000)
    enc = embed.MiniLMEncoder(embed.random_state(cfg, 5), cfg)
    ids, mask = _inputs(3000, 512, 128, 9)
    full = enc.encode_ids(ids, mask)
    assert np.isfinite(full).all() and np.allclose(np.linalg.norm(full, axis=1), 1.0, atol=1e-5)
    a = enc.encode_ids(ids[:256], mask[:256])
    b = enc.encode_ids(ids[256:], mask[256:])
    assert np.array_equal(np.concatenate([a, b]), full)      # 32768 / 65536 tokens: same kernels, same order
    small = enc.encode_ids(ids[:8], mask[:8])                 # 1024 tokens: 64x64 tiles + split-K planes
    assert _close(small, full[:8])
    again = enc.encode_ids(ids[:8], mask[:8])
    assert np.array_equal(small, again)                       # split-K planes are summed in plane order
    enc.close()


def test_fp32_fma_route_matches_the_mfma_route(gpu, tmp_path):
    """The library keeps its fp32-FMA GEMM / attention kernels reachable (EIOKU_GEMM_BF16=0, EIOKU_GEMM_S=0,
    EIOKU_ATTN_MFMA=0; read once per process): the same golden inputs through that route meet the same bar, and the
    two routes agree far inside it -- an independent check on the split-bf16 MFMA numerics."""
    import os
    import subprocess
    import sys

    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from eioku_amd import embed\n"
        "g = np.load(%r)\n"
        "cfg = dict(embed.MINILM_L6_V2, vocab=int(g['vocab']))\n"
        "enc = embed.MiniLMEncoder(embed.random_state(cfg, int(g['seed'])), cfg)\n"
        "np.save(sys.argv[1], enc.encode_ids(g['ids'], g['mask']))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(GOLDEN / "minilm_seed11.npz"))
    want = np.load(GOLDEN / "minilm_seed11.npz")["out"]
    outs = {}
    for name, env in {"mfma": {}, "fma": {"EIOKU_GEMM_BF16": "0", "EIOKU_GEMM_S": "0", "EIOKU_ATTN_MFMA": "0"},
                      "fma_s": {"EIOKU_GEMM_BF16": "0"}, "no_planes": {"EIOKU_GEMM_PLANES": "0"},
                      "attn_f32": {"EIOKU_ATTN_BF16": "0"}}.items():
        path = tmp_path / f"{name}.npy"
        subprocess.run([sys.executable, "-c", code, str(path)], check=True, env=dict(os.environ, **env), timeout=300)
        outs[name] = np.load(path)
        assert _close(outs[name], want), (name, float(np.abs(outs[name] - want).max()))
    assert np.abs(outs["mfma"] - outs["fma"]).max() <= 2e-5
    assert np.abs(outs["fma_s"] - outs["fma"]).max() <= 2e-5
    # r3: activations as bf16 planes written by their producers vs fp32 activations split inside the GEMM's staging
    # threads - the same split of the same values, so the same bytes
    assert np.array_equal(outs["no_planes"], outs["mfma"])
    # r3: attention on split bf16 (default) vs on the exact-fp32 matrix pipe
    assert np.abs(outs["attn_f32"] - outs["mfma"]).max() <= 2e-5
