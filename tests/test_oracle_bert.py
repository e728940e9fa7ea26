"""CPU tests: the numpy MiniLM oracle vs an independent implementation of the same graph
(transformers.BertModel built from a local config - no download), and its golden vector."""
import numpy as np
import pytest

from conftest import GOLDEN
from oracle import bert as obert
from eioku_amd import embed

SMALL = dict(vocab=500, hidden=128, layers=2, heads=4, ffn=256, max_pos=64, type_vocab=2, ln_eps=1e-12)


def _inputs(cfg, B, S, seed, ragged=True):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1, cfg["vocab"], (B, S)).astype(np.int32)
    mask = np.ones((B, S), dtype=np.uint8)
    if ragged:
        for b in range(B):
            n = int(rng.integers(1, S + 1))
            mask[b, n:] = 0
            ids[b, n:] = 0
    return ids, mask


def _hf_encode(state, cfg, ids, mask):
    import torch
    from transformers import BertConfig, BertModel

    hf = BertModel(BertConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"],
                              num_attention_heads=cfg["heads"], intermediate_size=cfg["ffn"],
                              max_position_embeddings=cfg["max_pos"], type_vocab_size=cfg["type_vocab"],
                              layer_norm_eps=cfg["ln_eps"], hidden_act="gelu", attn_implementation="eager"),
                   add_pooling_layer=False).double().eval()
    missing = hf.load_state_dict({k: torch.from_numpy(np.asarray(v, np.float64)) for k, v in state.items()}, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k]
    with torch.no_grad():
        out = hf(input_ids=torch.from_numpy(ids.astype(np.int64)), attention_mask=torch.from_numpy(mask.astype(np.int64))).last_hidden_state
        w = torch.from_numpy(mask.astype(np.float64))[:, :, None]
        pooled = (out * w).sum(1) / w.sum(1).clamp(min=1e-9)
        return torch.nn.functional.normalize(pooled, p=2, dim=1).numpy()


@pytest.mark.parametrize("cfg", [SMALL, dict(embed.MINILM_L6_V2, vocab=2000)])
def test_oracle_matches_transformers_bertmodel(cfg):
    state = embed.random_state(cfg, 11)
    ids, mask = _inputs(cfg, 3, 16, 5)
    a = obert.encode(state, cfg, ids, mask)
    b = _hf_encode(state, cfg, ids, mask)
    assert np.allclose(a, b, rtol=1e-9, atol=1e-10)
    assert np.allclose(np.linalg.norm(a, axis=1), 1.0)


def test_oracle_golden_and_padding_invariance():
    g = np.load(GOLDEN / "minilm_seed11.npz")
    cfg = dict(embed.MINILM_L6_V2, vocab=int(g["vocab"]))
    state = embed.random_state(cfg, int(g["seed"]))
    out = obert.encode(state, cfg, g["ids"], g["mask"])
    assert np.allclose(out, g["out"], rtol=1e-10, atol=1e-12)
    # extra padding columns (mask 0) do not change the embedding
    ids2 = np.concatenate([g["ids"], np.zeros((2, 5), np.int32)], 1)
    mask2 = np.concatenate([g["mask"], np.zeros((2, 5), np.uint8)], 1)
    assert np.allclose(obert.encode(state, cfg, ids2, mask2), out, rtol=1e-9, atol=1e-11)


def test_tensor_table_matches_library(built_lib):
    import ctypes as C

    if built_lib is None:
        pytest.skip("library not built")
    # names/shapes the C++ side registers are the ones the host side fills (no GPU call: create needs init)
    names = [n for n, _ in embed.tensor_table(embed.MINILM_L6_V2)]
    assert len(names) == 5 + 16 * 6 and len(set(names)) == len(names)
