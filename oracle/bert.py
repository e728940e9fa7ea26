"""Segment-embedding oracle (row a5 of SURVEY.md §8): all-MiniLM-L6-v2 forward in numpy.

Test infrastructure (see ``oracle/__init__.py``).  **Parity unpinned** against the real model:
sentence-transformers is neither a dependency of the reference (intent only,
``.kiro/specs/semantic-video-search/design.md:54-57``) nor installed, and no checkpoint is
available offline.  The published architecture is restated [PUBLIC-LIB] - BERT post-LN encoder,
GELU(erf), attention-mask mean pooling, L2 normalise - and cross-checked in
``tests/test_oracle_bert.py`` against ``transformers.BertModel`` built from a local config object
with the same weights (an independent implementation of the same graph).
"""
from __future__ import annotations

import math

import numpy as np


def layernorm(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def gelu(x):
    from scipy.special import erf

    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def encode(state: dict, cfg: dict, ids: np.ndarray, mask: np.ndarray, dtype=np.float64) -> np.ndarray:
    """ids/mask (B,S) -> (B,H) unit vectors.  float64 by default (ground truth for the 1e-4 bar)."""
    P = {k: np.asarray(v, dtype=dtype) for k, v in state.items()}
    B, S = ids.shape
    H, nh, eps = cfg["hidden"], cfg["heads"], cfg["ln_eps"]
    dh = H // nh
    x = (P["embeddings.word_embeddings.weight"][ids] + P["embeddings.token_type_embeddings.weight"][0]
         + P["embeddings.position_embeddings.weight"][np.arange(S)][None])
    x = layernorm(x, P["embeddings.LayerNorm.weight"], P["embeddings.LayerNorm.bias"], eps)
    m = mask.astype(bool)
    bias = np.where(m, 0.0, np.finfo(np.float32).min)[:, None, None, :]  # HF extended attention mask
    for l in range(cfg["layers"]):
        p = f"encoder.layer.{l}."

        def lin(t, n):
            return t @ P[p + n + ".weight"].T + P[p + n + ".bias"]

        def heads(t):
            return t.reshape(B, S, nh, dh).transpose(0, 2, 1, 3)

        q, k, v = heads(lin(x, "attention.self.query")), heads(lin(x, "attention.self.key")), heads(lin(x, "attention.self.value"))
        s = q @ k.transpose(0, 1, 3, 2) / math.sqrt(dh) + bias
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        a = e / e.sum(-1, keepdims=True)
        ctx = (a @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
        x = layernorm(lin(ctx, "attention.output.dense") + x, P[p + "attention.output.LayerNorm.weight"],
                      P[p + "attention.output.LayerNorm.bias"], eps)
        h = gelu(lin(x, "intermediate.dense"))
        x = layernorm(lin(h, "output.dense") + x, P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], eps)
    w = mask.astype(dtype)[:, :, None]
    pooled = (x * w).sum(1) / np.maximum(w.sum(1), 1e-9)
    return pooled / np.maximum(np.linalg.norm(pooled, axis=1, keepdims=True), 1e-12)
