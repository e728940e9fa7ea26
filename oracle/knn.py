"""kNN stage oracle (row a6 of SURVEY.md §8): FAISS ``IndexFlatL2`` semantics, brute force.

Test infrastructure (see ``oracle/__init__.py``).  **Parity unpinned**: faiss is neither a
dependency of the reference (intent only, ``.kiro/specs/semantic-video-search/tasks.md:304-313``)
nor installed here; the published semantics are restated [PUBLIC-LIB]: squared L2, ascending,
int64 labels, ``-1`` / +inf-like padding when fewer than k vectors exist.  Distances are computed in
float64 as ground truth (the 1e-4 relative bar of BASELINE.json is checked against these); ties are
ordered by the smaller id (FAISS leaves them unspecified).
"""
from __future__ import annotations

import numpy as np


def search(xb: np.ndarray, xq: np.ndarray, k: int, chunk: int = 65536):
    """(D float64 (nq,k), I int64 (nq,k)) by exhaustive float64 ``sum((q-x)^2)``."""
    xb = np.asarray(xb, dtype=np.float64)
    xq = np.asarray(xq, dtype=np.float64)
    n, nq = xb.shape[0], xq.shape[0]
    D = np.full((nq, k), np.inf)
    I = np.full((nq, k), -1, dtype=np.int64)
    qn = (xq * xq).sum(1)
    for lo in range(0, n, chunk):
        blk = xb[lo:lo + chunk]
        d = qn[:, None] + (blk * blk).sum(1)[None, :] - 2.0 * (xq @ blk.T)
        # refine the candidates exactly: the decomposition is only used to shortlist
        kk = min(k + 8, blk.shape[0])
        part = np.argpartition(d, kk - 1, axis=1)[:, :kk]
        for qi in range(nq):
            ids = part[qi]
            ex = ((xq[qi][None, :] - blk[ids]) ** 2).sum(1)
            allv = np.concatenate([D[qi], ex])
            alli = np.concatenate([I[qi], ids + lo])
            order = np.lexsort((alli, allv))[:k]
            D[qi], I[qi] = allv[order], alli[order]
    I[~np.isfinite(D)] = -1
    return D, I


def merge(d_lists: np.ndarray, i_lists: np.ndarray, k: int):
    """Reference merge of per-shard lists (L,nq,k) -> (nq,k), ties by smaller id, -1 ignored."""
    L, nq, _ = d_lists.shape
    D = np.empty((nq, k), dtype=d_lists.dtype)
    I = np.empty((nq, k), dtype=np.int64)
    for q in range(nq):
        v = d_lists[:, q, :].reshape(-1)
        i = i_lists[:, q, :].reshape(-1)
        ok = i >= 0
        order = np.lexsort((i[ok], v[ok]))[:k]
        D[q, :len(order)], I[q, :len(order)] = v[ok][order], i[ok][order]
        D[q, len(order):], I[q, len(order):] = np.finfo(np.float32).max, -1
    return D, I
