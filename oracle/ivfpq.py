"""IVF-PQ oracle (row a6, second half): FAISS ``IndexIVFPQ`` algorithm in numpy.

Test infrastructure (see ``oracle/__init__.py``).  **Parity unpinned** (faiss is not installed and
is not a dependency of the reference).  The published algorithm is restated [PUBLIC-LIB] with the
same deterministic choices as ``eioku_amd/ivfpq.py`` (initialisation from ``default_rng``
permutations, 25 Lloyd iterations, empty clusters keep their centroid, first minimum wins), so the
HIP path and this file can be compared code for code.
"""
from __future__ import annotations

import numpy as np

NITER = 25


def _sample(n, want, seed):
    perm = np.random.default_rng(seed).permutation(n)
    return np.sort(perm[:want]) if want < n else np.arange(n)


def assign(x, cent):
    """nearest centroid (squared L2, first minimum wins) in float64"""
    x = x.astype(np.float64)
    c = cent.astype(np.float64)
    d = (x * x).sum(1)[:, None] + (c * c).sum(1)[None, :] - 2.0 * x @ c.T
    return d.argmin(1)


def kmeans(x, k, seed, niter=NITER):
    init = np.sort(np.random.default_rng(seed).permutation(len(x))[:k])
    cent = x[init].astype(np.float32).copy()
    for _ in range(niter):
        a = assign(x, cent)
        for c in range(k):
            sel = a == c
            if sel.any():
                cent[c] = x[sel].astype(np.float64).mean(0).astype(np.float32)
    return cent


class IVFPQ:
    def __init__(self, d, nlist, m, seed=1234):
        self.d, self.nlist, self.m, self.dsub, self.seed = d, nlist, m, d // m, seed
        self.nprobe = 1

    def train(self, x):
        n = len(x)
        xs = x[_sample(n, 256 * self.nlist, self.seed)]
        self.coarse = kmeans(xs, self.nlist, self.seed + 1)
        xp = x[_sample(n, 256 * 256, self.seed + 2)]
        resid = (xp - self.coarse[assign(xp, self.coarse)]).astype(np.float32)
        init = np.sort(np.random.default_rng(self.seed + 3).permutation(len(resid))[:256])
        self.pq = np.ascontiguousarray(resid[init].reshape(256, self.m, self.dsub).transpose(1, 0, 2)).copy()
        for _ in range(NITER):
            for j in range(self.m):
                sub = resid[:, j * self.dsub:(j + 1) * self.dsub]
                a = self._sub_assign(sub, self.pq[j])
                for c in range(256):
                    sel = a == c
                    if sel.any():
                        self.pq[j, c] = sub[sel].astype(np.float64).mean(0).astype(np.float32)

    @staticmethod
    def _sub_assign(sub, cb):
        d = ((sub[:, None, :].astype(np.float32) - cb[None, :, :].astype(np.float32)) ** 2)
        s = np.zeros(d.shape[:2], dtype=np.float32)
        for t in range(d.shape[2]):  # same accumulation order as the kernel
            s = (s + d[:, :, t]).astype(np.float32)
        return s.argmin(1)

    def encode(self, x):
        lst = assign(x, self.coarse)
        resid = (x - self.coarse[lst]).astype(np.float32)
        codes = np.stack([self._sub_assign(resid[:, j * self.dsub:(j + 1) * self.dsub], self.pq[j])
                          for j in range(self.m)], 1).astype(np.uint8)
        return lst, codes

    def add(self, x):
        self.lst, self.codes = self.encode(x)

    def search(self, q, k):
        nq = len(q)
        qd = ((q.astype(np.float64)[:, None, :] - self.coarse.astype(np.float64)[None]) ** 2).sum(-1)
        probes = np.argsort(qd, axis=1, kind="stable")[:, :self.nprobe]
        D = np.full((nq, k), np.finfo(np.float32).max, dtype=np.float32)
        I = np.full((nq, k), -1, dtype=np.int64)
        for qi in range(nq):
            cand_d, cand_i = [], []
            for l in probes[qi]:
                ids = np.nonzero(self.lst == l)[0]
                if not len(ids):
                    continue
                r = (q[qi] - self.coarse[l]).astype(np.float32)
                lut = np.zeros((self.m, 256), dtype=np.float32)
                for j in range(self.m):
                    df = (r[j * self.dsub:(j + 1) * self.dsub][None, :] - self.pq[j]).astype(np.float32) ** 2
                    s = np.zeros(256, dtype=np.float32)
                    for t in range(self.dsub):
                        s = (s + df[:, t]).astype(np.float32)
                    lut[j] = s
                dist = np.zeros(len(ids), dtype=np.float32)
                for j in range(self.m):
                    dist = (dist + lut[j, self.codes[ids, j]]).astype(np.float32)
                cand_d.append(dist)
                cand_i.append(ids)
            if cand_d:
                cd, ci = np.concatenate(cand_d), np.concatenate(cand_i)
                o = np.lexsort((ci, cd))[:k]
                D[qi, :len(o)], I[qi, :len(o)] = cd[o], ci[o]
        return D, I
