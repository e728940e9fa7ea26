"""Approximate kNN: ``IndexIVFPQ`` (FAISS semantics: L2, residual encoding, 8-bit codes) on the HIP kernels.

BASELINE.json cfg5 (IVF-PQ over 100M x 384, nlist 4096).  The reference itself holds intent only for
vector search (``.kiro/specs/semantic-video-search/design.md:35-40``).  Host orchestration - the
k-means / PQ training loops and the inverted-list bookkeeping - is Python here, as the north star
asks; every data-parallel step is a kernel of ``libeioku_hip``: coarse assignment and probe
selection = K9 (``k=1`` / ``k=nprobe``), centroid update = fixed-point atomics (bit-reproducible),
PQ assignment / encoding, counting-sort into lists, ADC scan with LUT in LDS, top-k merge.

Deviations from FAISS defaults, all deterministic: k-means initialises from
``default_rng(seed).permutation(n)[:k]``, an empty cluster keeps its previous centroid (FAISS splits
the largest), 25 iterations, at most 256 training points per centroid.

Multi-GPU build (BASELINE cfg5: 100M x 384 over 8 GPUs): ``train(x, group=...)`` trains the
quantisers on the union of every rank's rows with one integer all-reduce per k-means iteration; ``add`` /
``search`` stay local to the rank's row shard, and ``search.ShardedFlatL2(local=this index, id_base=...)`` merges
the per-rank results with the single all-gather of the flat index.
"""

from __future__ import annotations

import numpy as np

from . import _lib
from ._buffers import current_stream, ptr
from .search import IndexFlatL2

NITER = 25
MAX_POINTS_PER_CENTROID = 256


def _sample(n: int, want: int, seed: int) -> np.ndarray:
    perm = np.random.default_rng(seed).permutation(n)
    return np.sort(perm[:want]) if want < n else np.arange(n)


class HipTrainOps:
    """The data-parallel steps of training on this rank's rows, on the HIP kernels.  (A seam: the world-size-2 gloo
    test drives :meth:`IndexIVFPQ.train` on CPU ranks with a numpy stand-in of the same integer arithmetic.)"""

    def __init__(self, device):
        self._lib = _lib.load()
        self.device = device

    def to_device(self, a):
        import torch

        return torch.as_tensor(a, dtype=torch.float32).to(self.device).contiguous()

    def take(self, x, idx):
        import torch

        return x[torch.from_numpy(np.asarray(idx)).to(x.device)].contiguous()

    def assign(self, x, cent):
        """nearest centroid of every row (int64): K9 with k = 1"""
        flat = IndexFlatL2(int(cent.shape[1]))
        flat.attach(cent)
        _, a = flat.search(x, 1)
        a = a.reshape(-1).contiguous()
        flat.close()
        return a

    def accumulate(self, x, assign, k):
        """int64 (k, d + 1): per-centroid sums of the assigned rows in 2^-32 fixed point, and the count in column d"""
        import torch

        n, d = (int(v) for v in x.shape)
        sums = torch.zeros((k, d), dtype=torch.int64, device=x.device)
        counts = torch.zeros((k,), dtype=torch.int32, device=x.device)
        _lib.check(self._lib.eioku_kmeans_accumulate(ptr(x), n, d, ptr(assign), k, ptr(sums), ptr(counts), current_stream(x)),
                   "eioku_kmeans_accumulate")
        return torch.cat([sums, counts.to(torch.int64)[:, None]], 1).contiguous()

    def finalize(self, packed, cent):
        """centroids <- sums / counts (a centroid without rows keeps its value); in place"""
        import torch

        k, d = (int(v) for v in cent.shape)
        sums = packed[:, :d].contiguous()
        counts = packed[:, d].to(torch.int32).contiguous()
        _lib.check(self._lib.eioku_kmeans_finalize(ptr(sums), ptr(counts), k, d, ptr(cent), current_stream(cent)),
                   "eioku_kmeans_finalize")
        return cent

    def residuals(self, x, coarse, lst, m):
        import torch

        resid = torch.empty_like(x)
        dummy = torch.zeros((m, 256, x.shape[1] // m), dtype=torch.float32, device=x.device)
        _lib.check(self._lib.eioku_pq_assign(ptr(x), x.shape[0], x.shape[1], m, ptr(coarse), ptr(lst), ptr(dummy), None, ptr(resid),
                                             current_stream(x)), "eioku_pq_assign(residual)")
        return resid

    def pq_codes(self, resid, pq):
        import torch

        n, d = (int(v) for v in resid.shape)
        codes = torch.empty((n, pq.shape[0]), dtype=torch.uint8, device=resid.device)
        _lib.check(self._lib.eioku_pq_assign(ptr(resid), n, d, int(pq.shape[0]), None, None, ptr(pq), ptr(codes), None,
                                             current_stream(resid)), "eioku_pq_assign(train)")
        return codes

    def column(self, x, lo, hi):
        return x[:, lo:hi].contiguous()

    def codes_column(self, codes, j):
        import torch

        return codes[:, j].to(torch.int64).contiguous()


class IndexIVFPQ:
    def __init__(self, d: int, nlist: int, m: int, nbits: int = 8, device=None, seed: int = 1234, train_ops=None):
        import torch

        if nbits != 8:
            raise ValueError("only 8-bit PQ codes are supported")
        if d % m or d // m not in (4, 8, 16):
            raise ValueError("d/m must be 4, 8 or 16")
        self.d, self.nlist, self.m, self.dsub, self.seed = d, nlist, m, d // m, seed
        self.nprobe = 1
        if train_ops is None:
            self._lib = _lib.load()
            _lib.init()
            self.device = device or torch.device("cuda", torch.cuda.current_device())
            train_ops = HipTrainOps(self.device)
        else:
            self._lib = None
            self.device = device
        self._ops = train_ops
        self.is_trained = False
        self.coarse = None          # (nlist, d) float32 cuda
        self.pq = None              # (m, 256, dsub) float32 cuda
        self._quantizer = None      # IndexFlatL2 over the coarse centroids
        self._pending = []          # (list int64, codes uint8, id_base) per add() batch
        self.ntotal = 0
        self._packed = None
        self.use_precomputed_table = True   # FAISS' IndexIVFPQ.use_precomputed_table; False = tables from the codebook per (query, list)
        self._list_tables = None
        # "lists": the list-major scan (codes cross HBM once per search, MFMA filter + exact re-rank; same (D, I) bits as
        # "queries"); "queries": one workgroup per (query, probe).  "lists" needs d/m = 8, d in {64, 128, 256, 384} and the
        # precomputed tables; other geometries always take "queries".
        self.scan_mode = "lists"
        self.cand_cap = 8192                # per-query candidate capacity of the list-major filter (overflow -> query-major redo)
        self._aux = None                    # (pqh, hx, pmax2) of the current pack
        self._ws = None
        self.last_stats = None              # device int32[6] after a list-major search: overflow bits, work items, largest candidate list, candidates in all, its query, its size
        self.allreduce_calls = 0

    # ---- training ------------------------------------------------------------------------------
    def _allreduce(self, packed, group):
        """Sum the integer (sums | counts) table over the ranks of ``group`` (RCCL on GPUs): exact and order
        independent, so every rank - and a single-GPU run over the same rows - finalises the same centroids."""
        if group is None:
            return packed
        import torch.distributed as dist

        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        self.allreduce_calls += 1
        return packed

    def _bcast(self, t, group):
        if group is not None:
            import torch.distributed as dist

            dist.broadcast(t, src=dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0, group=group)
        return t

    def _kmeans(self, x, k: int, seed: int, group=None):
        """Lloyd iterations on this rank's (n,d) rows -> (k,d) centroids shared by every rank of ``group``:
        assignment on the MFMA flat kernel, one all-reduce of k x (d + 1) int64 per iteration."""
        ops = self._ops
        n = int(x.shape[0])
        init = np.sort(np.random.default_rng(seed).permutation(n)[:k])
        cent = self._bcast(ops.take(x, init).clone(), group)  # rank 0's picks seed every rank
        for _ in range(NITER):
            packed = self._allreduce(ops.accumulate(x, ops.assign(x, cent), k), group)
            cent = ops.finalize(packed, cent)
        return cent

    def train(self, x, group=None) -> None:
        """``x``: float32 (n,d) rows of THIS rank (numpy array or CUDA tensor).  ``group``: a ``torch.distributed``
        group (RCCL on GPUs) whose ranks each hold a shard of the training rows: the quantisers are trained on the
        union - per k-means iteration ONE all-reduce of ``nlist x (d + 1)`` (coarse) / ``m x 256 x (dsub + 1)`` (PQ)
        integers - and come out identical on every rank."""
        ops = self._ops
        x = ops.to_device(x)
        n = int(x.shape[0])
        if n < max(self.nlist, 256):
            raise ValueError(f"need at least max(nlist={self.nlist}, 256) training vectors on every rank (the k-means seeds), got {n}")
        xs = ops.take(x, _sample(n, MAX_POINTS_PER_CENTROID * self.nlist, self.seed))
        self.coarse = self._kmeans(xs, self.nlist, self.seed + 1, group)
        # PQ on residuals of (at most 65536 per rank) training vectors
        xp = ops.take(x, _sample(n, MAX_POINTS_PER_CENTROID * 256, self.seed + 2))
        lst = ops.assign(xp, self.coarse)
        resid = ops.residuals(xp, self.coarse, lst, self.m)
        npq = int(resid.shape[0])
        init = np.sort(np.random.default_rng(self.seed + 3).permutation(npq)[:256])
        pq = self._bcast(ops.take(resid, init).reshape(256, self.m, self.dsub).permute(1, 0, 2).contiguous(), group)
        import torch

        for _ in range(NITER):
            codes = ops.pq_codes(resid, pq)
            packed = torch.stack([ops.accumulate(ops.column(resid, j * self.dsub, (j + 1) * self.dsub),
                                                 ops.codes_column(codes, j), 256) for j in range(self.m)])
            packed = self._allreduce(packed.contiguous(), group)  # (m, 256, dsub + 1): one collective for all sub-quantisers
            pq = torch.stack([ops.finalize(packed[j], pq[j].contiguous()) for j in range(self.m)]).contiguous()
        self.pq = pq
        self._list_tables = None
        self._aux = None
        if self._lib is not None:
            if self._quantizer is not None:
                self._quantizer.close()
            self._quantizer = IndexFlatL2(self.d)
            self._quantizer.attach(self.coarse)
        self.is_trained = True

    def set_codebooks(self, coarse, pq) -> None:
        """Install trained quantisers (e.g. broadcast from rank 0): coarse (nlist,d), pq (m,256,dsub)."""
        import torch

        self.coarse = torch.as_tensor(coarse, dtype=torch.float32).to(self.device).contiguous()
        self.pq = torch.as_tensor(pq, dtype=torch.float32).to(self.device).contiguous()
        self._list_tables = None
        self._aux = None
        if self._quantizer is not None:
            self._quantizer.close()
        self._quantizer = IndexFlatL2(self.d)
        self._quantizer.attach(self.coarse)
        self.is_trained = True

    # ---- add / search --------------------------------------------------------------------------------
    def add(self, x) -> None:
        import torch

        if not self.is_trained:
            raise RuntimeError("train() or set_codebooks() first")
        x = torch.as_tensor(x, dtype=torch.float32).to(self.device).contiguous()
        n = x.shape[0]
        if n == 0:
            return
        _, lst = self._quantizer.search(x, 1)
        lst = lst.reshape(-1).contiguous()
        codes = torch.empty((n, self.m), dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.eioku_pq_assign(ptr(x), n, self.d, self.m, ptr(self.coarse), ptr(lst), ptr(self.pq), ptr(codes),
                                             None, current_stream(x)), "eioku_pq_assign(encode)")
        self._pending.append((lst, codes, self.ntotal))
        self.ntotal += n
        self._packed = None
        self._aux = None

    def _pack(self):
        """Counting sort of everything added so far into contiguous inverted lists."""
        import torch

        if self._packed is not None:
            return self._packed
        counts = torch.zeros((self.nlist,), dtype=torch.int32, device=self.device)
        tmp = torch.empty_like(counts)
        for lst, _, _ in self._pending:
            _lib.check(self._lib.eioku_ivf_histogram(ptr(lst), lst.numel(), self.nlist, ptr(tmp), current_stream(lst)),
                       "eioku_ivf_histogram")
            counts += tmp
        offsets = (torch.cumsum(counts, 0) - counts).to(torch.int32).contiguous()
        cursor = offsets.clone()
        list_codes = torch.empty((max(self.ntotal, 1), self.m), dtype=torch.uint8, device=self.device)
        list_ids = torch.empty((max(self.ntotal, 1),), dtype=torch.int64, device=self.device)
        for lst, codes, base in self._pending:
            _lib.check(self._lib.eioku_ivf_scatter(ptr(lst), lst.numel(), self.m, ptr(codes), base, ptr(cursor),
                                                   ptr(list_codes), ptr(list_ids), current_stream(lst)), "eioku_ivf_scatter")
        self._packed = (offsets, counts.contiguous(), list_codes, list_ids)
        return self._packed

    def search(self, q, k: int):
        """(D, I) CUDA tensors: approximate squared L2 (ADC) ascending, int64 ids (-1 = fewer than k found)."""
        import torch

        q = torch.as_tensor(q, dtype=torch.float32).to(self.device).contiguous()
        nq = q.shape[0]
        nprobe = min(self.nprobe, self.nlist)
        offsets, sizes, list_codes, list_ids = self._pack()
        _, probes = self._quantizer.search_many(q, nprobe)  # nprobe > 32: chained rounds of 32 (search_after)
        probes = probes.contiguous()
        K = 16 if k <= 16 else 32
        lists = (self.scan_mode == "lists" and self.use_precomputed_table and self.dsub == 8
                 and self.d in (64, 128, 256, 384) and self.ntotal > 0)
        if not lists:
            pd = torch.empty((nprobe, nq, K), dtype=torch.float32, device=self.device)
            pi = torch.empty((nprobe, nq, K), dtype=torch.int64, device=self.device)
        if self.use_precomputed_table:
            # FAISS' decomposition: the per-list half of every look-up table is part of the index (nlist x m x 1 KB, built
            # on the first search after the codebooks change), the per-query half is built once per search
            if self._list_tables is None:
                self._list_tables = torch.empty((self.nlist, self.m, 256), dtype=torch.float32, device=self.device)
                _lib.check(self._lib.eioku_ivfpq_tables(ptr(self.coarse), self.nlist, self.d, self.m, ptr(self.pq), 1.0, 2.0,
                                                        ptr(self._list_tables), current_stream(q)), "eioku_ivfpq_tables")
            qt = torch.empty((nq, self.m, 256), dtype=torch.float32, device=self.device)
            _lib.check(self._lib.eioku_ivfpq_tables(ptr(q), nq, self.d, self.m, ptr(self.pq), 0.0, -2.0, ptr(qt),
                                                    current_stream(q)), "eioku_ivfpq_tables")
            if lists:
                return self._search_lists(q, k, probes, nprobe, qt, offsets, sizes, list_codes, list_ids)
            _lib.check(self._lib.eioku_ivfpq_scan_tables(ptr(q), nq, self.d, self.m, ptr(probes), nprobe, ptr(self.coarse),
                                                         ptr(self.pq), ptr(offsets), ptr(sizes), ptr(list_codes), ptr(list_ids),
                                                         ptr(self._list_tables), ptr(qt), k, ptr(pd), ptr(pi),
                                                         current_stream(q)), "eioku_ivfpq_scan_tables")
        else:
            _lib.check(self._lib.eioku_ivfpq_scan(ptr(q), nq, self.d, self.m, ptr(probes), nprobe, ptr(self.coarse), ptr(self.pq),
                                                  ptr(offsets), ptr(sizes), ptr(list_codes), ptr(list_ids), k, ptr(pd), ptr(pi),
                                                  current_stream(q)), "eioku_ivfpq_scan")
        D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        _lib.check(self._lib.eioku_topk_merge_ex(ptr(pd), ptr(pi), nprobe, nq, K, k, ptr(D), ptr(I), current_stream(q)),
                   "eioku_topk_merge_ex")
        return D, I

    def _search_lists(self, q, k, probes, nprobe, qt, offsets, sizes, list_codes, list_ids):
        """The list-major scan (``eioku_ivfpq_search_lists``): one C call enqueues the whole search."""
        import torch

        nq = int(q.shape[0])
        if self._aux is None:
            pqh = torch.empty((self.m * 256, 4), dtype=torch.int32, device=self.device)
            hx = torch.empty((max(self.ntotal, 1),), dtype=torch.float32, device=self.device)
            pmax2 = torch.empty((self.nlist,), dtype=torch.float32, device=self.device)
            _lib.check(self._lib.eioku_ivfpq_lists_aux(ptr(list_codes), ptr(offsets), ptr(sizes), self.nlist, self.d, self.m,
                                                       ptr(self._list_tables), ptr(self.pq), ptr(pqh), ptr(hx), ptr(pmax2),
                                                       current_stream(q)), "eioku_ivfpq_lists_aux")
            self._aux = (pqh, hx, pmax2)
        pqh, hx, pmax2 = self._aux
        need = int(self._lib.eioku_ivfpq_lists_workspace(nq, self.d, nprobe, self.nlist, self.ntotal, k, self.cand_cap))
        if need < 0:
            raise _lib.EiokuHipError("eioku_ivfpq_lists_workspace: bad argument")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        stats = torch.zeros((6,), dtype=torch.int32, device=self.device)
        _lib.check(self._lib.eioku_ivfpq_search_lists(ptr(q), nq, self.d, self.m, ptr(probes), nprobe, self.nlist, self.ntotal, ptr(self.coarse),
                                                      ptr(self.pq), ptr(offsets), ptr(sizes), ptr(list_codes), ptr(list_ids),
                                                      ptr(self._list_tables), ptr(qt), ptr(pqh), ptr(hx), ptr(pmax2), k,
                                                      self.cand_cap, ptr(self._ws), self._ws.numel(), ptr(D), ptr(I), ptr(stats),
                                                      current_stream(q)), "eioku_ivfpq_search_lists")
        self.last_stats = stats
        return D, I
