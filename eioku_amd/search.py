"""Semantic-search stage: exact kNN over segment embeddings (FAISS ``IndexFlatL2`` surface).

The reference never built this stage (``.kiro/specs/semantic-video-search/tasks.md:304-313`` is
unchecked; the design names a FAISS index over 384-d all-MiniLM-L6-v2 vectors,
``design.md:35-40,1105-1113``), so the interface mirrored here is the FAISS one BASELINE.json's
north_star names: ``add(x)``, ``search(q, k) -> (D, I)`` with squared-L2 distances ascending, int64
ids, ``-1`` padding, ``ntotal``, ``reset()``.

Multi-GPU: rows are sharded across ranks, every rank searches its shard with the
same replicated queries, ONE all-gather (RCCL over xGMI) moves ``nq*k*(4+8)`` bytes per rank and
every rank merges the ``world*k`` candidates locally (``eioku_topk_merge``).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._buffers import current_stream, on_device, ptr


class IndexFlatL2:
    """Exact squared-L2 index resident in HBM (d in {64,128,256,384,512}, k <= 32)."""

    def __init__(self, d: int):
        lib = _lib.load()
        _lib.init()
        self._lib = lib
        self.d = int(d)
        h = C.c_void_p()
        _lib.check(lib.eioku_index_flat_create(self.d, C.byref(h)), "eioku_index_flat_create")
        self._h = h
        self._attached = None  # keeps an attached tensor alive

    @property
    def ntotal(self) -> int:
        return int(self._lib.eioku_index_ntotal(self._h))

    def add(self, x) -> None:
        """Append vectors: float32 ``(n,d)`` numpy (staged over PCIe) or CUDA tensor (device copy)."""
        n, d = (int(s) for s in x.shape)
        if d != self.d:
            raise ValueError(f"expected dimension {self.d}, got {d}")
        if not on_device(x):
            x = np.ascontiguousarray(x, dtype=np.float32)
        _lib.check(self._lib.eioku_index_add(self._h, ptr(x), n, _lib.MEM_DEVICE if on_device(x) else _lib.MEM_HOST,
                                             current_stream(x)), "eioku_index_add")

    def attach(self, x_cuda) -> None:
        """Search an existing CUDA float32 ``(n,d)`` tensor in place (no copy; the tensor must outlive the index)."""
        n, d = (int(s) for s in x_cuda.shape)
        if d != self.d or not on_device(x_cuda):
            raise ValueError("attach() needs a CUDA tensor of shape (n, d)")
        _lib.check(self._lib.eioku_index_attach(self._h, ptr(x_cuda), n, current_stream(x_cuda)), "eioku_index_attach")
        self._attached = x_cuda

    def reset(self) -> None:
        _lib.check(self._lib.eioku_index_reset(self._h), "eioku_index_reset")
        self._attached = None

    def search(self, q, k: int):
        """``(D, I)``: float32 ``(nq,k)`` squared distances ascending, int64 ``(nq,k)`` ids (-1 = none).

        numpy queries -> numpy results (synchronous); CUDA queries -> CUDA results (asynchronous).
        """
        nq, d = (int(s) for s in q.shape)
        if d != self.d:
            raise ValueError(f"expected dimension {self.d}, got {d}")
        if on_device(q):
            import torch

            D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            mem = _lib.MEM_DEVICE
        else:
            q = np.ascontiguousarray(q, dtype=np.float32)
            D = np.empty((nq, k), dtype=np.float32)
            I = np.empty((nq, k), dtype=np.int64)
            mem = _lib.MEM_HOST
        _lib.check(self._lib.eioku_index_search(self._h, ptr(q), nq, int(k), ptr(D), ptr(I), mem, current_stream(q)),
                   "eioku_index_search")
        return D, I

    def search_after(self, q, k: int, after_D, after_I):
        """The next ``k`` results after a previous answer: rows with ``(distance, id) > (after_D[q], after_I[q])``
        in the result order (``after_*``: shape ``(nq,)``, same side of PCIe as ``q``)."""
        nq, d = (int(s) for s in q.shape)
        if d != self.d:
            raise ValueError(f"expected dimension {self.d}, got {d}")
        if on_device(q):
            import torch

            D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            after_D = after_D.to(torch.float32).contiguous()
            after_I = after_I.to(torch.int64).contiguous()
            mem = _lib.MEM_DEVICE
        else:
            q = np.ascontiguousarray(q, dtype=np.float32)
            after_D = np.ascontiguousarray(after_D, dtype=np.float32)
            after_I = np.ascontiguousarray(after_I, dtype=np.int64)
            D = np.empty((nq, k), dtype=np.float32)
            I = np.empty((nq, k), dtype=np.int64)
            mem = _lib.MEM_HOST
        _lib.check(self._lib.eioku_index_search_after(self._h, ptr(q), nq, int(k), ptr(after_D), ptr(after_I), ptr(D),
                                                      ptr(I), mem, current_stream(q)), "eioku_index_search_after")
        return D, I

    def search_many(self, q, k: int):
        """``search`` for any ``k``: rounds of at most 32 results chained with :meth:`search_after`."""
        if k <= 32:
            return self.search(q, k)
        parts_d, parts_i = [], []
        got = 0
        while got < k:
            kk = min(32, k - got)
            if not parts_d:
                D, I = self.search_after(q, kk, *self._before_everything(q))
            else:
                D, I = self.search_after(q, kk, parts_d[-1][:, -1], parts_i[-1][:, -1])
            parts_d.append(D)
            parts_i.append(I)
            got += kk
        if on_device(q):
            import torch

            return torch.cat(parts_d, 1), torch.cat(parts_i, 1)
        return np.concatenate(parts_d, 1), np.concatenate(parts_i, 1)

    @staticmethod
    def _before_everything(q):
        nq = int(q.shape[0])
        if on_device(q):
            import torch

            return (torch.full((nq,), -1.0, dtype=torch.float32, device=q.device),
                    torch.full((nq,), -1, dtype=torch.int64, device=q.device))
        return np.full((nq,), -1.0, np.float32), np.full((nq,), -1, np.int64)

    def set_param(self, name: str, value: int) -> None:
        """Knobs of the wide-search path (``eioku_index_set_param``): scan_mode, scan_cap, scan_min_rows, ..."""
        _lib.check(self._lib.eioku_index_set_param(self._h, name.encode(), int(value)), f"eioku_index_set_param({name})")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.eioku_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk(d_lists, i_lists, k: int):
    """HIP merge of per-shard results: CUDA ``(L,nq,k)`` float32 / int64 (global ids) -> ``(nq,k)``."""
    import torch

    if not on_device(d_lists):
        raise _lib.EiokuHipError("merge_topk runs on the GPU (eioku_topk_merge); got host tensors")
    lib = _lib.load()
    _lib.init()
    L, nq, kk = (int(s) for s in d_lists.shape)
    assert kk == k and tuple(i_lists.shape) == (L, nq, k)
    D = torch.empty((nq, k), dtype=torch.float32, device=d_lists.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=d_lists.device)
    _lib.check(lib.eioku_topk_merge(ptr(d_lists.contiguous()), ptr(i_lists.contiguous()), L, nq, k, ptr(D), ptr(I),
                                    current_stream(d_lists)), "eioku_topk_merge")
    return D, I


def shard_bounds(n_total: int, world: int, rank: int) -> tuple[int, int]:
    """Rows ``[lo, hi)`` of rank ``rank``: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedFlatL2:
    """Row-sharded exact index: local shard search + one all-gather + local merge.

    ``local`` is this rank's shard (anything with ``search(q, k) -> (D, I)`` of torch tensors and local
    ids); ``id_base`` the global id of its first row.  ``group`` is a ``torch.distributed`` group whose
    backend is RCCL ("nccl") on GPUs; ``merge`` defaults to the HIP merge kernel.
    """

    def __init__(self, local, id_base: int, group=None, merge=merge_topk):
        self.local, self.id_base, self.group, self.merge = local, int(id_base), group, merge

    def search(self, q, k: int):
        import torch
        import torch.distributed as dist

        D, I = self.local.search(q, k)
        I = torch.where(I >= 0, I + self.id_base, I)
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return D, I
        world = dist.get_world_size(self.group)
        # one collective of 12 bytes per (distance, id): an int32 payload [nq][3k] = the float bits | the id's two halves
        payload = torch.cat([D.contiguous().view(torch.int32), I.contiguous().view(torch.int32)], dim=1).contiguous()
        nq = payload.shape[0]
        flat = torch.empty((world * nq, 3 * k), dtype=torch.int32, device=payload.device)
        dist.all_gather_into_tensor(flat, payload, group=self.group)  # rank-major concatenation
        gathered = flat.view(world, nq, 3 * k)
        dl = gathered[:, :, :k].contiguous().view(torch.float32)
        il = gathered[:, :, k:].contiguous().view(torch.int64)
        return self.merge(dl, il, k)


class RcclComm:
    """The library's own RCCL communicator (``eioku_comm_*``): for hosts that hold no ``torch.distributed`` process
    group.  Rank 0 calls :meth:`unique_id` and ships the 128 bytes to its peers by whatever channel the service has;
    every rank (one process per GPU) then constructs ``RcclComm(id, rank, world)`` -- a collective call."""

    ID_BYTES = 128

    def __init__(self, unique_id: bytes, rank: int, world: int):
        import ctypes as C

        if len(unique_id) != self.ID_BYTES:
            raise ValueError(f"unique id must be {self.ID_BYTES} bytes, got {len(unique_id)}")
        self._lib = _lib.load()
        _lib.init()
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), self.ID_BYTES)
        _lib.check(self._lib.eioku_comm_create(C.cast(buf, C.c_void_p), int(rank), int(world), C.byref(h)), "eioku_comm_create")
        self._h, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        lib = _lib.load()
        buf = C.create_string_buffer(RcclComm.ID_BYTES)
        _lib.check(lib.eioku_comm_unique_id(C.cast(buf, C.c_void_p)), "eioku_comm_unique_id")
        return buf.raw

    def close(self):
        if getattr(self, "_h", None):
            self._lib.eioku_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CommShardedFlatL2:
    """:class:`ShardedFlatL2` with the collective inside the C ABI (``eioku_index_search_sharded``): local shard
    search, one RCCL all-gather of the packed answers, local merge.  ``local`` is this rank's :class:`IndexFlatL2`."""

    def __init__(self, local: "IndexFlatL2", id_base: int, comm: RcclComm):
        self.local, self.id_base, self.comm = local, int(id_base), comm

    def search(self, q, k: int):
        import torch

        if not on_device(q):
            raise _lib.EiokuHipError("eioku_index_search_sharded takes device queries")
        nq, d = (int(s) for s in q.shape)
        if d != self.local.d:
            raise ValueError(f"expected dimension {self.local.d}, got {d}")
        q = q.contiguous()
        D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        _lib.check(self.local._lib.eioku_index_search_sharded(self.local._h, self.comm._h, self.id_base, ptr(q), nq, int(k),
                                                              ptr(D), ptr(I), current_stream(q)), "eioku_index_search_sharded")
        return D, I
