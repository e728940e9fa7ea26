"""CPU, world size 2 over gloo: the sharded IVF-PQ build of BASELINE cfg5.

`IndexIVFPQ.train(x, group=...)` is the PRODUCT's training loop; on the GPU its data-parallel steps are HIP kernels
(`HipTrainOps`), here a numpy stand-in with the same integer arithmetic is injected through the `train_ops` seam so
that the collective pattern runs on CPU ranks: one all-reduce of nlist x (d + 1) int64 per coarse k-means iteration,
one of m x 256 x (dsub + 1) per PQ iteration, rank 0's seeds broadcast.  Because the sums are integers (2^-32 fixed
point) the reduction is exact and order independent: every rank ends with the same quantisers, and they equal -
bit for bit - a single-process Lloyd run over the union of the rows from the same seeds.
"""
import os
import socket

import numpy as np
import pytest

from eioku_amd import ivfpq, search

FIX = 4294967296.0


class NumpyTrainOps:
    """`HipTrainOps` on torch-CPU tensors: float64 argmin assignment, rint(x * 2^32) integer sums."""

    def to_device(self, a):
        import torch

        return torch.as_tensor(np.asarray(a), dtype=torch.float32).contiguous()

    def take(self, x, idx):
        import torch

        return x[torch.from_numpy(np.asarray(idx))].contiguous()

    def assign(self, x, cent):
        import torch

        a, c = x.double().numpy(), cent.double().numpy()
        d2 = (a * a).sum(1)[:, None] + (c * c).sum(1)[None, :] - 2.0 * a @ c.T
        return torch.from_numpy(d2.argmin(1).astype(np.int64))

    def accumulate(self, x, assign, k):
        import torch

        a = assign.numpy()
        q = np.rint(x.double().numpy() * FIX).astype(np.int64)
        out = np.zeros((k, x.shape[1] + 1), np.int64)
        np.add.at(out[:, :-1], a, q)
        np.add.at(out[:, -1], a, 1)
        return torch.from_numpy(out)

    def finalize(self, packed, cent):
        import torch

        p = packed.numpy()
        out = cent.clone().numpy()
        have = p[:, -1] > 0
        out[have] = ((p[have, :-1].astype(np.float64) / FIX) / p[have, -1:].astype(np.float64)).astype(np.float32)
        return torch.from_numpy(out)

    def residuals(self, x, coarse, lst, m):
        return (x - coarse[lst]).contiguous()

    def pq_codes(self, resid, pq):
        import torch

        m, _, dsub = pq.shape
        r = resid.double().numpy().reshape(len(resid), m, dsub)
        c = pq.double().numpy()
        d2 = ((r[:, :, None, :] - c[None]) ** 2).sum(-1)
        return torch.from_numpy(d2.argmin(-1).astype(np.uint8))

    def column(self, x, lo, hi):
        return x[:, lo:hi].contiguous()

    def codes_column(self, codes, j):
        import torch

        return codes[:, j].to(torch.int64).contiguous()


def _rows(seed, n, d):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((12, d)).astype(np.float32)
    return (c[rng.integers(0, 12, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)


D, NLIST, M = 16, 8, 4
SHARDS = [_rows(1, 700, D), _rows(2, 900, D)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ix = ivfpq.IndexIVFPQ(D, NLIST, M, train_ops=NumpyTrainOps(), seed=5)
        ix.train(SHARDS[rank], group=dist.group.WORLD)
        out[rank] = (ix.coarse.numpy().tobytes(), ix.pq.numpy().tobytes(), ix.allreduce_calls)
    finally:
        dist.destroy_process_group()


def _lloyd_union(rows, cent, k):
    ops = NumpyTrainOps()
    import torch

    x = torch.from_numpy(rows)
    for _ in range(ivfpq.NITER):
        cent = ops.finalize(ops.accumulate(x, ops.assign(x, cent), k), cent)
    return cent


def test_sharded_train_gloo_world2_equals_single_process_union():
    import torch
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        out = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        res = dict(out)
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]          # identical quantisers on both ranks
    assert res[0][2] == res[1][2] == 2 * ivfpq.NITER                  # one collective per k-means iteration (coarse, PQ)
    coarse = np.frombuffer(res[0][0], np.float32).reshape(NLIST, D)
    # single-process Lloyd over the union of the shards from rank 0's seeds (the documented rule): bit-identical
    seeds = np.sort(np.random.default_rng(5 + 1).permutation(len(SHARDS[0]))[:NLIST])
    want = _lloyd_union(np.concatenate(SHARDS), torch.from_numpy(SHARDS[0][seeds].copy()), NLIST)
    assert np.array_equal(coarse, want.numpy())
    assert len(np.unique(coarse, axis=0)) == NLIST


def test_group_none_is_the_single_gpu_build_and_needs_enough_rows():
    ix = ivfpq.IndexIVFPQ(D, NLIST, M, train_ops=NumpyTrainOps(), seed=5)
    ix.train(SHARDS[0])
    assert ix.is_trained and ix.allreduce_calls == 0 and tuple(ix.pq.shape) == (M, 256, D // M)
    with pytest.raises(ValueError, match="nlist"):
        ivfpq.IndexIVFPQ(D, 64, M, train_ops=NumpyTrainOps()).train(SHARDS[0][:10])


def test_shard_bounds_cover_cfg5_rows_evenly():
    """100 M rows over 8 GPUs: contiguous, disjoint, 12.5 M each (BASELINE cfg5)."""
    b = [search.shard_bounds(100_000_000, 8, r) for r in range(8)]
    assert b[0][0] == 0 and b[-1][1] == 100_000_000 and all(hi - lo == 12_500_000 for lo, hi in b)
    assert all(b[i][1] == b[i + 1][0] for i in range(7))
