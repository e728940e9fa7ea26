"""CPU tests: kNN oracle vs its golden vector; shard bounds; N>1 search path on gloo (world_size 2)."""
import os
import socket

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import knn as oknn, prng
from eioku_amd import search


def unit_rows(seed, n, d):
    x = prng.approx_normal_f32(seed, n * d).reshape(n, d)
    return (x / np.sqrt((x.astype(np.float64) ** 2).sum(1, keepdims=True))).astype(np.float32)


def test_oracle_golden_and_basic_properties():
    g = np.load(GOLDEN / "flatl2_4096x384.npz")
    xb, xq = unit_rows(21, 4096, 384), unit_rows(22, 16, 384)
    D, I = oknn.search(xb, xq, 10)
    assert np.array_equal(I, g["I"]) and np.allclose(D, g["D"], rtol=1e-12)
    assert np.all(np.diff(D, axis=1) >= 0)
    # unit vectors: d2 = 2 - 2cos in [0,4]
    assert D.min() >= 0 and D.max() <= 4
    # fewer vectors than k -> -1 padding (FAISS semantics)
    D2, I2 = oknn.search(xb[:3], xq[:2], 5)
    assert np.all(I2[:, 3:] == -1) and np.all(I2[:, :3] >= 0)


def test_shard_bounds_partition_rows():
    for n in (0, 1, 7, 1000, 10_000_000):
        for w in (1, 2, 3, 8):
            b = [search.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_oracle_merge_equals_whole_search():
    xb, xq = unit_rows(1, 999, 64), unit_rows(2, 7, 64)
    Dw, Iw = oknn.search(xb, xq, 6)
    dl, il = [], []
    for r in range(4):
        lo, hi = search.shard_bounds(999, 4, r)
        D, I = oknn.search(xb[lo:hi], xq, 6)
        dl.append(D)
        il.append(np.where(I >= 0, I + lo, I))
    D, I = oknn.merge(np.stack(dl), np.stack(il), 6)
    assert np.array_equal(I, Iw) and np.allclose(D, Dw)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        xb, xq = unit_rows(3, 1001, 64), unit_rows(4, 9, 64)
        lo, hi = search.shard_bounds(len(xb), world, rank)

        class CpuShard:  # stands in for the HBM shard: the collective + id offsets are what is under test
            def search(self, q, k):
                D, I = oknn.search(xb[lo:hi], q.numpy(), k)
                return torch.from_numpy(D.astype(np.float32)), torch.from_numpy(I)

        def cpu_merge(dl, il, k):  # test-only merge (the product default is the HIP kernel)
            D, I = oknn.merge(dl.numpy(), il.numpy(), k)
            return torch.from_numpy(D), torch.from_numpy(I)

        sh = search.ShardedFlatL2(CpuShard(), lo, merge=cpu_merge)
        D, I = sh.search(torch.from_numpy(xq), 5)
        Dw, Iw = oknn.search(xb, xq, 5)
        out[rank] = bool(np.array_equal(I.numpy(), Iw) and np.allclose(D.numpy(), Dw, rtol=1e-6))
    finally:
        dist.destroy_process_group()


def test_sharded_search_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        out = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(out) == {0: True, 1: True}


def test_product_merge_refuses_host_tensors():
    import torch
    from eioku_amd._lib import EiokuHipError

    with pytest.raises(EiokuHipError):
        search.merge_topk(torch.zeros((2, 1, 3)), torch.zeros((2, 1, 3), dtype=torch.int64), 3)
