"""GPU parity: K9 flat-L2 kNN (fp32 MFMA + register top-k) vs float64 brute force.

Bar (BASELINE.json): distances within 1e-4 relative of the float64 truth; ids identical wherever the
truth's own ordering has more margin than that tolerance (ties / near-ties may legitimately swap)."""
import numpy as np
import pytest

from oracle import knn as oknn, prng
from eioku_amd import search, synth

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def unit_rows(seed, n, d):
    x = prng.approx_normal_f32(seed, n * d).reshape(n, d)
    return (x / np.sqrt((x.astype(np.float64) ** 2).sum(1, keepdims=True))).astype(np.float32)


def check(D, I, Dt, It, xb, xq):
    D, I = np.asarray(D), np.asarray(I)
    assert D.shape == Dt.shape and I.dtype == np.int64
    assert np.all(np.diff(D, axis=1) >= 0)  # ascending
    finite = It >= 0
    assert np.array_equal(I >= 0, finite)
    assert np.allclose(D[finite], Dt[finite], rtol=RTOL, atol=1e-6)
    # every returned id's TRUE distance is within tolerance of the true k-th: a valid top-k
    for q in range(D.shape[0]):
        for r in range(D.shape[1]):
            if I[q, r] < 0:
                continue
            true = float(((xq[q].astype(np.float64) - xb[I[q, r]].astype(np.float64)) ** 2).sum())
            assert abs(true - Dt[q, r]) <= RTOL * max(Dt[q, r], 1e-6) + 1e-6
    # where the truth is well separated, ids match exactly
    gap = np.diff(Dt, axis=1, append=Dt[:, -1:] + 1)
    sure = np.ones_like(It, dtype=bool)
    sure[:, :-1] &= gap[:, :-1] > 4 * RTOL * Dt[:, :-1]
    sure[:, 1:] &= gap[:, :-1] > 4 * RTOL * Dt[:, 1:]
    sure &= finite
    assert np.array_equal(I[sure], It[sure])


@pytest.mark.parametrize("n,nq,k,d", [(4096, 16, 10, 384), (8, 8, 10, 384), (1000, 1, 10, 384), (5000, 70, 10, 384),
                                      (777, 33, 1, 384), (3000, 5, 32, 384), (2048, 9, 10, 128), (129, 4, 16, 64),
                                      (20000, 64, 10, 384)])
def test_search_matches_float64_truth(gpu, n, nq, k, d):
    xb = unit_rows(21, n, d)
    xq = unit_rows(22, nq, d)
    ix = search.IndexFlatL2(d)
    ix.add(xb)
    assert ix.ntotal == n
    D, I = ix.search(xq, k)
    Dt, It = oknn.search(xb, xq, k)
    check(D, I, Dt, It, xb, xq)
    ix.close()


def test_golden_flatl2_fixture(gpu):
    from conftest import GOLDEN

    g = np.load(GOLDEN / "flatl2_4096x384.npz")
    xb, xq = unit_rows(int(g["seed_db"]), int(g["n"]), int(g["d"])), unit_rows(int(g["seed_q"]), int(g["nq"]), int(g["d"]))
    assert np.allclose(xq[0, :8], g["xq_first"]) and abs(xb.astype(np.float64).sum() - float(g["xb_sum"])) < 1e-3
    ix = search.IndexFlatL2(384)
    ix.add(xb)
    D, I = ix.search(xq, 10)
    assert np.array_equal(I, g["I"])
    assert np.allclose(D, g["D"], rtol=RTOL, atol=1e-6)


def test_self_query_duplicates_and_incremental_add(gpu):
    """Queries that ARE database rows come back first with distance ~0 (cancellation clamps at 0, never
    negative); duplicated rows tie and are ordered by id; add() in pieces equals one add()."""
    import torch

    xb = unit_rows(5, 3000, 384)
    xb[100] = xb[7]  # duplicate
    ix = search.IndexFlatL2(384)
    for lo in range(0, 3000, 700):
        ix.add(xb[lo:lo + 700])
    D, I = ix.search(xb[[7, 1500, 2999]], 5)
    assert list(I[:, 0]) == [7, 1500, 2999] and I[0, 1] == 100
    assert np.all(D >= 0) and np.all(D[:, 0] < 1e-5)
    # device-resident queries give the same answer as host-staged ones
    Dd, Id = ix.search(torch.from_numpy(xb[[7, 1500, 2999]]).to(gpu), 5)
    assert np.array_equal(Id.cpu().numpy(), I) and np.array_equal(Dd.cpu().numpy(), D)
    ix.reset()
    assert ix.ntotal == 0
    D, I = ix.search(xb[:2], 3)
    assert np.all(I == -1)


def test_merge_kernel_and_sharded_equals_whole(gpu):
    """Row shards searched separately + eioku_topk_merge == one search over everything."""
    import torch

    xb = unit_rows(9, 10007, 384)
    xq = unit_rows(10, 40, 384)
    whole = search.IndexFlatL2(384)
    whole.add(xb)
    Dw, Iw = whole.search(xq, 10)
    dl, il = [], []
    for r in range(3):
        lo, hi = search.shard_bounds(len(xb), 3, r)
        ix = search.IndexFlatL2(384)
        ix.add(xb[lo:hi])
        D, I = ix.search(torch.from_numpy(xq).to(gpu), 10)
        dl.append(D)
        il.append(torch.where(I >= 0, I + lo, I))
    D, I = search.merge_topk(torch.stack(dl), torch.stack(il), 10)
    assert np.array_equal(I.cpu().numpy(), Iw) and np.array_equal(D.cpu().numpy(), Dw)
    Dm, Im = oknn.merge(torch.stack(dl).cpu().numpy(), torch.stack(il).cpu().numpy(), 10)
    assert np.array_equal(Im, Iw) and np.array_equal(Dm, Dw)


def test_full_size_properties_1m(gpu):
    """cfg3 size (1M x 384, generated in HBM): (1) planted exact copies are found at distance ~0;
    (2) results agree with an independent torch fp64 brute force on device; (3) shard+merge == whole."""
    import torch

    n, d, nq, k = 1_000_000, 384, 64, 10
    xb = synth.normal_f32(21, n, d, gpu, l2_normalise=True)
    q = synth.normal_f32(22, nq, d, gpu, l2_normalise=True)
    plant = torch.arange(0, nq // 2, device=gpu) * 31337 % n
    q[: nq // 2] = xb[plant]
    ix = search.IndexFlatL2(d)
    ix.attach(xb)
    D, I = ix.search(q, k)
    assert torch.equal(I[: nq // 2, 0], plant) and float(D[: nq // 2, 0].max()) < 1e-5
    q64 = q.double()
    best_d = torch.full((nq, k), float("inf"), dtype=torch.float64, device=gpu)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=gpu)
    for lo in range(0, n, 250_000):
        blk = xb[lo:lo + 250_000].double()
        dd = (q64 * q64).sum(1)[:, None] + (blk * blk).sum(1)[None, :] - 2 * q64 @ blk.T
        cd, ci = torch.topk(dd, k, dim=1, largest=False)
        alld = torch.cat([best_d, cd], 1)
        alli = torch.cat([best_i, ci + lo], 1)
        o = torch.argsort(alld, dim=1, stable=True)[:, :k]
        best_d, best_i = torch.gather(alld, 1, o), torch.gather(alli, 1, o)
    err = (D.double() - best_d.clamp(min=0)).abs()
    # absolute floor: |q|^2 + |x|^2 - 2q.x cancels to a few ulps of 2.0 (as FAISS' BLAS path does) for the
    # planted copies whose true distance is 0
    assert bool((err <= 5e-6 + RTOL * best_d.abs()).all()), float(err.max())
    assert (I == best_i).float().mean() > 0.995  # near-ties may swap within tolerance
    # shards
    dl, il = [], []
    for r in range(4):
        lo, hi = search.shard_bounds(n, 4, r)
        s = search.IndexFlatL2(d)
        s.set_param("scan_min_rows", 4096)  # 250 k rows: the same search path as the whole index, so the same bytes
        s.attach(xb[lo:hi])
        Ds, Is = s.search(q, k)
        dl.append(Ds)
        il.append(Is + lo)
    Dm, Im = search.merge_topk(torch.stack(dl), torch.stack(il), k)
    assert torch.equal(Im, I) and torch.equal(Dm, D)
