"""GPU parity of the wide-search ("scan") path of K9: database tiles stationary in registers, every query tile
streamed past them, per-query candidate lists bounded by an exact search of a strided sample, k-best selection
(csrc/knn.hip: k_split_planes / k_l2_scan / k_scan_select).

Same bar as test_knn_gpu.py: distances within 1e-4 relative of the float64 truth, ids identical wherever the truth
is separated by more than that.  The scan filters with ONE bf16 product term and a rigorous margin, the distances that
come back are fp32 sums of (q - x)^2 over the fp32 rows.  `scan_min_rows` is lowered so that small databases take the
path; scan_rt picks the 12-wave / one-row-tile or the 8-wave / two-row-tile kernel; the full-size case runs at
BASELINE's 10 M x 384."""
import numpy as np
import pytest

from oracle import knn as oknn
from eioku_amd import search, synth
from test_knn_gpu import check, unit_rows

pytestmark = pytest.mark.gpu


def scan_index(d, mode=1, **params):
    ix = search.IndexFlatL2(d)
    ix.set_param("scan_min_rows", 4096)
    ix.set_param("scan_mode", mode)
    for k, v in params.items():
        ix.set_param(k, v)
    return ix


@pytest.mark.parametrize("n,nq,k,d,rt", [
    (20000, 70, 10, 384, 1), (5000, 130, 10, 128, 1), (33333, 200, 16, 256, 1), (4099, 65, 2, 384, 1),
    (50001, 97, 10, 384, 1), (8192, 1024, 10, 384, 1), (50001, 300, 10, 384, 1), (40000, 129, 10, 128, 1),
    (70001, 257, 10, 384, 2), (50033, 300, 10, 384, 2), (20000, 100, 10, 128, 2), (45000, 200, 10, 256, 2),
    (4096, 66, 5, 384, 2), (30000, 40, 1, 384, 2), (30000, 100, 32, 384, 2), (25000, 7, 24, 128, 1), (6000, 5, 1, 256, 2),
])
def test_scan_matches_float64_truth(gpu, n, nq, k, d, rt):
    xb = unit_rows(31, n, d)
    xq = unit_rows(32, nq, d)
    # near neighbours (distance ~4e-6) and an exact copy: the re-rank sums (q - x)^2 directly, so the cancellation of
    # |q|^2 + |x|^2 - 2 q.x (a few ulps of 2.0 = ~1e-6 absolute, what the register-tile kernels carry, as FAISS' BLAS
    # path does) never enters the result
    xq[:4] = xb[[5, n // 2, n - 1, 5]] + 0.002 * xq[:4]
    xq[4] = xb[77]
    ix = scan_index(d, scan_rt=rt)
    ix.add(xb)
    D, I = ix.search(xq, k)
    Dt, It = oknn.search(xb, xq, k)
    check(D, I, Dt, It, xb, xq)
    assert I[4, 0] == 77 and D[4, 0] == 0.0
    ix.close()


@pytest.mark.parametrize("n,nq,k,d,rt,stride", [(20000, 70, 10, 384, 2, 4), (50001, 300, 10, 384, 1, 8), (33333, 200, 16, 256, 2, 2),
                                                (40000, 129, 10, 128, 2, 4), (9000, 40, 3, 384, 2, 0)])
def test_prescan_bound_keeps_the_answer_exact(gpu, n, nq, k, d, rt, stride):
    """The bound is tightened by scanning every `stride`-th row tile first (k-th best of that subset) before the full
    scan; small strides so that small databases have a meaningful subset (the default, 32, needs >= 64 sampled tiles
    and is exercised by the 10 M case); 0 = the sample alone.  A tiny candidate cap truncates the pre-scan's lists:
    still a valid bound, still the exact answer."""
    xb = unit_rows(33, n, d)
    for cap in (4096, 64):
        xq = unit_rows(34, nq, d)
        if cap == 4096:  # (a final list that overflows goes to the register-tile kernels, whose |q|^2 + |x|^2 - 2 q.x
            xq[:3] = xb[[7, n // 3, n - 2]] + 0.002 * xq[:3]  # carries ~5e-5 absolute on near-zero distances)
        Dt, It = oknn.search(xb, xq, k)
        ix = scan_index(d, scan_rt=rt, scan_prescan=stride, scan_cap=cap)
        ix.add(xb)
        D, I = ix.search(xq, k)
        check(D, I, Dt, It, xb, xq)
        ix.close()


def test_scan_agrees_with_register_tile_kernels_and_is_deterministic(gpu):
    import torch

    n, nq, d = 60000, 300, 384
    xb = torch.from_numpy(unit_rows(41, n, d)).to(gpu)
    xq = torch.from_numpy(unit_rows(42, nq, d)).to(gpu)
    res = {}
    for mode in (0, 1):
        ix = scan_index(d, mode)
        ix.attach(xb)
        res[mode] = [tuple(t.clone() for t in ix.search(xq, 10)) for _ in range(2)]
        assert torch.equal(res[mode][0][0], res[mode][1][0]) and torch.equal(res[mode][0][1], res[mode][1][1])
        ix.close()
    D0, I0 = res[0][0]
    D, I = res[1][0]
    assert (I == I0).float().mean() > 0.999
    assert torch.allclose(D, D0, rtol=1e-4, atol=2e-6)


def test_candidate_list_overflow_falls_back_to_the_exact_kernels(gpu):
    """A list of 16 slots cannot hold the ~20 candidates the sample bound admits per query: the overflow flag gates
    the register-tile search in, and the answer is still the exact one (rule 26: force the rare branch)."""
    n, nq, d = 20000, 100, 384
    xb, xq = unit_rows(51, n, d), unit_rows(52, nq, d)
    Dt, It = oknn.search(xb, xq, 10)
    for rt in (1, 2):
        ix = scan_index(d, scan_cap=16, scan_rt=rt)
        ix.add(xb)
        D, I = ix.search(xq, 10)
        check(D, I, Dt, It, xb, xq)
        ix.set_param("scan_cap", 4096)  # and back on the fast path with the same handle
        D, I = ix.search(xq, 10)
        check(D, I, Dt, It, xb, xq)
        ix.close()


def test_incremental_add_extends_the_planes_and_ties_order_by_id(gpu):
    d = 384
    xb = unit_rows(61, 17777, d)
    xb[9000] = xb[123]
    xb[17000] = xb[123]   # three copies of one row, one of them in the second add()
    xq = unit_rows(62, 80, d)
    xq[0] = xb[123]
    ix = scan_index(d)
    ix.add(xb[:10000])
    D, I = ix.search(xq, 10)
    check(D, I, *oknn.search(xb[:10000], xq, 10), xb[:10000], xq)
    assert list(I[0, :2]) == [123, 9000]
    ix.add(xb[10000:])     # partial last tile of the first batch is rebuilt, the rest appended
    D, I = ix.search(xq, 10)
    check(D, I, *oknn.search(xb, xq, 10), xb, xq)
    assert list(I[0, :3]) == [123, 9000, 17000] and np.all(D[0, :3] == 0)
    ix.close()


def test_more_than_1024_queries_run_in_groups(gpu):
    n, nq, d = 30000, 1100, 128
    xb, xq = unit_rows(71, n, d), unit_rows(72, nq, d)
    ix = scan_index(d)
    ix.add(xb)
    D, I = ix.search(xq, 10)
    check(D, I, *oknn.search(xb, xq, 10), xb, xq)
    ix.close()


def test_unnormalised_vectors_keep_the_margin_rigorous(gpu):
    """The one-term filter's margin scales with |q| |x|: rows of very different norms must not lose neighbours."""
    rng = np.random.default_rng(5)
    n, nq, d = 25000, 90, 384
    xb = unit_rows(81, n, d) * rng.uniform(0.2, 6.0, (n, 1)).astype(np.float32)
    xq = unit_rows(82, nq, d) * rng.uniform(0.2, 6.0, (nq, 1)).astype(np.float32)
    for rt in (1, 2):
        ix = scan_index(d, scan_rt=rt)
        ix.add(xb)
        D, I = ix.search(xq, 10)
        check(D, I, *oknn.search(xb, xq, 10), xb, xq)
        ix.close()


def test_search_after_and_search_many(gpu):
    """eioku_index_search_after: the next k results after a previous answer; chained for k > 32 (IVF nprobe > 32)."""
    n, nq, d = 3000, 7, 384
    xb, xq = unit_rows(91, n, d), unit_rows(92, nq, d)
    xb[2000] = xb[11]
    xq[0] = xb[11]  # a tie at distance 0 across a round boundary must not be lost or repeated
    ix = search.IndexFlatL2(d)
    ix.add(xb)
    Dt, It = oknn.search(xb, xq, 70)
    D, I = ix.search_many(xq, 70)
    assert D.shape == (nq, 70) and np.all(np.diff(D, axis=1) >= 0)
    for q in range(nq):
        assert len(set(I[q])) == 70
    assert np.allclose(D, Dt, rtol=1e-4, atol=1e-6)
    assert (I == It).mean() > 0.98 and list(I[0, :2]) == [11, 2000]
    D1, I1 = ix.search(xq, 5)
    D2, I2 = ix.search_after(xq, 5, D1[:, -1], I1[:, -1])
    assert np.array_equal(np.concatenate([I1, I2], 1), I[:, :10])
    small = search.IndexFlatL2(d)
    small.add(xb[:40])
    D, I = small.search_many(xq, 64)  # fewer rows than k: padded with -1
    assert np.all(I[:, 40:] == -1) and np.all(I[:, :40] >= 0)
    assert np.all(D[:, 40:] == np.finfo(np.float32).max)  # FAISS pads distances with FLT_MAX


def test_full_size_10m_scan_equals_register_tile_search(gpu):
    """BASELINE metric size (10 M x 384, generated in HBM): both scan kernels return the register-tile kernels' ids,
    planted copies come back first at distance ~0, and a strip of queries agrees with a torch fp64 brute force."""
    import torch

    n, d, nq, k = 10_000_000, 384, 1024, 10
    xb = synth.normal_f32(21, n, d, gpu, l2_normalise=True)
    q = synth.normal_f32(22, nq, d, gpu, l2_normalise=True)
    plant = (torch.arange(0, 64, device=gpu) * 154_321 + 7) % n
    q[:64] = xb[plant]
    ix = search.IndexFlatL2(d)
    ix.attach(xb)
    ix.set_param("scan_mode", 0)
    D0, I0 = ix.search(q, k)
    ix.set_param("scan_mode", 1)
    for rt in (2, 1):
        ix.set_param("scan_rt", rt)
        D, I = ix.search(q, k)
        assert torch.equal(I[:64, 0], plant) and float(D[:64, 0].max()) == 0.0
        assert (I == I0).float().mean() > 0.999, rt  # near-ties within the 1e-4 band may swap
        # (the register-tile kernel's own absolute error on a self-distance is ~5e-5: truncating bf16 split)
        assert bool(((D - D0).abs() <= 1e-4 + 1e-4 * D0).all()), rt
    # independent check of 16 queries against fp64 on device
    q64 = q[60:76].double()
    best = torch.full((16, k), float("inf"), dtype=torch.float64, device=gpu)
    for lo in range(0, n, 500_000):
        blk = xb[lo:lo + 500_000].double()
        dd = (q64 * q64).sum(1)[:, None] + (blk * blk).sum(1)[None, :] - 2 * q64 @ blk.T
        best = torch.cat([best, torch.topk(dd, k, dim=1, largest=False).values], 1).sort(dim=1).values[:, :k]
    err = (D[60:76].double() - best.clamp(min=0)).abs()
    assert bool((err <= 5e-6 + 1e-4 * best.abs()).all()), float(err.max())
    ix.close()


@pytest.mark.parametrize("rt", [1, 2])
def test_rows_just_under_bf16_midpoints_are_never_filtered_out(gpu, rt):
    """the scan's filter margin must cover round-to-nearest bf16 on BOTH operands:
    |q.x - bf(q).bf(x)| <= (2^-7 + 2^-16) sum|q_i x_i|.  q = x = 1.0039053 in every coordinate sits just under a bf16
    rounding midpoint (bf16 spacing at 1.0 is 2^-7): the bf16 product is 384.0 against a true 387.02, so with half that
    margin (round 2) the exact copy was filtered out whenever the bound tau was below ~2.98.  400 rows are
    near-duplicates of the query (distance^2 ~ 0.15): the sample and the pre-scan make tau small, the copy must
    still come back first at distance exactly 0, and so must unnormalised near-midpoint rows of other signs."""
    n, d, k = 60000, 384, 10
    rng = np.random.default_rng(41)
    base = np.full((d,), 1.0039053, dtype=np.float32)
    sign = np.where(np.arange(d) % 3 == 0, -1.0, 1.0).astype(np.float32)
    xb = (3.0 * rng.standard_normal((n, d))).astype(np.float32)                 # far rows
    near = rng.choice(n, 400, replace=False)  # enough for the sample (>= 32768 rows) to hold k of them, few enough for the candidate lists
    xb[near] = base + 0.02 * rng.standard_normal((len(near), d)).astype(np.float32)
    near2 = near[: len(near) // 2]
    xb[near2] = sign * xb[near2]
    xb[near[0]] = base * sign                                                   # the exact copies
    xb[near[-1]] = base
    xq = np.stack([base, base * sign, 2.0 * base]).astype(np.float32)           # 2 x base: also midpoints, no copy
    ix = scan_index(d, scan_rt=rt)
    ix.add(xb)
    D, I = ix.search(xq, k)
    Dt, It = oknn.search(xb, xq, k)
    assert I[0, 0] == near[-1] and D[0, 0] == 0.0
    assert I[1, 0] == near[0] and D[1, 0] == 0.0
    check(D, I, Dt, It, xb, xq)
    ix.close()


def test_large_norm_rows_with_small_distances(gpu):
    """the bound's slack must scale with the norms (the sample's distances are |q|^2 + |x|^2 - 2 q.x in
    fp32, absolute error ~1e-3 at norm 100): rows of norm ~100 whose true nearest distances are ~2.6e-2."""
    n, d, k = 50000, 128, 10
    rng = np.random.default_rng(43)
    centre = (100.0 / np.sqrt(d)) * np.ones((d,), np.float32)
    # far rows at distance^2 ~ 1150: beyond the filter's margin, which scales with the norms (2 x 2^-7 |q| |x| ~ 160 here)
    xb = (centre + 3.0 * rng.standard_normal((n, d))).astype(np.float32)
    near = rng.choice(n, 300, replace=False)
    xb[near] = centre + 0.01 * rng.standard_normal((300, d)).astype(np.float32)  # distance^2 ~ 0.026: tau is tiny
    xb[123] = centre + 0.01 * rng.standard_normal(d).astype(np.float32)
    xq = (centre + 0.01 * rng.standard_normal((16, d))).astype(np.float32)
    xq[0] = xb[123]
    ix = scan_index(d)
    ix.add(xb)
    D, I = ix.search(xq, k)
    Dt, It = oknn.search(xb, xq, k)
    assert I[0, 0] == 123 and D[0, 0] == 0.0
    # ids exact wherever the float64 truth separates them; distances to 1e-4 relative
    check(D, I, Dt, It, xb, xq)
    ix.close()
