"""BASELINE cfg5 on one GPU's share: IndexIVFPQ build + search over N x 384 (nlist 4096, m 48, nprobe 32).
python tools/ivfpq_bench.py [N=10_000_000] [nq=1024]  -> one JSON line (times, QPS, recall@10 vs flat L2)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from eioku_amd import _lib, ivfpq, search, synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    nlist, m, nprobe, k, d = 4096, 48, 32, 10, 384
    _lib.init(0)
    gpu = torch.device("cuda:0")
    # clustered data (uniform-random 384-d vectors have no neighbourhood structure for an IVF to find)
    ncl = 20000
    sigma = 0.02  # per coordinate: |noise| ~ 0.39 against unit-norm centres
    centres = synth.normal_f32(5, ncl, d, gpu, l2_normalise=True)
    assign = torch.randint(0, ncl, (n,), device=gpu, generator=torch.Generator(device=gpu).manual_seed(6))
    xb = torch.empty((n, d), dtype=torch.float32, device=gpu)
    step = 2_000_000
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        xb[lo:hi] = centres[assign[lo:hi]] + sigma * synth.normal_f32(100 + lo // step, hi - lo, d, gpu)
    # queries = database rows + a small perturbation: each has one planted true neighbour (inside a cluster all
    # other rows are near-equidistant in 384-d, so an unplanted top-10 is arbitrary for ANY 48-byte code)
    qa = torch.randint(0, n, (nq,), device=gpu, generator=torch.Generator(device=gpu).manual_seed(7))
    q = xb[qa] + 0.1 * sigma * synth.normal_f32(9, nq, d, gpu)
    torch.cuda.synchronize()

    ix = ivfpq.IndexIVFPQ(d, nlist, m, device=gpu)
    t0 = time.perf_counter()
    ix.train(xb)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    t0 = time.perf_counter()
    for lo in range(0, n, step):
        ix.add(xb[lo:min(n, lo + step)])
    ix._pack()
    torch.cuda.synchronize()
    t_add = time.perf_counter() - t0
    ix.nprobe = nprobe
    iters = 5
    times = {}
    ix.scan_mode = "queries"
    lists_only = os.environ.get("IVFPQ_LISTS_ONLY") == "1"  # profiling runs: only the default path in the kernel statistics
    for pre in (() if lists_only else (False, True)):  # look-up tables from the codebook per (query, list) / FAISS' precomputed-table form (default)
        ix.use_precomputed_table = pre
        for _ in range(2):
            D, I = ix.search(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            D, I = ix.search(q, k)
        torch.cuda.synchronize()
        times[pre] = (time.perf_counter() - t0) / iters
    if lists_only:
        times = {True: float("nan"), False: float("nan")}
        ix.use_precomputed_table = True
        iters = 20
    else:
        Dq, Iq = D, I
    ix.scan_mode = "lists"  # round 3: list-major scan (the default)
    for _ in range(2):
        D, I = ix.search(q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        D, I = ix.search(q, k)
    torch.cuda.synchronize()
    t_search = (time.perf_counter() - t0) / iters
    identical = None if lists_only else bool(torch.equal(D, Dq) and torch.equal(I, Iq))
    stats = ix.last_stats.cpu().tolist()

    flat = search.IndexFlatL2(d)
    flat.attach(xb)
    _, It = flat.search(q, k)
    torch.cuda.synchronize()
    hit = (I.unsqueeze(2) == It.unsqueeze(1)).any(dim=2).float().mean().item()
    planted = (I == qa.unsqueeze(1)).any(dim=1).float().mean().item()
    planted_flat = (It[:, 0] == qa).float().mean().item()
    codes_scanned = float(n) / nlist * nprobe * nq * m  # bytes of PQ codes an ideal balanced index would read
    # ... and what THIS index reads: the lists a query probes are the ones near database rows, i.e. the large ones
    offsets, sizes, _, _ = ix._pack()
    _, probes = ix._quantizer.search_many(q, nprobe)
    actual = float(sizes.long()[probes.clamp(min=0)].sum().item()) * m
    szs = sizes.float()
    print(json.dumps({"metric": "IVF-PQ build + search", "n": n, "d": d, "nlist": nlist, "m": m, "nprobe": nprobe, "nq": nq, "k": k,
                      "train_s": t_train, "add_s": t_add, "add_vectors_per_s": n / t_add, "search_ms": t_search * 1e3,
                      "qps": nq / t_search, "search_ms_query_major": times[True] * 1e3, "search_ms_query_major_tables_from_codebook": times[False] * 1e3,
                      "list_major_identical_to_query_major": identical, "overflow_flag": stats[0], "work_items": stats[1], "largest_candidate_list": stats[2],
                      "candidates_per_query": stats[3] / nq,
                      "code_bytes_list_major": float(n) * m, "recall_at_10_vs_flat": hit, "planted_neighbour_in_top10": planted,
                      "planted_neighbour_is_flat_top1": planted_flat,
                      "code_bytes_per_search": codes_scanned, "code_GBps": codes_scanned / t_search / 1e9,
                      "code_bytes_actually_scanned": actual, "actual_code_GBps": actual / t_search / 1e9,
                      "list_size_mean_max": [float(szs.mean().item()), float(szs.max().item())],
                      "mean_probed_list_size": actual / m / nq / nprobe}))


main()
