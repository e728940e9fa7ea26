"""kNN over N x 384 on one GPU, every search path side by side (one process, interleaved rounds):
python tools/knn_bench.py [N] [nq] [mode:rt[:prescan],...] -> JSON lines (ms per search, scan-kernel ms from HIP events)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from eioku_amd import _lib, search, synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    variants = [tuple(int(x) for x in v.split(":")) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0:1", "1:1", "1:2"])]
    _lib.init()
    dev = torch.device("cuda:0")
    xb = synth.normal_f32(21, n, 384, dev, l2_normalise=True)
    q = synth.normal_f32(22, nq, 384, dev, l2_normalise=True)
    ix = search.IndexFlatL2(384)
    ix.attach(xb)
    ref = None
    results = {}
    for rnd in range(3):
        for var in variants:
            mode, rt = var[0], var[1]
            ix.set_param("scan_mode", mode)
            ix.set_param("scan_rt", rt)
            ix.set_param("scan_prescan", var[2] if len(var) > 2 else 32)
            D, I = ix.search(q, 10)  # warm (plane, workspaces)
            torch.cuda.synchronize()
            _lib.prof_enable(True, tags=[_lib.PROF_KNN])
            _lib.prof_reset()
            t0 = time.perf_counter()
            for _ in range(3):
                D, I = ix.search(q, 10)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            _lib.prof_enable(False)
            ms, cnt = _lib.prof_read(_lib.PROF_KNN)
            if ref is None:
                ref = I.clone()
            results.setdefault(tuple(var), []).append((dt * 1e3, ms / max(cnt, 1), float((I == ref).float().mean())))
    for var, r in results.items():
        mode, rt = var[0], var[1]
        best = min(x[0] for x in r)
        print(json.dumps({"n": n, "nq": nq, "scan_mode": mode, "scan_rt": rt, "scan_prescan": var[2] if len(var) > 2 else 32, "ms_per_search_min": best,
                          "ms_per_search_all": [round(x[0], 3) for x in r], "main_kernel_ms": [round(x[1], 3) for x in r],
                          "qps": nq / best * 1e3, "ids_equal_to_first_variant": r[0][2],
                          "algorithmic_TFLOPs_on_main_kernel": 2.0 * nq * n * 384 / (min(x[1] for x in r) * 1e-3) / 1e12}), flush=True)


main()
