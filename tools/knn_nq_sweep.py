"""Which path serves a narrow search faster?  python tools/knn_nq_sweep.py [N] -> ms per search at nq in {1..64} through the
register-tile kernels (fp32 rows, one or two passes) and through the scan path (bf16 plane, one pass, bounded + re-ranked)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from eioku_amd import _lib, search, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
_lib.init(0)
gpu = torch.device("cuda:0")
xb = synth.normal_f32(21, n, 384, gpu, l2_normalise=True)
ix = search.IndexFlatL2(384)
ix.attach(xb)
for nq in (1, 4, 16, 32, 64, 128):
    q = synth.normal_f32(22, nq, 384, gpu, l2_normalise=True)
    res = {}
    for name, min_nq in (("register_tile", 100000), ("scan", 1)):
        ix.set_param("scan_min_nq", min_nq)
        for _ in range(2):
            D, I = ix.search(q, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            D, I = ix.search(q, 10)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 5 * 1e3
        res[name + "_I"] = I.clone()
    same = float((res["scan_I"] == res["register_tile_I"]).float().mean())
    print(json.dumps({"n": n, "nq": nq, "register_tile_ms": round(res["register_tile"], 3), "scan_ms": round(res["scan"], 3), "ids_equal": same}), flush=True)
