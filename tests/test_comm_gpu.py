"""C1 through the C ABI (csrc/comm.hip): ``eioku_index_search_sharded`` = local shard search + ONE RCCL all-gather of
the packed answers + local merge.  A one-GPU box can only form a world of one (RCCL refuses two ranks on one device),
which still runs every step -- RCCL bound by dlopen, communicator bootstrap, pack, the collective, unpack, merge -- on
the real library; the world-size-2 arithmetic of the same protocol is covered on CPU by tests/test_oracle_knn.py
(gloo) and by the shard + merge == whole-index GPU tests of tests/test_knn_gpu.py."""
import numpy as np
import pytest

from oracle import knn as oknn
from eioku_amd import _lib, search
from test_knn_gpu import check, unit_rows

pytestmark = pytest.mark.gpu


def test_world_of_one_equals_the_plain_search_with_global_ids(gpu):
    import torch

    n, nq, d, k = 5000, 37, 384, 10
    xb, xq = unit_rows(3, n, d), unit_rows(4, nq, d)
    ix = search.IndexFlatL2(d)
    ix.add(xb)
    uid = search.RcclComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = search.RcclComm(uid, 0, 1)
    sharded = search.CommShardedFlatL2(ix, id_base=1_000_000, comm=comm)
    q = torch.from_numpy(xq).to(gpu)
    D, I = sharded.search(q, k)
    Dt, It = oknn.search(xb, xq, k)
    check(D.cpu().numpy(), I.cpu().numpy() - 1_000_000, Dt, It, xb, xq)
    D2, I2 = sharded.search(q, k)  # workspace reuse, deterministic
    assert torch.equal(D, D2) and torch.equal(I, I2)
    # odd nq * k (the 4-byte pad word of the message) and fewer rows than k (padding ids stay -1, not id_base - 1)
    small = search.IndexFlatL2(d)
    small.add(xb[:3])
    D3, I3 = search.CommShardedFlatL2(small, 500, comm).search(q[:3], 5)
    assert np.array_equal(I3.cpu().numpy()[:, 3:], -np.ones((3, 2), np.int64))
    assert set(I3.cpu().numpy()[:, :3].ravel()) == {500, 501, 502}
    assert np.all(D3.cpu().numpy()[:, 3:] == np.finfo(np.float32).max)
    with pytest.raises(_lib.EiokuHipError):
        sharded.search(xq, k)  # host queries are refused, not staged
    comm.close()
    ix.close()
    small.close()


def test_bad_arguments_are_refused_before_any_collective(gpu):
    with pytest.raises(ValueError):
        search.RcclComm(b"short", 0, 1)
    with pytest.raises(_lib.EiokuHipError):
        search.RcclComm(bytes(128), 2, 2)  # rank out of range
