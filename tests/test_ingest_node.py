"""CPU: the per-node ingest launcher (one worker process per GPU, BASELINE cfg4) - planning, ordering, and that every
child has its GPU pinned BEFORE anything of the GPU stack is imported in it."""
import os
import sys

from eioku_amd import ingest


def _jobs(weights):
    return [ingest.Job(f"t{i}", "scene_detection", f"v{i}", f"/videos/{i}.mp4", {}, w) for i, w in enumerate(weights)]


def test_plan_is_longest_first_balanced_and_complete():
    jobs = _jobs([10, 1, 1, 1, 7, 3, 3, 2])
    shares = ingest.plan(jobs, 3)
    assert sorted(i for s in shares for i in s) == list(range(8))
    loads = [sum(jobs[i].weight for i in s) for s in shares]
    assert shares[0][0] == 0 and shares[1][0] == 4 and max(loads) - min(loads) <= 3
    # cfg4: 8 equal videos on 8 GPUs -> one each, in order
    assert ingest.plan(_jobs([1] * 8), 8) == [[i] for i in range(8)]
    assert ingest.plan(_jobs([1] * 3), 8)[:3] == [[0], [1], [2]]


def _fake_worker(gpu, jobs, results, ctx_factory, extra_env):
    """Stands in for ingest._worker on a GPU-less box: performs the same pin and reports what a job would see."""
    os.environ["HIP_VISIBLE_DEVICES"] = str(gpu)
    from eioku_amd import _lib

    clean = "torch" not in sys.modules and _lib._lib is None  # neither torch nor libeioku_hip loaded yet
    for index, job in jobs:
        if job.video_path.endswith("bad.mp4"):
            results.put((index, {"task_id": job.task_id, "status": "failed", "error": "boom", "gpu": gpu}))
        else:
            results.put((index, {"task_id": job.task_id, "status": "completed", "gpu": gpu, "pid": os.getpid(),
                                 "visible": os.environ["HIP_VISIBLE_DEVICES"], "clean_at_entry": clean}))
    results.put((-1, gpu))


def test_run_node_pins_one_process_per_gpu_and_keeps_input_order():
    jobs = _jobs([5, 4, 3, 2, 1, 1])
    jobs[3].video_path = "/videos/bad.mp4"
    out = ingest.run_node(jobs, [0, 1, 2], worker=_fake_worker, timeout=60)
    assert [r["task_id"] for r in out] == [f"t{i}" for i in range(6)]
    ok = [r for r in out if r["status"] == "completed"]
    assert len(ok) == 5 and out[3]["status"] == "failed" and out[3]["error"] == "boom"
    assert all(r["visible"] == str(r["gpu"]) and r["clean_at_entry"] for r in ok)
    assert len({r["pid"] for r in ok}) == 3 and len({(r["pid"], r["gpu"]) for r in ok}) == 3  # one process per GPU


def test_real_worker_pins_before_importing_the_gpu_stack(tmp_path):
    """ingest._worker itself, in a child, on a job that fails fast (no GPU here): the failure is reported per job, the
    pin happened first (the assert inside _worker would kill the child otherwise)."""
    jobs = [ingest.Job("t0", "scene_detection", "v0", str(tmp_path / "missing.npy"), {})]
    out = ingest.run_node(jobs, [5], timeout=120)
    assert out[0]["status"] == "failed" and out[0]["gpu"] == 5 and "Failed to process task t0" in out[0]["error"]


def _dying_worker(gpu, jobs, results, ctx_factory, extra_env):
    """GPU 1's worker finishes one job, then dies without posting its sentinel (GPU fault / OOM kill / segfault)."""
    for n, (index, job) in enumerate(jobs):
        if gpu == 1 and n == 1:
            import time

            time.sleep(1.0)  # the queue's feeder thread has flushed the first result by now
            os._exit(9)
        results.put((index, {"task_id": job.task_id, "status": "completed", "gpu": gpu}))
    results.put((-1, gpu))


def test_a_worker_that_dies_does_not_hang_the_node_or_lose_the_others_results():
    """with timeout=None the parent used to block forever in results.get(); with a timeout
    queue.Empty escaped and the healthy workers' results were lost."""
    jobs = _jobs([1] * 6)
    out = ingest.run_node(jobs, [0, 1], worker=_dying_worker, timeout=None)
    shares = ingest.plan(jobs, 2)
    assert [r["task_id"] for r in out] == [f"t{i}" for i in range(6)]
    for i in shares[0]:
        assert out[i]["status"] == "completed" and out[i]["gpu"] == 0
    assert out[shares[1][0]]["status"] == "completed"              # what the worker finished before it died is kept
    for i in shares[1][1:]:
        assert out[i]["status"] == "failed" and "exit code 9" in out[i]["error"] and out[i]["gpu"] == 1


def _slow_worker(gpu, jobs, results, ctx_factory, extra_env):
    import time

    for index, job in jobs:
        if gpu == 1:
            time.sleep(30)
        results.put((index, {"task_id": job.task_id, "status": "completed", "gpu": gpu}))
    results.put((-1, gpu))


def test_deadline_returns_the_partial_results():
    jobs = _jobs([1] * 4)
    out = ingest.run_node(jobs, [0, 1], worker=_slow_worker, timeout=3)
    shares = ingest.plan(jobs, 2)
    assert all(out[i]["status"] == "completed" for i in shares[0])
    assert all(out[i]["status"] == "failed" and out[i]["error"] == "timed out" for i in shares[1])
