"""CPU oracle for the eioku ml-service hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``eioku_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker.

Every function restates, in numpy / torch-CPU, the algorithm that the reference
(``/root/reference``, codihuston/eioku @ 2026-01-28) runs for one stage of the
path, citing the reference file:line it follows.  Where the arithmetic lives in
an un-vendored third-party dependency (ffmpeg, OpenCV, Ultralytics, torchvision)
or in a library the reference only *plans* to use (PySceneDetect,
sentence-transformers, FAISS), the published algorithm of that library is
restated and marked ``[PUBLIC-LIB]``.

PARITY STATUS: the reference's own tests hold no golden vectors for any numeric
stage, and none of the third-party libraries is installed in the
build container, so every numeric stage here is **parity unpinned**; only the
orchestration (frame sampling, timestamps, result-dict shape, scene-list quirk,
artifact span rules) is pinned, by fixtures captured from the reference's own
Python code (``tests/golden/make_reference_fixtures.py``).
"""
