"""CPU: the classify_places drop-in (host logic: sampling, timestamps, labels, result dict) against the dicts captured from
the reference's own loop (tests/golden/ref_places_loop.json, made by make_reference_fixtures.py with stub cv2 /
torchvision around the real torch and PIL), and the product's resize tables against Pillow's arithmetic."""
import asyncio
import json

import numpy as np
import pytest

from conftest import GOLDEN
from eioku_amd import places, task_handler
from eioku_amd.model_manager import ModelManager
from test_host_boundary import ScriptedSource

CASES = json.loads((GOLDEN / "ref_places_loop.json").read_text())


def scripted_logits(seed: int, frame_idx: int) -> np.ndarray:
    """make_reference_fixtures.places_logits: what the stub model returned for this frame"""
    rng = np.random.default_rng(seed * 1_000_003 + frame_idx)
    return (3.0 * rng.standard_normal(365)).astype(np.float32)


class ScriptedClassifier:
    """Stands in for the HIP classifier: replays the scripted logits through torch's own softmax / sort (the HIP head is
    compared with that arithmetic in tests/test_places_gpu.py)."""

    def __init__(self, case, cache_dir):
        self.seed = case["seed"]
        self.labels = places.load_labels(cache_dir)
        self.batches, self.frames = [], []

    def classify(self, frames, top_k):
        import torch

        self.batches.append(len(frames))
        probs, idx = [], []
        for f in frames:
            i = int(f[0, 0, 0]) | (int(f[0, 0, 1]) << 8) | (int(f[0, 0, 2]) << 16)
            self.frames.append(i)
            h = torch.nn.functional.softmax(torch.from_numpy(scripted_logits(self.seed, i))[None], 1).squeeze()
            p, k = h.sort(0, True)
            probs.append(p[:top_k].numpy())
            idx.append(k[:top_k].numpy())
        return np.stack(probs), np.stack(idx)


@pytest.mark.parametrize("case", CASES, ids=[f"{c['fps']}-{c['total_frames']}" for c in CASES])
@pytest.mark.parametrize("batch", [1, 4, 64])
def test_classify_places_equals_reference_capture(case, batch, tmp_path):
    if case["label_file"]:
        d = tmp_path / "places365"
        d.mkdir()
        (d / "categories_places365.txt").write_text("\n".join(case["label_lines"]) + "\n")
    made = []

    def factory(cache_dir):
        made.append(ScriptedClassifier(case, cache_dir))
        return made[-1]

    mm = ModelManager(cache_dir=str(tmp_path), frame_source=lambda p: ScriptedSource(case["fps"], case["total_frames"]),
                      place_classifier_factory=factory, batch_size=batch)
    got = asyncio.run(mm.classify_places("/videos/fake.mp4", dict(case["config"])))
    assert got == case["result"]  # ints, labels and every float bit for bit
    assert json.dumps(got) == json.dumps(case["result"])
    assert made[0].frames == case["model_calls"] and all(b <= batch for b in made[0].batches)


def test_place_detection_task_maps_to_place_classification_artifacts(tmp_path, monkeypatch):
    case = CASES[1]
    monkeypatch.setenv("MODEL_CACHE_DIR", str(tmp_path))
    (tmp_path / "places365").mkdir()
    (tmp_path / "places365" / "categories_places365.txt").write_text("\n".join(case["label_lines"]) + "\n")
    sink = []
    ctx = {"artifact_sink": sink.extend,
           "model_manager_factory": lambda cache_dir: ModelManager(
               cache_dir=cache_dir, frame_source=lambda p: ScriptedSource(case["fps"], case["total_frames"]),
               place_classifier_factory=lambda cd: ScriptedClassifier(case, cd))}
    out = asyncio.run(task_handler.process_ml_task(ctx, "t9", "place_detection", "vid", "/videos/fake.mp4", dict(case["config"])))
    n = len(case["result"]["classifications"])
    assert out == {"task_id": "t9", "status": "completed", "artifact_count": n}
    assert [e.artifact_type for e in sink] == ["place.classification"] * n
    for e, c in zip(sink, case["result"]["classifications"]):
        assert e.span_start_ms == e.span_end_ms == c["timestamp_ms"] and json.loads(e.payload_json) == c


def test_label_file_parsing_and_fallback(tmp_path):
    assert places.load_labels(tmp_path) == [f"place_{i}" for i in range(365)]
    d = tmp_path / "places365"
    d.mkdir()
    (d / "categories_places365.txt").write_text("/a/airfield 0\n/a/airplane_cabin 1\n/b/bar 2\n")
    assert places.load_labels(tmp_path) == ["airfield", "airplane_cabin", "bar"]


@pytest.mark.parametrize("n_in", [1920, 1080, 854, 480, 224, 300, 100, 9])
def test_resize_tables_reproduce_pillow(n_in):
    """The product's coefficient tables drive a resample that equals Pillow's on a one-row image (the kernels apply
    exactly these integer taps; the full two-pass resize is compared on the device in tests/test_places_gpu.py)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(n_in)
    row = rng.integers(0, 256, (1, n_in, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(row).resize((224, 1), Image.BILINEAR))
    b, k, ks = places.resize_tables(n_in)
    assert k.shape == (224, ks) and b[0, 0] == 0 and b[-1].sum() == n_in
    got = np.empty((1, 224, 3), np.uint8)
    for xx in range(224):
        lo, n = b[xx]
        acc = (row[0, lo:lo + n].astype(np.int64) * k[xx, :n, None].astype(np.int64)).sum(0)
        got[0, xx] = np.clip((acc + (1 << 21)) >> 22, 0, 255)
    assert np.array_equal(got, want)


def test_fold_state_handles_the_release_checkpoint_layout():
    """`{"state_dict": {"module.conv1.weight": ...}}` (model_manager.py:612-621) -> folded per-convolution weights."""
    from oracle import places as op

    rng = np.random.default_rng(2)
    sd = {}

    def conv_bn(cname, bname, co, ci, k):
        sd["module." + cname + ".weight"] = rng.standard_normal((co, ci, k, k)).astype(np.float32)
        for nm, v in (("weight", rng.uniform(0.5, 1.5, co)), ("bias", rng.standard_normal(co)),
                      ("running_mean", rng.standard_normal(co)), ("running_var", rng.uniform(0.5, 2, co))):
            sd[f"module.{bname}.{nm}"] = v.astype(np.float32)
        sd[f"module.{bname}.num_batches_tracked"] = np.int64(7)

    for name, co, ci, k, _ in op.LAYERS:
        if name == "conv1":
            conv_bn("conv1", "bn1", co, ci, k)
        elif "downsample" in name:
            conv_bn(name, name[:-1] + "1", co, ci, k)
        else:
            conv_bn(name, name.replace("conv", "bn"), co, ci, k)
    sd["module.fc.weight"] = rng.standard_normal((365, 512)).astype(np.float32)
    sd["module.fc.bias"] = rng.standard_normal(365).astype(np.float32)
    st = places.fold_state({"state_dict": sd, "epoch": 90})
    assert sorted(st) == sorted([n for n, *_ in op.LAYERS] + ["fc"])
    w, b = st["layer2.0.downsample.0"]
    fw, fb = op.fold_bn(sd["module.layer2.0.downsample.0.weight"], sd["module.layer2.0.downsample.1.weight"],
                        sd["module.layer2.0.downsample.1.bias"], sd["module.layer2.0.downsample.1.running_mean"],
                        sd["module.layer2.0.downsample.1.running_var"])
    assert np.array_equal(w, fw) and np.array_equal(b, fb) and st["fc"][0].shape == (365, 512)
