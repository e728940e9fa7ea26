"""CPU tests of the checkpoint loaders (VERDICT r1 "no real checkpoint has ever been loaded").

No YOLOv8 / MiniLM checkpoint exists offline, so the tests BUILD files of the shapes the real ones have:

  * an Ultralytics-shaped ``.pt``: ``{"model": DetectionModel, "ema": None, ...}`` whose ``model.model`` is an
    ``nn.Sequential`` of ``Conv`` (``conv`` = Conv2d without bias, ``bn`` = BatchNorm2d, eps 1e-3) / ``C2f`` /
    ``SPPF`` / ``Detect`` modules pickled under ``ultralytics.nn.*`` class paths that do NOT exist in this
    container - what ``torch.load`` meets on the GPU box, where ultralytics is not installed either;
  * a Hugging Face ``BertModel`` state dict saved as ``model.safetensors`` / ``pytorch_model.bin``.

They go through ``weights.load_state`` / ``load_checkpoint`` / ``embed.load_state`` (the replacement of
``YOLO(model_path); model.to(device)``, ``/root/reference/ml-service/src/services/model_manager.py:252-254``)
and the folded ``(w, b)`` is compared with a hand fold in float64.
"""
import pickle
import sys
import types

import numpy as np
import pytest
import torch
from torch import nn

from eioku_amd import embed, weights as W


# --- stand-ins for the ultralytics classes, registered under the real module paths only while saving -----------
def _fake_ultralytics():
    mods = {}
    for name in ("ultralytics", "ultralytics.nn", "ultralytics.nn.tasks", "ultralytics.nn.modules",
                 "ultralytics.nn.modules.conv", "ultralytics.nn.modules.block", "ultralytics.nn.modules.head"):
        mods[name] = types.ModuleType(name)

    def cls(module, name, base=nn.Module):
        c = type(name, (base,), {"__module__": module})
        setattr(mods[module], name, c)
        return c

    Conv = cls("ultralytics.nn.modules.conv", "Conv")
    Bottleneck = cls("ultralytics.nn.modules.block", "Bottleneck")
    C2f = cls("ultralytics.nn.modules.block", "C2f")
    SPPF = cls("ultralytics.nn.modules.block", "SPPF")
    Detect = cls("ultralytics.nn.modules.head", "Detect")
    Model = cls("ultralytics.nn.tasks", "DetectionModel")
    return mods, dict(Conv=Conv, Bottleneck=Bottleneck, C2f=C2f, SPPF=SPPF, Detect=Detect, Model=Model)


def _build_model(variant, nc, seed, K):
    """An nn.Module tree whose state_dict keys are exactly Ultralytics' (``model.N...conv.weight`` / ``.bn.*``)."""
    rng = np.random.default_rng(seed)

    def conv(cout, cin, k):
        m = K["Conv"]()
        nn.Module.__init__(m)
        m.conv = nn.Conv2d(cin, cout, k, bias=False)
        m.bn = nn.BatchNorm2d(cout, eps=1e-3)
        with torch.no_grad():
            m.conv.weight.copy_(torch.from_numpy(rng.standard_normal((cout, cin, k, k)).astype(np.float32)))
            m.bn.weight.copy_(torch.from_numpy((1 + 0.2 * rng.standard_normal(cout)).astype(np.float32)))
            m.bn.bias.copy_(torch.from_numpy((0.1 * rng.standard_normal(cout)).astype(np.float32)))
            m.bn.running_mean.copy_(torch.from_numpy((0.3 * rng.standard_normal(cout)).astype(np.float32)))
            m.bn.running_var.copy_(torch.from_numpy((0.5 + rng.random(cout)).astype(np.float32)))
        return m

    def plain(cout, cin):
        c = nn.Conv2d(cin, cout, 1, bias=True)
        with torch.no_grad():
            c.weight.copy_(torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)).astype(np.float32)))
            c.bias.copy_(torch.from_numpy(rng.standard_normal(cout).astype(np.float32)))
        return c

    # place every convolution of weights.conv_table at its dotted path
    root = K["Model"]()
    nn.Module.__init__(root)
    seq = nn.Module()
    root.model = seq
    for name, cout, cin, k, _ in W.conv_table(variant, nc):
        parts = name.split(".")[1:]  # drop the leading "model"
        leaf_is_plain = not name.endswith(".conv")
        if not leaf_is_plain:
            parts = parts[:-1]  # the Conv wrapper owns ".conv" / ".bn"
        node = seq
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, nn.Module())
            node = getattr(node, p)
        node.add_module(parts[-1], plain(cout, cin) if leaf_is_plain else conv(cout, cin, k))
    root.names = {i: f"thing{i}" for i in range(nc)}
    root.stride = torch.tensor([8.0, 16.0, 32.0])
    return root


def _hand_fold(sd, name):
    """float64 fold of Conv2d(no bias) + BatchNorm2d(eps 1e-3), as ``fuse_conv_and_bn`` defines it."""
    w = sd[name + ".weight"].double().numpy()
    bn = name[:-len(".conv")] + ".bn"
    if bn + ".weight" in sd:
        g, b, m, v = (sd[f"{bn}.{s}"].double().numpy() for s in ("weight", "bias", "running_mean", "running_var"))
        sc = g / np.sqrt(v + 1e-3)
        return w * sc[:, None, None, None], b - m * sc
    return w, sd[name + ".bias"].double().numpy()


@pytest.fixture(scope="module")
def ultralytics_pt(tmp_path_factory):
    mods, K = _fake_ultralytics()
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        model = _build_model("n", 3, seed=5, K=K)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        path = tmp_path_factory.mktemp("ckpt") / "yolov8n.pt"
        torch.save({"epoch": -1, "best_fitness": None, "model": model.half(), "ema": None, "updates": None,
                    "optimizer": None, "train_args": {"imgsz": 640}, "date": "2024-01-01", "version": "8.4.8"}, path)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert "ultralytics" not in sys.modules  # the loader must cope without the package, as on the GPU box
    return path, sd


def test_ultralytics_checkpoint_is_read_and_batchnorm_folded(ultralytics_pt):
    path, sd = ultralytics_pt
    state, names = W.load_checkpoint(path, "n", 3)
    assert names == {0: "thing0", 1: "thing1", 2: "thing2"}  # result.names of the checkpoint, not COCO
    table = W.conv_table("n", 3)
    assert sorted(state) == sorted(n for n, *_ in table)
    for name, cout, cin, k, _ in table:
        w, b = state[name]
        assert w.shape == (cout, cin, k, k) and b.shape == (cout,) and w.dtype == np.float32
        # the checkpoint stores fp16 (Ultralytics saves model.half()): fold the fp16-rounded tensors by hand
        sdh = {kk: v.half().float() for kk, v in sd.items()}
        rw, rb = _hand_fold(sdh, name)
        assert np.allclose(w, rw, rtol=2e-6, atol=1e-7), name
        assert np.allclose(b, rb, rtol=2e-6, atol=1e-6), name


def test_plain_state_dict_npz_and_safetensors_give_the_same_state(ultralytics_pt, tmp_path):
    path, sd = ultralytics_pt
    from safetensors.torch import save_file

    ref = W.state_from_tensors({k: v.numpy() for k, v in sd.items()}, "n", 3)
    torch.save({k: v for k, v in sd.items()}, tmp_path / "sd.pt")                     # bare state_dict
    torch.save({"module." + k: v for k, v in sd.items()}, tmp_path / "ddp.pt")        # DataParallel prefix
    np.savez(tmp_path / "sd.npz", **{k: v.numpy() for k, v in sd.items()})
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "sd.safetensors"))
    for f in ("sd.pt", "ddp.pt", "sd.npz", "sd.safetensors"):
        got = W.load_state(tmp_path / f, "n", 3)
        for name in ref:
            assert np.array_equal(got[name][0], ref[name][0]) and np.array_equal(got[name][1], ref[name][1]), (f, name)
    assert W.load_checkpoint(tmp_path / "sd.npz", "n", 3)[1] is None  # no names in a bare state dict


def test_wrong_variant_and_missing_file_fail_loudly(ultralytics_pt, tmp_path):
    path, _ = ultralytics_pt
    with pytest.raises(ValueError, match="shape"):
        W.load_state(path, "s", 3)
    with pytest.raises((KeyError, ValueError)):
        W.load_state(path, "n", 80)
    with pytest.raises(FileNotFoundError):
        W.load_state(tmp_path / "absent.pt", "n", 80)


class _Boom:
    target = "/dev/null"

    def __reduce__(self):
        import os

        return (os.system, (f"touch {self.target}",))


def test_checkpoint_cannot_execute_code_through_reduce(tmp_path):
    """``torch.load(weights_only=False)`` with an unpickler that resolves every importable global runs
    whatever a checkpoint's ``__reduce__`` names.  The loader's unpickler only resolves an allow-list (tensor
    rebuilding, containers, numpy); ``os.system`` becomes an inert stub class, so nothing runs."""
    ran = tmp_path / "ran"
    _Boom.target = str(ran)
    payload = {"model": None, "boom": _Boom()}
    p = tmp_path / "evil.pt"
    torch.save(payload, p)
    pickle.loads(pickle.dumps(_Boom()))  # the stock unpickler does run it ...
    assert ran.exists()
    ran.unlink()
    with pytest.raises(ValueError, match="no tensors"):
        W.load_state(p, "n", 80)
    assert not ran.exists()              # ... the loader's does not
    # and a raw pickle through the same unpickler: the callable is a stub type, calling it builds a stub instance
    import io

    obj = W._TolerantUnpickler(io.BytesIO(pickle.dumps(_Boom()))).load()
    assert isinstance(obj, W._Stub) and not ran.exists()


def test_detector_uses_checkpoint_names_via_model_manager_factory(ultralytics_pt, monkeypatch):
    """from_model_name(path=...) hands the checkpoint's own names to the detector (no GPU: the handle is stubbed)."""
    from eioku_amd import detect as D

    path, _ = ultralytics_pt
    seen = {}

    def fake_init(self, variant, nc, state, names):
        seen.update(variant=variant, nc=nc, names=names, nstate=len(state))

    monkeypatch.setattr(D.Yolov8Detector, "__init__", fake_init)
    monkeypatch.setattr(W, "variant_from_model_name", lambda name: ("n", 3, {0: "a", 1: "b", 2: "c"}))
    D.Yolov8Detector.from_model_name("yolov8n.pt", path=path)
    assert seen["names"] == {0: "thing0", 1: "thing1", 2: "thing2"} and seen["nstate"] == len(W.conv_table("n", 3))


# --- MiniLM ------------------------------------------------------------------------------------------------------
SMALL = dict(embed.MINILM_L6_V2, vocab=300, max_pos=64)


def _hf_bert(cfg, seed):
    from transformers import BertConfig, BertModel

    torch.manual_seed(seed)
    m = BertModel(BertConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"],
                             num_attention_heads=cfg["heads"], intermediate_size=cfg["ffn"],
                             max_position_embeddings=cfg["max_pos"], type_vocab_size=cfg["type_vocab"],
                             layer_norm_eps=cfg["ln_eps"]), add_pooling_layer=False)
    return m.eval()


def test_hf_state_dict_loads_from_safetensors_bin_and_directory(tmp_path):
    from safetensors.torch import save_file

    m = _hf_bert(SMALL, 3)
    sd = {k: v.contiguous() for k, v in m.state_dict().items()}
    d = tmp_path / "all-MiniLM-L6-v2"
    d.mkdir()
    save_file(sd, str(d / "model.safetensors"))
    torch.save({"bert." + k: v for k, v in sd.items()}, tmp_path / "pytorch_model.bin")  # BertForX prefix
    a = embed.load_state(d, SMALL)                      # directory -> model.safetensors
    b = embed.load_state(tmp_path / "pytorch_model.bin", SMALL)
    names = [n for n, _ in embed.tensor_table(SMALL)]
    assert sorted(a) == sorted(names)
    for n in names:
        assert np.array_equal(a[n], sd[n].numpy()) and np.array_equal(b[n], a[n])
    with pytest.raises(ValueError, match="shape"):
        embed.load_state(d, dict(SMALL, vocab=301))
    with pytest.raises(FileNotFoundError):
        embed.load_state(tmp_path / "nope", SMALL)


def test_loaded_hf_weights_reproduce_the_transformers_forward(tmp_path):
    """The state dict that went through embed.load_state drives the float64 oracle to the HF model's own
    sentence embedding (mean pooling + L2 norm): the loader keeps every tensor in its place."""
    from safetensors.torch import save_file

    from oracle import bert as obert

    m = _hf_bert(SMALL, 4)
    save_file({k: v.contiguous() for k, v in m.state_dict().items()}, str(tmp_path / "model.safetensors"))
    state = embed.load_state(tmp_path / "model.safetensors", SMALL)
    rng = np.random.default_rng(0)
    ids = rng.integers(1, SMALL["vocab"], (3, 12)).astype(np.int32)
    mask = np.ones((3, 12), np.uint8)
    mask[1, 7:] = 0
    with torch.no_grad():
        h = m(input_ids=torch.from_numpy(ids.astype(np.int64)), attention_mask=torch.from_numpy(mask.astype(np.int64))
              ).last_hidden_state.double().numpy()
    w = mask[..., None].astype(np.float64)
    pooled = (h * w).sum(1) / np.maximum(w.sum(1), 1e-9)
    want = pooled / np.linalg.norm(pooled, axis=1, keepdims=True)
    got = obert.encode(state, SMALL, ids, mask)
    assert np.abs(got - want).max() < 2e-6
