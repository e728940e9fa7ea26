"""Segment-embedding stage: all-MiniLM-L6-v2 on the HIP encoder (``eioku_bert_t``).

The reference planned this stage but never built it (``.kiro/specs/semantic-video-search/tasks.md:
297-302``); the surface mirrored here is sentence-transformers' ``SentenceTransformer.encode`` for
that model: WordPiece tokens -> BERT(6 x 384) -> attention-mask mean pooling -> L2 normalise.
PyTorch / safetensors are used only to read the checkpoint; tokenisation stays on the host.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import _lib
from ._buffers import current_stream, on_device, ptr

MINILM_L6_V2 = dict(vocab=30522, hidden=384, layers=6, heads=12, ffn=1536, max_pos=512, type_vocab=2, ln_eps=1e-12)


def tensor_table(cfg: dict) -> list[tuple[str, tuple[int, ...]]]:
    """(state-dict name, shape) of every tensor the encoder needs (Hugging Face ``BertModel`` names)."""
    H, F = cfg["hidden"], cfg["ffn"]
    t = [("embeddings.word_embeddings.weight", (cfg["vocab"], H)),
         ("embeddings.position_embeddings.weight", (cfg["max_pos"], H)),
         ("embeddings.token_type_embeddings.weight", (cfg["type_vocab"], H)),
         ("embeddings.LayerNorm.weight", (H,)), ("embeddings.LayerNorm.bias", (H,))]
    for l in range(cfg["layers"]):
        p = f"encoder.layer.{l}."
        t += [(p + "attention.self.query.weight", (H, H)), (p + "attention.self.key.weight", (H, H)),
              (p + "attention.self.value.weight", (H, H)), (p + "attention.self.query.bias", (H,)),
              (p + "attention.self.key.bias", (H,)), (p + "attention.self.value.bias", (H,)),
              (p + "attention.output.dense.weight", (H, H)), (p + "attention.output.dense.bias", (H,)),
              (p + "attention.output.LayerNorm.weight", (H,)), (p + "attention.output.LayerNorm.bias", (H,)),
              (p + "intermediate.dense.weight", (F, H)), (p + "intermediate.dense.bias", (F,)),
              (p + "output.dense.weight", (H, F)), (p + "output.dense.bias", (H,)),
              (p + "output.LayerNorm.weight", (H,)), (p + "output.LayerNorm.bias", (H,))]
    return t


def random_state(cfg: dict, seed: int) -> dict[str, np.ndarray]:
    """Random-init weights of the exact shapes (BERT init: N(0, 0.02^2) x a gain that keeps
    post-LayerNorm activations O(1); LayerNorm gamma ~ 1, beta ~ 0)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in tensor_table(cfg):
        if name.endswith("LayerNorm.weight"):
            out[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith(".bias"):
            out[name] = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        elif "embeddings" in name:
            out[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        else:
            out[name] = (rng.standard_normal(shape) / np.sqrt(shape[1])).astype(np.float32)
    return out


def load_state(path: str | Path, cfg: dict) -> dict[str, np.ndarray]:
    """``model.safetensors`` / ``pytorch_model.bin`` of all-MiniLM-L6-v2 (keys with or without ``bert.``)."""
    path = Path(path)
    if path.is_dir():
        for cand in ("model.safetensors", "pytorch_model.bin"):
            if (path / cand).exists():
                path = path / cand
                break
    if not path.exists():
        raise FileNotFoundError(f"encoder weights not found: {path}")
    if path.suffix == ".safetensors":
        from safetensors.numpy import load_file

        raw = load_file(str(path))
    else:
        import torch

        raw = {k: v.float().numpy() for k, v in torch.load(str(path), map_location="cpu", weights_only=True).items()}
    raw = {k.removeprefix("bert.").removeprefix("0.auto_model."): v for k, v in raw.items()}
    out = {}
    for name, shape in tensor_table(cfg):
        if name not in raw:
            raise KeyError(f"checkpoint has no {name}")
        a = np.asarray(raw[name], dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {a.shape} != {shape}")
        out[name] = a
    return out


class MiniLMEncoder:
    """BERT sentence encoder resident on one GPU."""

    def __init__(self, state: dict[str, np.ndarray] | None = None, cfg: dict = MINILM_L6_V2, tokenizer=None):
        lib = _lib.load()
        _lib.init()
        self._lib, self.cfg, self.tokenizer = lib, dict(cfg), tokenizer
        h = C.c_void_p()
        _lib.check(lib.eioku_bert_create(cfg["vocab"], cfg["hidden"], cfg["layers"], cfg["heads"], cfg["ffn"],
                                         cfg["max_pos"], cfg["type_vocab"], float(cfg["ln_eps"]), C.byref(h)),
                   "eioku_bert_create")
        self._h = h
        if state is not None:
            self.load_state(state)

    def load_state(self, state: dict[str, np.ndarray]) -> None:
        lib = self._lib
        for i in range(lib.eioku_bert_num_tensors(self._h)):
            name = C.create_string_buffer(160)
            r, c = C.c_int(), C.c_int()
            _lib.check(lib.eioku_bert_tensor_info(self._h, i, name, 160, C.byref(r), C.byref(c)), "eioku_bert_tensor_info")
            a = np.ascontiguousarray(state[name.value.decode()], dtype=np.float32)
            if a.size != r.value * c.value:
                raise ValueError(f"{name.value.decode()}: {a.shape} does not have {r.value}x{c.value} elements")
            _lib.check(lib.eioku_bert_set_tensor(self._h, i, ptr(a), a.size), f"eioku_bert_set_tensor({name.value.decode()})")

    def encode_ids(self, ids, mask):
        """int32 ids / uint8 mask ``(B,S)`` -> float32 ``(B,hidden)`` unit vectors (numpy in -> numpy out)."""
        B, S = (int(s) for s in ids.shape)
        H = self.cfg["hidden"]
        dev = on_device(ids)
        if dev:
            import torch

            ids = ids.to(torch.int32).contiguous()
            mask = mask.to(torch.uint8).contiguous()
            out = torch.empty((B, H), dtype=torch.float32, device=ids.device)
        else:
            ids = np.ascontiguousarray(ids, dtype=np.int32)
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            out = np.empty((B, H), dtype=np.float32)
        _lib.check(self._lib.eioku_bert_embed(self._h, ptr(ids), ptr(mask), B, S, ptr(out),
                                              _lib.MEM_DEVICE if dev else _lib.MEM_HOST, current_stream(ids)),
                   "eioku_bert_embed")
        return out

    def encode(self, texts: list[str], max_seq_length: int = 256):
        """sentence-transformers style entry: needs a WordPiece tokenizer (``tokenizers.Tokenizer`` or HF fast
        tokenizer) supplied at construction - the vocabulary file is not part of this repository."""
        if self.tokenizer is None:
            raise RuntimeError("MiniLMEncoder.encode() needs a tokenizer (vocab.txt of all-MiniLM-L6-v2); "
                               "use encode_ids() with pre-tokenised input otherwise")
        if hasattr(self.tokenizer, "encode_batch"):  # eioku_amd.semantic.WordPieceTokenizer
            return self.encode_ids(*self.tokenizer.encode_batch(texts, max_seq_length))
        enc = self.tokenizer(texts, padding=True, truncation=True, max_length=max_seq_length, return_tensors="np")
        return self.encode_ids(enc["input_ids"].astype(np.int32), enc["attention_mask"].astype(np.uint8))

    def last_flops(self) -> float:
        f = C.c_double(0)
        _lib.check(self._lib.eioku_bert_last_flops(self._h, C.byref(f)), "eioku_bert_last_flops")
        return f.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.eioku_bert_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

class SegmentBatcher:
    """Collects tokenised transcript segments on the device and encodes them ``batch_segments`` at a time.

    K8's cost per 128-token segment falls from 54 us at 8 segments per call to 19 us at 128 (``tools/embed_batch_sweep.py``:
    at 8 x 128 tokens the ~44 launches of the encoder are latency bound, at 128 x 128 the GEMMs take the 128 x 128-tile
    path), and segment embedding is not latency critical - the reference design indexes a video's segments after its
    transcription task has finished (``.kiro/specs/semantic-video-search/tasks.md:297-302``) - so the ingest path hands the
    encoder whole batches.  ``add`` copies the rows into a device staging buffer (no host round trip) and returns the
    embeddings of every batch it completed; ``flush`` encodes what is left.
    """

    def __init__(self, encoder: "MiniLMEncoder", seq_len: int, batch_segments: int = 128, device=None):
        import torch

        self.enc, self.S, self.cap = encoder, int(seq_len), int(batch_segments)
        dev = device or torch.device("cuda", torch.cuda.current_device())
        self._ids = torch.zeros((self.cap, self.S), dtype=torch.int32, device=dev)
        self._mask = torch.zeros((self.cap, self.S), dtype=torch.uint8, device=dev)
        self.fill = 0
        self.encoded = 0  # segments encoded so far

    def add(self, ids, mask) -> list:
        """ids int32 / mask uint8 ``(b, S)`` CUDA tensors -> list of ``(n, hidden)`` embedding tensors (often empty)."""
        out = []
        b, at = int(ids.shape[0]), 0
        while at < b:
            n = min(self.cap - self.fill, b - at)
            self._ids[self.fill:self.fill + n].copy_(ids[at:at + n], non_blocking=True)
            self._mask[self.fill:self.fill + n].copy_(mask[at:at + n], non_blocking=True)
            self.fill += n
            at += n
            if self.fill == self.cap:
                out.extend(self.flush())
        return out

    def flush(self) -> list:
        if not self.fill:
            return []
        n, self.fill = self.fill, 0
        self.encoded += n
        return [self.enc.encode_ids(self._ids[:n], self._mask[:n])]
