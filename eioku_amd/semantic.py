"""Search & index layer of the semantic-video-search design, on the HIP encoder and index.

The reference specifies this layer but never built it (``.kiro/specs/semantic-video-search/design.md:1092-1133``:
Embedding Generator / Vector Store / Semantic Search Engine; ``tasks.md:297-325``: task 13 unchecked).  The three
classes below carry the interfaces that design names - ``generateEmbedding / generateBatchEmbeddings``,
``indexSegment / search / deleteByVideoId`` + "save/load index from file" (``tasks.md:306``), ``search(query, filters)
-> SearchResult`` - with Python spelling, over K8 (all-MiniLM-L6-v2 on ``libeioku_hip``) and K9 (flat L2 kNN; on unit
vectors L2^2 = 2 - 2 cos, so the ranking IS the cosine ranking the design asks for and ``relevance_score`` is the cosine).

Tokenisation stays on the host: WordPiece over the model's own ``vocab.txt`` with BERT's uncased normalisation, built
from the ``tokenizers`` library (no network: the vocabulary file ships with the checkpoint directory).
"""
from __future__ import annotations

import json
import struct
from dataclasses import asdict, dataclass
from pathlib import Path

import numpy as np

MAGIC = b"EIOKUIDX1\n"


class WordPieceTokenizer:
    """``[CLS] wordpieces [SEP]`` ids + attention mask, as sentence-transformers feeds all-MiniLM-L6-v2
    (``BertTokenizer``: lower-case, strip accents, split punctuation / CJK, greedy longest-match WordPiece,
    ``[UNK]`` for words over 100 characters; truncation at ``max_seq_length`` 256 word pieces incl. the specials)."""

    def __init__(self, vocab_path: str | Path, lowercase: bool = True):
        from tokenizers import Tokenizer, normalizers, pre_tokenizers, processors
        from tokenizers.models import WordPiece

        vocab = {}
        with open(vocab_path, encoding="utf-8") as f:
            for i, line in enumerate(f):
                vocab[line.rstrip("\n")] = i
        for tok in ("[PAD]", "[UNK]", "[CLS]", "[SEP]"):
            if tok not in vocab:
                raise ValueError(f"{vocab_path}: vocabulary has no {tok}")
        self.vocab = vocab
        self.pad_id = vocab["[PAD]"]
        tk = Tokenizer(WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
        tk.normalizer = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None,
                                                   lowercase=lowercase)
        tk.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
        tk.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", vocab["[CLS]"]),
                                                                                                 ("[SEP]", vocab["[SEP]"])])
        self._tk = tk

    def encode_batch(self, texts: list[str], max_seq_length: int = 256):
        """-> (ids int32 (B,S), mask uint8 (B,S)), padded to the longest sequence of the batch."""
        self._tk.enable_truncation(max_length=max_seq_length)
        enc = self._tk.encode_batch(list(texts))
        S = max((len(e.ids) for e in enc), default=2)
        ids = np.full((len(enc), S), self.pad_id, np.int32)
        mask = np.zeros((len(enc), S), np.uint8)
        for i, e in enumerate(enc):
            ids[i, :len(e.ids)] = e.ids
            mask[i, :len(e.ids)] = 1
        return ids, mask


class EmbeddingGenerator:
    """design.md 2.1: text -> 384-d unit vector (K8).  ``encoder``: a loaded :class:`eioku_amd.embed.MiniLMEncoder`."""

    def __init__(self, encoder, tokenizer: WordPieceTokenizer, max_seq_length: int = 256, batch_size: int = 512):
        self.encoder, self.tokenizer = encoder, tokenizer
        self.max_seq_length, self.batch_size = max_seq_length, batch_size

    @classmethod
    def from_directory(cls, model_dir: str | Path, **kw):
        """``model_dir``: the all-MiniLM-L6-v2 snapshot (``model.safetensors`` / ``pytorch_model.bin`` + ``vocab.txt``)."""
        from . import embed

        model_dir = Path(model_dir)
        return cls(embed.MiniLMEncoder(embed.load_state(model_dir, embed.MINILM_L6_V2)), WordPieceTokenizer(model_dir / "vocab.txt"), **kw)

    def generate_batch_embeddings(self, texts: list[str]) -> np.ndarray:
        out = []
        for lo in range(0, len(texts), self.batch_size):
            ids, mask = self.tokenizer.encode_batch(texts[lo:lo + self.batch_size], self.max_seq_length)
            out.append(np.asarray(self.encoder.encode_ids(ids, mask)))
        H = self.encoder.cfg["hidden"]
        return np.concatenate(out) if out else np.zeros((0, H), np.float32)

    def generate_embedding(self, text: str) -> np.ndarray:
        return self.generate_batch_embeddings([text])[0]


@dataclass
class SearchResult:
    """design.md 2.3 output format."""

    video_id: str
    segment_id: str
    start_time: float
    end_time: float
    relevance_score: float
    matched_text: str
    thumbnail_path: str | None = None


class VectorStore:
    """design.md 2.2 + "single .index file per library" (design.md:37): embeddings and segment metadata.

    The authoritative copy lives on the host (fp32 rows + metadata, what the file holds); the HIP index over it is
    (re)built lazily before a search, so indexing / deleting / persistence need no GPU."""

    def __init__(self, d: int = 384):
        self.d = d
        self._rows: list[np.ndarray] = []
        self._meta: list[dict] = []
        self._index = None

    def __len__(self) -> int:
        return len(self._meta)

    def index_segment(self, segment_id: str, embedding, metadata: dict) -> bool:
        e = np.asarray(embedding, dtype=np.float32).reshape(-1)
        if e.shape[0] != self.d:
            raise ValueError(f"expected a {self.d}-d embedding, got {e.shape[0]}")
        self._rows.append(e)
        self._meta.append(dict(metadata, segment_id=segment_id))
        self._drop_index()
        return True

    def index_segments(self, segment_ids: list[str], embeddings, metadata: list[dict]) -> int:
        for s, e, m in zip(segment_ids, np.asarray(embeddings, dtype=np.float32), metadata):
            self.index_segment(s, e, m)
        return len(segment_ids)

    def delete_by_video_id(self, video_id: str) -> bool:
        keep = [i for i, m in enumerate(self._meta) if m.get("video_id") != video_id]
        removed = len(keep) != len(self._meta)
        self._rows = [self._rows[i] for i in keep]
        self._meta = [self._meta[i] for i in keep]
        self._drop_index()
        return removed

    def _drop_index(self):
        if self._index is not None:
            self._index.close()
            self._index = None

    def matrix(self) -> np.ndarray:
        return np.stack(self._rows).astype(np.float32) if self._rows else np.zeros((0, self.d), np.float32)

    def search(self, query_embedding, top_k: int = 10, filters: dict | None = None) -> list[tuple[float, dict]]:
        """``[(squared L2 distance, metadata)]`` ascending, up to ``top_k`` entries (any ``top_k``: rounds of 32 chained by
        ``IndexFlatL2.search_many``).  ``filters``: ``{"video_id": id or [ids]}``; the fetch grows until ``top_k`` rows of
        the wanted videos are found or the index is exhausted, so a narrow filter still returns what the videos hold."""
        if top_k < 1:
            raise ValueError(f"top_k must be >= 1, got {top_k}")
        if not self._meta:
            return []
        if self._index is None:
            from .search import IndexFlatL2

            self._index = IndexFlatL2(self.d)
            self._index.add(self.matrix())
        q = np.asarray(query_embedding, dtype=np.float32).reshape(1, self.d)
        allowed = None
        if filters and filters.get("video_id") is not None:
            v = filters["video_id"]
            allowed = set(v) if isinstance(v, (list, tuple, set)) else {v}
        n = len(self._meta)
        fetch = min(n, top_k if allowed is None else max(32, 4 * top_k))
        while True:
            D, I = self._index.search_many(q, fetch)
            out = []
            for dist, i in zip(D[0], I[0]):
                if i < 0:
                    break
                m = self._meta[int(i)]
                if allowed is not None and m.get("video_id") not in allowed:
                    continue
                out.append((float(dist), m))
                if len(out) == top_k:
                    break
            if len(out) == top_k or fetch >= n:
                return out
            fetch = min(n, fetch * 4)

    # ---- the .index file -----------------------------------------------------------------------------
    def save(self, path: str | Path) -> None:
        """``EIOKUIDX1\\n`` | u64 n | u32 d | u64 len(meta json) | n*d fp32 rows | metadata JSON (UTF-8)."""
        meta = json.dumps(self._meta, ensure_ascii=False).encode("utf-8")
        x = self.matrix()
        with open(path, "wb") as f:
            f.write(MAGIC)
            f.write(struct.pack("<QIQ", x.shape[0], self.d, len(meta)))
            f.write(np.ascontiguousarray(x).tobytes())
            f.write(meta)

    @classmethod
    def load(cls, path: str | Path) -> "VectorStore":
        with open(path, "rb") as f:
            if f.read(len(MAGIC)) != MAGIC:
                raise ValueError(f"{path}: not an eioku .index file")
            n, d, mlen = struct.unpack("<QIQ", f.read(20))
            x = np.frombuffer(f.read(n * d * 4), dtype=np.float32).reshape(n, d)
            meta = json.loads(f.read(mlen).decode("utf-8"))
        if len(meta) != n:
            raise ValueError(f"{path}: {n} rows but {len(meta)} metadata records")
        st = cls(d)
        st._rows = [r.copy() for r in x]
        st._meta = meta
        return st


class SemanticSearchEngine:
    """design.md 2.3: query text -> ranked SearchResults."""

    def __init__(self, generator: EmbeddingGenerator, store: VectorStore):
        self.generator, self.store = generator, store

    def index_transcript(self, video_id: str, segments: list[dict]) -> int:
        """``segments``: what ``transcribe_video`` returns (``model_manager.py:409-467``): dicts with ``text`` and
        ``start`` / ``end`` seconds (or ``start_ms`` / ``end_ms``)."""
        texts = [s["text"] for s in segments]
        emb = self.generator.generate_batch_embeddings(texts)
        meta = []
        for i, s in enumerate(segments):
            start = s["start_ms"] / 1000.0 if "start_ms" in s else float(s.get("start", 0.0))
            end = s["end_ms"] / 1000.0 if "end_ms" in s else float(s.get("end", start))
            meta.append({"video_id": video_id, "start_time": start, "end_time": end, "text": s["text"],
                         "thumbnail_path": s.get("thumbnail_path")})
        return self.store.index_segments([f"{video_id}_seg{i}" for i in range(len(segments))], emb, meta)

    def search(self, query: str, filters: dict | None = None, top_k: int = 10) -> list[SearchResult]:
        q = self.generator.generate_embedding(query)
        return [SearchResult(m["video_id"], m["segment_id"], m["start_time"], m["end_time"], 1.0 - dist / 2.0, m["text"],
                             m.get("thumbnail_path")) for dist, m in self.store.search(q, top_k, filters)]

    def search_dicts(self, query: str, filters: dict | None = None, top_k: int = 10) -> list[dict]:
        return [asdict(r) for r in self.search(query, filters, top_k)]
