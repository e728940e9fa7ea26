"""The search & index layer the reference designed but never built (design.md:1092-1133, tasks.md:297-325):
tokenizer, vector store + its .index file (CPU), and - on the GPU - the segment_embedding task and semantic search
end to end on the HIP encoder / index."""
import asyncio
import json

import numpy as np
import pytest

from eioku_amd import semantic, task_handler

WORDS = ["the", "a", "cat", "dog", "sat", "on", "mat", "run", "##ning", "##s", "play", "##ed", "video", "scene", "tokyo",
         "cafe", "resume", "hello", "world", "un", "##believ", "##able", "!", ",", ".", "?", "'", "s", "1", "2", "##3",
         "東", "京", "music", "guitar", "solo", "ocean", "waves", "beach", "sunset", "recipe", "pasta", "tomato", "sauce"]


@pytest.fixture(scope="module")
def vocab_file(tmp_path_factory):
    p = tmp_path_factory.mktemp("vocab") / "vocab.txt"
    p.write_text("\n".join(["[PAD]", "[unused0]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS) + "\n", encoding="utf-8")
    return p


TEXTS = ["The cat sat on the mat.", "Running dogs played!", "Café résumé, TOKYO 東京?", "unbelievable hello-world's 123",
         "", "x" * 150 + " cat", "the " * 400, "  spaced\tout\nvideo   scene  "]


def test_tokenizer_matches_transformers_bert_tokenizer(vocab_file):
    """transformers' BertTokenizer (Hugging Face's own assembly of normaliser, pre-tokeniser, WordPiece model and
    [CLS]/[SEP] template - what sentence-transformers loads for all-MiniLM-L6-v2) on the same vocabulary: ids identical,
    including accents, CJK, punctuation splits, [UNK] for over-long words, truncation at 256."""
    from transformers import BertTokenizer

    ref = BertTokenizer(str(vocab_file), do_lower_case=True)
    tk = semantic.WordPieceTokenizer(vocab_file)
    ids, mask = tk.encode_batch(TEXTS, max_seq_length=256)
    assert ids.dtype == np.int32 and mask.dtype == np.uint8 and ids.shape == mask.shape and ids.shape[1] == 256
    for i, t in enumerate(TEXTS):
        want = ref(t, truncation=True, max_length=256)["input_ids"]
        n = int(mask[i].sum())
        assert list(ids[i, :n]) == want, t
        assert not ids[i, n:].any() and not mask[i, n:].any()  # [PAD] = 0 beyond the sentence
    assert list(ids[4, :2]) == [3, 4]  # empty text: [CLS] [SEP]
    short, _ = tk.encode_batch(["cat dog"], 256)
    assert short.shape == (1, 4)  # padded to the batch's longest, not to 256
    with pytest.raises(ValueError, match="no \\[PAD\\]"):
        bad = vocab_file.parent / "bad.txt"
        bad.write_text("a\nb\n")
        semantic.WordPieceTokenizer(bad)


def test_vector_store_index_file_round_trip_and_delete(tmp_path):
    rng = np.random.default_rng(0)
    st = semantic.VectorStore(384)
    emb = rng.standard_normal((7, 384)).astype(np.float32)
    for i in range(7):
        st.index_segment(f"s{i}", emb[i], {"video_id": f"v{i % 3}", "start_time": i * 1.5, "end_time": i * 1.5 + 1, "text": f"t{i} é東"})
    path = tmp_path / "library.index"
    st.save(path)
    assert path.read_bytes().startswith(b"EIOKUIDX1\n")
    back = semantic.VectorStore.load(path)
    assert len(back) == 7 and np.array_equal(back.matrix(), emb)
    assert back._meta == st._meta and back._meta[3]["segment_id"] == "s3" and back._meta[3]["text"] == "t3 é東"
    assert back.delete_by_video_id("v1") and len(back) == 5 and not back.delete_by_video_id("v1")
    assert [m["segment_id"] for m in back._meta] == ["s0", "s2", "s3", "s5", "s6"]
    assert np.array_equal(back.matrix(), emb[[0, 2, 3, 5, 6]])
    with pytest.raises(ValueError, match="384-d"):
        st.index_segment("x", np.zeros(10), {})
    (tmp_path / "junk.index").write_bytes(b"not an index")
    with pytest.raises(ValueError, match="not an eioku"):
        semantic.VectorStore.load(tmp_path / "junk.index")
    assert semantic.VectorStore().search(np.zeros(384), 5) == []


SEGMENTS = [{"text": "the cat sat on the mat", "start": 0.0, "end": 2.5}, {"text": "guitar solo music", "start": 2.5, "end": 6.0},
            {"text": "ocean waves on the beach at sunset", "start_ms": 6000, "end_ms": 9500},
            {"text": "pasta recipe tomato sauce", "start": 9.5, "end": 14.0}]


@pytest.mark.gpu
def test_segment_embedding_task_and_semantic_search_end_to_end(gpu, vocab_file, tmp_path, monkeypatch):
    from eioku_amd import embed

    monkeypatch.setenv("MODEL_CACHE_DIR", str(tmp_path / "models"))  # the reference default, /models, is the container's
    from oracle import bert as obert

    cfg = dict(embed.MINILM_L6_V2, vocab=64)
    state = embed.random_state(cfg, 11)
    gen = semantic.EmbeddingGenerator(embed.MiniLMEncoder(state, cfg), semantic.WordPieceTokenizer(vocab_file))
    engine = semantic.SemanticSearchEngine(gen, semantic.VectorStore(384))
    sink = []
    ctx = {"artifact_sink": sink.extend, "search_engine": engine}
    out = asyncio.run(task_handler.process_ml_task(ctx, "t1", "segment_embedding", "vidA", "/videos/a.mp4", {"segments": SEGMENTS}))
    assert out == {"task_id": "t1", "status": "completed", "artifact_count": 4} and len(engine.store) == 4
    assert [e.artifact_type for e in sink] == ["segment.embedding"] * 4
    assert (sink[2].span_start_ms, sink[2].span_end_ms) == (6000, 9500) and (sink[0].span_start_ms, sink[0].span_end_ms) == (0, 2500)
    payload = json.loads(sink[1].payload_json)
    # the stored vector is the encoder's, which is the float64 oracle's to 1e-4 (BASELINE's bar)
    ids, mask = gen.tokenizer.encode_batch([SEGMENTS[1]["text"]])
    want = obert.encode(state, cfg, ids, mask)[0]
    assert np.abs(np.array(payload["embedding"]) - want).max() <= 1e-4 * np.abs(want).max() and payload["text"] == "guitar solo music"
    asyncio.run(task_handler.process_ml_task(ctx, "t2", "segment_embedding", "vidB", "/videos/b.mp4",
                                             {"segments": [{"text": "dog running on the beach", "start": 1.0, "end": 3.0}]}))
    asyncio.run(task_handler.process_ml_task(ctx, "t3", "segment_embedding", "vidA", "/videos/a.mp4", {"segments": SEGMENTS}))
    assert len(engine.store) == 5  # re-running a video replaces its vectors
    # property 13.5 "search result relevance": a segment's own text finds that segment first with cosine ~ 1
    for s in SEGMENTS:
        r = engine.search(s["text"], top_k=3)
        assert r[0].matched_text == s["text"] and r[0].video_id == "vidA" and abs(r[0].relevance_score - 1.0) < 1e-4
        assert r[0].relevance_score >= r[1].relevance_score >= r[2].relevance_score
    r = engine.search("ocean waves on the beach at sunset", filters={"video_id": "vidB"})
    assert [x.video_id for x in r] == ["vidB"] and r[0].start_time == 1.0 and r[0].end_time == 3.0
    # the library's single .index file: searching the reloaded store gives the same answer
    engine.store.save(tmp_path / "lib.index")
    engine2 = semantic.SemanticSearchEngine(gen, semantic.VectorStore.load(tmp_path / "lib.index"))
    assert engine2.search_dicts("guitar solo music", top_k=2) == engine.search_dicts("guitar solo music", top_k=2)
    with pytest.raises(RuntimeError, match="needs config"):
        asyncio.run(task_handler.process_ml_task(ctx, "t4", "segment_embedding", "vidC", "/videos/c.mp4", {}))


@pytest.mark.gpu
def test_vector_store_serves_large_top_k_and_narrow_filters(gpu):
    """top_k > 32 used to be clamped to 32 silently, and a video_id filter only saw the 32 global nearest
    (a video whose segments are not among them returned nothing)."""
    rng = np.random.default_rng(5)
    st = semantic.VectorStore(384)
    x = rng.standard_normal((300, 384)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = x[0] + 0.01 * rng.standard_normal(384).astype(np.float32)
    # video "far" owns the 60 rows FARTHEST from q: none of them is among the 32 global nearest
    order = np.argsort(((x - q) ** 2).sum(1))
    far = set(order[-60:].tolist())
    for i in range(300):
        st.index_segment(f"s{i}", x[i], {"video_id": "far" if i in far else "near", "row": i})
    got = st.search(q, top_k=100)
    assert len(got) == 100 and [m["row"] for _, m in got] == order[:100].tolist()
    assert all(a[0] <= b[0] for a, b in zip(got, got[1:]))
    only_far = st.search(q, top_k=40, filters={"video_id": "far"})
    assert len(only_far) == 40 and [m["row"] for _, m in only_far] == order[-60:-20].tolist()
    assert len(st.search(q, top_k=500, filters={"video_id": ["far"]})) == 60  # all the video has
    with pytest.raises(ValueError):
        st.search(q, top_k=0)
