"""Host-logic tests of the BPE tokenizer behind `encode_text` (slip.py:68 `clip.tokenize`): the published algorithm's
invariants, and agreement with an independent implementation (HF `CLIPTokenizer`) on a shared merges table.  The OpenAI
merges file itself is not available offline."""
import json
import os
from collections import Counter

import pytest
import torch

from pixray_amd.tokenizer import BpeTokenizer, byte_symbols, default_tokenizer

CORPUS = ("a photo of a cat . a painting of the quick brown fox jumps over the lazy dog ! don 't stop , it 's 42 cats "
          "café naïve über 日本語 the the the of of a a a trending on artstation unreal engine")


def train_merges(corpus, n):
    """a small deterministic BPE training run (most frequent pair, ties by symbol order) to get a merges table"""
    sym = byte_symbols()
    words = []
    for w in corpus.split():
        m = "".join(sym[b] for b in w.encode())
        words.append(list(m[:-1]) + [m[-1] + "</w>"])
    merges = []
    for _ in range(n):
        c = Counter()
        for w in words:
            for a, b in zip(w[:-1], w[1:]):
                c[(a, b)] += 1
        if not c:
            break
        (a, b), _ = sorted(c.items(), key=lambda kv: (-kv[1], kv[0]))[0]
        merges.append((a, b))
        nw = []
        for w in words:
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and w[i] == a and w[i + 1] == b:
                    out.append(a + b); i += 2
                else:
                    out.append(w[i]); i += 1
            nw.append(out)
        words = nw
    return merges


@pytest.fixture(scope="module")
def tok():
    return BpeTokenizer(train_merges(CORPUS, 120))


def test_vocabulary_layout(tok):
    sym = byte_symbols()
    assert len(sym) == 256 and len(set(sym.values())) == 256 and not any(c.isspace() for c in sym.values())
    assert sym[ord("a")] == "a" and sym[ord("!")] == "!" and sym[ord(" ")] != " "
    assert tok.vocab_size == 512 + len(tok.merges) + 2
    assert tok.sot_token == tok.vocab_size - 2 and tok.eot_token == tok.vocab_size - 1     # EOT is the largest id (argmax pooling)
    # with the 48894 merges of the OpenAI table this layout gives CLIP's 49408 ids
    assert 512 + 48894 + 2 == 49408


def test_tokenize_shape_padding_and_errors(tok):
    t = tok.tokenize(["a cat", "the quick brown fox"])
    assert t.shape == (2, 77) and t.dtype == torch.int32
    for row in t:
        n = int((row != 0).sum())
        assert row[0] == tok.sot_token and row[n - 1] == tok.eot_token and (row[n:] == 0).all()
        assert int(row.argmax()) == n - 1
    assert tok.tokenize("a cat").shape == (1, 77)
    with pytest.raises(RuntimeError):
        tok.tokenize("x " * 100)
    tr = tok.tokenize("x " * 100, truncate=True)
    assert tr[0, -1] == tok.eot_token and tr[0, 0] == tok.sot_token


def test_cleaning_lowercase_and_roundtrip(tok):
    assert tok.encode("A  Photo\tof a CAT") == tok.encode("a photo of a cat")
    assert tok.encode("cats &amp;amp; dogs") == tok.encode("cats & dogs")            # html-unescaped twice, as upstream
    s = "don't stop, café 42 日本語"
    assert tok.decode(tok.encode(s)).replace(" ", "") == s.replace(" ", "")
    # greedy lowest-rank merging: the most frequent corpus words became single tokens
    assert len(tok.encode("the")) == 1 and len(tok.encode("of")) == 1
    # digits are split one by one by the pattern
    assert len(tok.encode("42")) == 2


def test_matches_independent_implementation(tok, tmp_path):
    transformers = pytest.importorskip("transformers")
    from transformers import CLIPTokenizer
    vf, mf = tmp_path / "vocab.json", tmp_path / "merges.txt"
    vf.write_text(json.dumps(tok.encoder))
    mf.write_text("#version: 0.2\n" + "\n".join(a + " " + b for a, b in tok.merges) + "\n")
    hf = CLIPTokenizer(str(vf), str(mf))
    for t in ["a photo of a cat", "A Painting of the QUICK brown fox!", "don't stop, it's 42 cats", "café naïve über",
              "the   lazy\tdog...", "日本語 cat", "x" * 30, "it's the cat's 7 lives!!! ok?", "trending on artstation | unreal engine"]:
        mine = tok.tokenize(t, 77, truncate=True)[0].tolist()
        theirs = hf(t, padding="max_length", max_length=77, truncation=True)["input_ids"]
        n = mine.index(tok.eot_token) + 1
        assert mine[:n] == theirs[:n], t


def test_merges_file_reader_and_default(tmp_path, monkeypatch, tok):
    import gzip
    p = tmp_path / "bpe.txt.gz"
    with gzip.open(p, "wt", encoding="utf-8") as f:
        f.write('"bpe_simple_vocab_16e6.txt#version: 0.2\n' + "\n".join(a + " " + b for a, b in tok.merges) + "\n")
    t2 = BpeTokenizer(str(p))
    assert t2.merges == tok.merges and t2.encode("a photo of a cat") == tok.encode("a photo of a cat")
    monkeypatch.delenv("PIXRAY_CLIP_BPE", raising=False)
    import pixray_amd.tokenizer as T
    monkeypatch.setattr(T, "_default", None)
    with pytest.raises(FileNotFoundError):
        default_tokenizer()
    assert default_tokenizer(str(p)).vocab_size == tok.vocab_size
