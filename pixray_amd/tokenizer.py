"""Byte-pair tokenizer for `CLIP_Base.encode_text` (/root/reference/slip.py:68-70 calls `clip.tokenize(text)`).

`clip.tokenize` lives in openai/CLIP (`clip/simple_tokenizer.py`, `clip/clip.py`; un-vendored, requirements.txt:29)
[UPSTREAM]; this is a from-the-published-algorithm implementation of the same scheme:

  * text -> html-unescape twice, collapse whitespace, lower-case (the upstream additionally runs `ftfy.fix_text`, which is
    not installed here; it is the identity on well-formed text);
  * split with the CLIP pattern (special tokens | English contractions | letter runs | single digits | other runs);
  * every piece is mapped byte-by-byte onto printable code points, the last symbol gets the `</w>` marker, and adjacent
    pairs are merged greedily in the order of the merges table;
  * ids: 256 byte symbols, 256 byte symbols + `</w>`, one id per merge, `<|startoftext|>`, `<|endoftext|>` (49408 ids for
    the 48894 merges the OpenAI checkpoints use);
  * `tokenize`: `[SOT] + ids + [EOT]`, zero padded to the context length (77); longer inputs raise unless `truncate`.

The merges table (`bpe_simple_vocab_16e6.txt.gz`) ships with the openai/CLIP package and is not available offline: pass
its path (or set PIXRAY_CLIP_BPE).  Tests build small tables and cross-check against HF's independent `CLIPTokenizer`.
"""
import gzip
import html
import os
from functools import lru_cache
from typing import Dict, List, Sequence, Tuple, Union

import torch

try:                                    # \p{L} / \p{N} classes need the third-party `regex` module (present in the image)
    import regex as _re
    _PATTERN = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                           _re.IGNORECASE)
except ImportError:                     # pragma: no cover
    import re as _re
    _PATTERN = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[^\W\d_]+|\d|[^\s\w]+", _re.IGNORECASE)

SOT, EOT = "<|startoftext|>", "<|endoftext|>"
N_MERGES_OPENAI = 49152 - 256 - 2        # rows of the merges file the OpenAI vocabulary uses


@lru_cache()
def byte_symbols() -> Dict[int, str]:
    """a printable, whitespace-free code point for each of the 256 byte values (printable latin-1 bytes map to
    themselves, the rest to 256, 257, ...)"""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _ordered_byte_symbols() -> List[str]:
    # vocabulary order: the kept bytes first (in the order above), then the remapped ones in byte order
    sym = byte_symbols()
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    rest = [b for b in range(256) if b not in set(keep)]
    return [sym[b] for b in keep + rest]


def read_merges(path: str, limit: int = N_MERGES_OPENAI) -> List[Tuple[str, str]]:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt", encoding="utf-8") as f:
        lines = f.read().split("\n")
    rows = [tuple(l.split()) for l in lines[1:1 + limit]]          # first line is a header
    return [r for r in rows if len(r) == 2]


class BpeTokenizer:
    def __init__(self, merges: Union[str, Sequence[Tuple[str, str]]]):
        if isinstance(merges, str):
            merges = read_merges(merges)
        self.merges = [tuple(m) for m in merges]
        base = _ordered_byte_symbols()
        vocab = base + [s + "</w>" for s in base] + ["".join(m) for m in self.merges] + [SOT, EOT]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(self.merges)}
        self.cache = {SOT: SOT, EOT: EOT}
        self.sot_token, self.eot_token = self.encoder[SOT], self.encoder[EOT]

    @property
    def vocab_size(self) -> int:
        return len(self.encoder)

    def _bpe(self, piece: str) -> List[str]:
        hit = self.cache.get(piece)
        if hit is not None:
            return hit.split(" ")
        word = list(piece[:-1]) + [piece[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for a, b in zip(word[:-1], word[1:]):
                r = self.rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                    merged.append(best[0] + best[1]); i += 2
                else:
                    merged.append(word[i]); i += 1
            word = merged
        self.cache[piece] = " ".join(word)
        return word

    @staticmethod
    def clean(text: str) -> str:
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split()).strip().lower()

    def encode(self, text: str) -> List[int]:
        sym = byte_symbols()
        ids: List[int] = []
        for piece in _PATTERN.findall(self.clean(text)):
            mapped = "".join(sym[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._bpe(mapped))
        return ids

    def decode(self, ids: Sequence[int]) -> str:
        inv = {v: k for k, v in byte_symbols().items()}
        raw = bytearray()
        for i in ids:
            tok = self.decoder[int(i)]
            if tok in (SOT, EOT):
                raw += tok.encode()
                continue
            end = tok.endswith("</w>")
            raw += bytes(inv[c] for c in (tok[:-4] if end else tok))
            if end:
                raw += b" "
        return raw.decode("utf-8", errors="replace")

    def tokenize(self, texts: Union[str, Sequence[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        """`clip.tokenize`: int tensor [n, context_length]"""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.int32)
        for i, t in enumerate(texts):
            ids = [self.sot_token] + self.encode(t) + [self.eot_token]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = self.eot_token
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
        return out


_default = None


def default_tokenizer(path: str = None) -> BpeTokenizer:
    """the process-wide tokenizer (upstream keeps one `_tokenizer` too); needs the OpenAI merges file"""
    global _default
    if _default is None or path is not None:
        path = path or os.environ.get("PIXRAY_CLIP_BPE")
        if not path or not os.path.exists(path):
            raise FileNotFoundError(
                "CLIP's BPE merges table (bpe_simple_vocab_16e6.txt.gz, shipped inside the openai/CLIP package) is needed to "
                "tokenize text; pass its path or set PIXRAY_CLIP_BPE.  Without it, pass token ids or precomputed embeddings.")
        _default = BpeTokenizer(path)
    return _default
