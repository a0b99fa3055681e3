"""Byte-level ChatML tokenizer of the smoltts checkpoints.

The reference builds its tokenizer with data_pipeline/scripts/create_bytelevel_init.py:15-57: a
merge-less BPE whose base vocabulary is the 256 latin-1 code points (id == code point), followed
by added special tokens in this order: 15 control tokens (``system``, ``user``, ``assistant``,
..., ``<|im_start|>``, ``<|im_end|>``), 49 ``<|speaker:k|>`` tokens and ``codebook_size``
``<|semantic:k|>`` tokens.  No normalizer, no pre-tokenizer; characters above U+00FF have no
vocabulary entry and are dropped (``unk_token`` is None).  The inference side loads it with
``tokenizers.Tokenizer.from_file`` (mlx_inference/.../__init__.py:42) and calls
``encode(text, add_special_tokens=True).ids`` (lm/utils/prompt.py:40-46).

``ByteLevelTokenizer`` is a dependency-free implementation of exactly that behaviour (added
tokens are matched leftmost-longest before byte fallback, as the ``tokenizers`` added-vocabulary
matcher does); ``load_tokenizer`` prefers a checkpoint's ``tokenizer.json`` when present.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List, Optional

CONTROL_TOKENS = [
    "system", "user", "assistant", "<|british|>", "<|american|>", "<|male|>", "<|female|>",
    "<|unknown|>", "<|endoftext|>", "<|voice|>", "<|semantic|>", "<|pad|>", "<|epad|>",
    "<|im_start|>", "<|im_end|>",
]


def default_added_tokens(codebook_size: int = 2048) -> List[str]:
    speakers = [f"<|speaker:{i}|>" for i in range(64 - len(CONTROL_TOKENS))]
    semantic = [f"<|semantic:{i}|>" for i in range(codebook_size)]
    return [*CONTROL_TOKENS, *speakers, *semantic]


class _Encoding:
    def __init__(self, ids: List[int]):
        self.ids = ids


class ByteLevelTokenizer:
    def __init__(self, added_tokens: Optional[List[str]] = None, base_vocab: Optional[Dict[str, int]] = None):
        self.base = base_vocab if base_vocab is not None else {chr(i): i for i in range(256)}
        added = added_tokens if added_tokens is not None else default_added_tokens()
        start = max(self.base.values()) + 1
        self.added: Dict[str, int] = {}
        for i, t in enumerate(added):
            self.added.setdefault(t, start + i)
        self._by_first: Dict[str, List[str]] = {}
        for t in self.added:
            self._by_first.setdefault(t[0], []).append(t)
        for lst in self._by_first.values():
            lst.sort(key=len, reverse=True)
        self._id_to_tok = {v: k for k, v in {**self.base, **self.added}.items()}

    # -- tokenizers.Tokenizer-compatible surface used by the reference
    def token_to_id(self, token: str) -> Optional[int]:
        if token in self.added:
            return self.added[token]
        return self.base.get(token)

    def id_to_token(self, idx: int) -> Optional[str]:
        return self._id_to_tok.get(idx)

    def get_vocab_size(self) -> int:
        return len(self.base) + len(self.added)

    def encode(self, text: str, add_special_tokens: bool = True) -> _Encoding:
        ids: List[int] = []
        i, n = 0, len(text)
        while i < n:
            hit = None
            for cand in self._by_first.get(text[i], ()):
                if text.startswith(cand, i):
                    hit = cand
                    break
            if hit is not None:
                ids.append(self.added[hit])
                i += len(hit)
                continue
            tid = self.base.get(text[i])
            if tid is not None:  # characters outside latin-1 are dropped (no unk token)
                ids.append(tid)
            i += 1
        return _Encoding(ids)

    @classmethod
    def from_file(cls, path) -> "ByteLevelTokenizer":
        """Parse a HF ``tokenizer.json`` written by the reference script (BPE, no merges)."""
        d = json.loads(Path(path).read_text(encoding="utf-8"))
        model = d["model"]
        if model.get("type") != "BPE" or model.get("merges"):
            raise ValueError("only the merge-less byte-level BPE tokenizer of smoltts is supported")
        added = sorted(d.get("added_tokens", []), key=lambda t: t["id"])
        tok = cls(added_tokens=[], base_vocab=dict(model["vocab"]))
        tok.added = {t["content"]: t["id"] for t in added}
        tok._by_first = {}
        for t in tok.added:
            tok._by_first.setdefault(t[0], []).append(t)
        for lst in tok._by_first.values():
            lst.sort(key=len, reverse=True)
        tok._id_to_tok = {v: k for k, v in {**tok.base, **tok.added}.items()}
        return tok

    def to_tokenizer_json(self) -> dict:
        """A ``tokenizer.json`` that ``tokenizers.Tokenizer.from_file`` loads to the same mapping."""
        return {
            "version": "1.0", "truncation": None, "padding": None,
            "added_tokens": [
                {"id": i, "content": t, "single_word": False, "lstrip": False, "rstrip": False,
                 "normalized": False, "special": True}
                for t, i in sorted(self.added.items(), key=lambda kv: kv[1])
            ],
            "normalizer": None, "pre_tokenizer": None, "post_processor": None,
            "decoder": {"type": "ByteLevel", "add_prefix_space": True, "trim_offsets": True, "use_regex": True},
            "model": {
                "type": "BPE", "dropout": None, "unk_token": None, "continuing_subword_prefix": None,
                "end_of_word_suffix": None, "fuse_unk": False, "byte_fallback": False,
                "ignore_merges": False, "vocab": self.base, "merges": [],
            },
        }

    def save(self, path) -> None:
        Path(path).write_text(json.dumps(self.to_tokenizer_json(), ensure_ascii=False), encoding="utf-8")


def load_tokenizer(checkpoint_dir=None, codebook_size: int = 2048):
    """``tokenizer.json`` of the checkpoint when present, else the default byte-level layout."""
    if checkpoint_dir is not None:
        p = Path(checkpoint_dir) / "tokenizer.json"
        if p.exists():
            return ByteLevelTokenizer.from_file(p)
    return ByteLevelTokenizer(default_added_tokens(codebook_size))
