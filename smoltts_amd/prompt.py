"""ChatML prompt grid construction.

Mirrors ``PromptEncoder`` (mlx_inference/.../lm/utils/prompt.py:10-63) and ``SmolTTS._get_prompt``
(mlx_inference/.../__init__.py:120-151): the prompt is an int grid ``(1 + depth, T)`` whose row 0
holds tokenizer ids and whose code rows are zero for text turns.  At inference the three turns are
``<|im_start|>system\\n<|speaker:v|><|im_end|>``, ``<|im_start|>user\\n{text}<|im_end|>`` and
``<|im_start|>assistant\\n`` with *no* newline after ``<|im_end|>`` (unlike the training template).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

VOICES = ["heart", "bella", "nova", "sky", "sarah", "michael", "fenrir", "liam", "emma", "isabella", "fable"]
VOICE_MAP = {k: v for v, k in enumerate(VOICES)}


class PromptEncoder:
    def __init__(self, tokenizer, semantic_offset: int, num_codebooks: int = 8, duplicate_code_0: bool = True):
        self.tokenizer = tokenizer
        self.depth = num_codebooks if duplicate_code_0 else num_codebooks - 1
        self.semantic_offset = semantic_offset

    @classmethod
    def from_config(cls, tokenizer, config, token_config) -> "PromptEncoder":
        dc0 = config.duplicate_code_0 if config.duplicate_code_0 is not None else True
        return cls(tokenizer, token_config.semantic_start_id, config.num_codebooks, dc0)

    def tokenize_text(self, text: str) -> np.ndarray:
        ids = self.tokenizer.encode(text, add_special_tokens=True).ids
        grid = np.zeros((1 + self.depth, len(ids)), dtype=np.int32)
        grid[0] = ids
        return grid

    def encode_text_turn(self, role: str, content: Optional[str] = None) -> np.ndarray:
        suffix = f"{content}<|im_end|>" if content is not None else ""
        return self.tokenize_text(f"<|im_start|>{role}\n{suffix}")

    def encode_vq(self, codes: np.ndarray) -> np.ndarray:
        """codes (n_codebooks>=depth, T) of one utterance -> grid incl. the closing <|im_end|>\\n."""
        if codes.ndim != 2:
            raise ValueError("Must be single batch")
        semantic = (codes[0:1, :] + self.semantic_offset).astype(np.int32)
        lower = codes[codes.shape[0] - self.depth :, :].astype(np.int32)
        block = np.concatenate([semantic, lower], axis=0)
        return np.concatenate([block, self.tokenize_text("<|im_end|>\n")], axis=1)

    def build_prompt(self, text: str, voice: str = "heart", sysprompt: Optional[np.ndarray] = None) -> np.ndarray:
        """``SmolTTS._get_prompt``: unknown voice names fall back to id 0. Returns ``(1+depth, T)``."""
        voice_id = VOICE_MAP.get(voice, 0)
        if sysprompt is None:
            sysprompt = self.encode_text_turn("system", f"<|speaker:{voice_id}|>")
        user = self.encode_text_turn("user", text)
        assistant = self.encode_text_turn("assistant")
        return np.concatenate([sysprompt, user, assistant], axis=1)
