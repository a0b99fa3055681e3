"""Model / token / generation configuration for the DualAR hot path.

Mirrors the reference's config surface so a reference checkpoint directory loads unchanged:

* ``RQTransformerModelArgs``  <- modeling/model/rq_transformer.py:25-114 (dataclass, derived
  defaults in ``__post_init__``) and its pydantic twin mlx_inference/.../lm/rq_transformer.py:10-48.
  Unknown JSON keys are ignored (the MLX twin ignores extras; training-only keys such as
  ``dropout`` / ``initializer_range`` / ``is_reward_model`` are accepted and kept).
* ``TokenConfig``             <- lm/rq_transformer.py:51-89.
* ``GenerationSettings``      <- lm/generate.py:12-16.
* ``ModelType``               <- lm/config.py:5-12.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field, fields
from pathlib import Path
from typing import Optional


def find_multiple(n: int, k: int) -> int:
    return n if n % k == 0 else n + k - (n % k)


@dataclass
class RQTransformerModelArgs:
    model_type: str = "dual_ar"

    vocab_size: int = 32000
    n_layer: int = 32
    n_head: int = 32
    dim: int = 4096
    intermediate_size: Optional[int] = 16_384
    n_local_heads: int = -1
    head_dim: int = 64
    rope_base: float = 10000
    norm_eps: float = 1e-5
    max_seq_len: int = 2048
    dropout: float = 0.0
    tie_word_embeddings: bool = True
    attention_qkv_bias: bool = False

    codebook_size: int = 160
    num_codebooks: int = 4

    use_gradient_checkpointing: bool = False
    initializer_range: float = 0.02
    is_reward_model: bool = False
    share_codebook_embeddings: bool = True
    scale_codebook_embeddings: bool = False

    fast_dim: Optional[int] = 1024
    n_fast_layer: int = 4
    fast_n_head: Optional[int] = 16
    fast_n_local_heads: Optional[int] = None
    fast_head_dim: Optional[int] = None
    fast_intermediate_size: Optional[int] = None
    fast_attention_qkv_bias: Optional[bool] = None
    depthwise_wte: Optional[bool] = False
    depthwise_output: Optional[bool] = False
    duplicate_code_0: Optional[bool] = True

    def __post_init__(self):
        if self.n_local_heads == -1:
            self.n_local_heads = self.n_head
        if self.intermediate_size is None:
            self.intermediate_size = find_multiple(int(2 * 4 * self.dim / 3), 256)
        self.head_dim = self.dim // self.n_head
        self.fast_dim = self.fast_dim or self.dim
        self.fast_n_head = self.fast_n_head or self.n_head
        self.fast_n_local_heads = self.fast_n_local_heads or self.n_local_heads
        self.fast_head_dim = self.fast_head_dim or self.head_dim
        self.fast_intermediate_size = self.fast_intermediate_size or self.intermediate_size
        self.fast_attention_qkv_bias = (
            self.fast_attention_qkv_bias
            if self.fast_attention_qkv_bias is not None
            else self.attention_qkv_bias
        )
        if self.duplicate_code_0 is None:
            self.duplicate_code_0 = True
        self.depthwise_wte = bool(self.depthwise_wte)
        self.depthwise_output = bool(self.depthwise_output)

    # ---- derived
    @property
    def max_fast_seqlen(self) -> int:
        return self.num_codebooks - (0 if self.duplicate_code_0 else 1)

    @property
    def grid_height(self) -> int:
        """Rows of the prompt / frame grid: 1 text row + code rows (9 when code 0 is duplicated)."""
        return 1 + self.max_fast_seqlen

    # ---- io
    @classmethod
    def from_dict(cls, data: dict) -> "RQTransformerModelArgs":
        known = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in data.items() if k in known})

    @classmethod
    def from_json_file(cls, path) -> "RQTransformerModelArgs":
        p = Path(path)
        if p.is_dir():
            p = p / "config.json"
        with open(p, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))

    from_pretrained = from_json_file

    def save(self, path) -> None:
        with open(path, "w") as f:
            json.dump(self.__dict__, f, indent=4, sort_keys=True, ensure_ascii=False)

    def validate_for_engine(self) -> None:
        """What the HIP engine supports; anything else fails loudly instead of mis-computing."""
        errs = []
        if self.head_dim != 64 or self.fast_head_dim != 64:
            errs.append("head_dim and fast_head_dim must be 64")
        if self.attention_qkv_bias or self.fast_attention_qkv_bias:
            errs.append("qkv bias is not supported")
        for name in ("dim", "fast_dim", "intermediate_size", "fast_intermediate_size"):
            if getattr(self, name) % 32:
                errs.append(f"{name} must be a multiple of 32")
        if self.n_head % self.n_local_heads or self.fast_n_head % self.fast_n_local_heads:
            errs.append("n_head must be a multiple of n_local_heads")
        if self.n_head * self.head_dim != self.dim or self.fast_n_head * self.fast_head_dim != self.fast_dim:
            errs.append("n_head * head_dim must equal dim")
        if self.codebook_size % 16 or self.vocab_size % 16:
            errs.append("codebook_size and vocab_size must be multiples of 16")
        if errs:
            raise ValueError("unsupported model config: " + "; ".join(errs))


@dataclass
class ModelType:
    family: str = "dual_ar"
    version: Optional[str] = None
    codec: str = "mimi"

    @classmethod
    def smoltts_v0(cls) -> "ModelType":
        return cls(family="dual_ar", version=None, codec="mimi")


@dataclass
class TokenConfig:
    im_end_id: int
    pad_id: int
    semantic_start_id: int
    semantic_end_id: Optional[int]

    @classmethod
    def from_tokenizer(cls, tokenizer, config: RQTransformerModelArgs) -> "TokenConfig":
        """dual_ar branch of lm/rq_transformer.py:58-89."""
        im_end = tokenizer.token_to_id("<|im_end|>")
        if im_end is None:
            raise ValueError("Tokenizer does not have <|im_end|>")
        start = tokenizer.token_to_id("<|semantic:0|>")
        end = tokenizer.token_to_id(f"<|semantic:{config.codebook_size - 1}|>")
        if start is None or end is None:
            raise ValueError("Tokenizer does not have the <|semantic:k|> range")
        pad = tokenizer.token_to_id("<|semantic|>") or 5
        return cls(im_end_id=im_end, pad_id=pad, semantic_start_id=start, semantic_end_id=end)


@dataclass
class GenerationSettings:
    """lm/generate.py:12-16. ``default_temp == 0.0`` / ``default_fast_temp`` None-or-<=0 => greedy."""

    default_temp: float = 0.7
    default_fast_temp: Optional[float] = 0.7
    min_p: Optional[float] = None
    max_new_tokens: int = 1024
    seed: Optional[int] = None  # extension: None draws a fresh seed per generator
    # What ``min_p`` does.  "reference": exactly what lm/utils/samplers.py:22-28 computes -- the threshold is built from the
    # token's OWN log-probability (``top_logprobs`` is the same array as ``sorted_logprobs``), so nothing is ever removed
    # and the draw is plain categorical over logits / temp.  "intended": the rule the code was taken from (MLX examples):
    # keep tokens with p >= min_p * p_max.  The drop-in default is what the reference does.
    min_p_mode: str = "reference"

    def __post_init__(self):
        if self.min_p_mode not in ("reference", "intended"):
            raise ValueError(f"min_p_mode must be 'reference' or 'intended', got {self.min_p_mode!r}")

    @property
    def effective_min_p(self) -> float:
        """The cut the device sampler applies (0 = none)."""
        return float(self.min_p or 0.0) if self.min_p_mode == "intended" else 0.0

    @classmethod
    def greedy(cls, max_new_tokens: int = 1024) -> "GenerationSettings":
        return cls(default_temp=0.0, default_fast_temp=0.0, min_p=None, max_new_tokens=max_new_tokens)

    @property
    def is_greedy(self) -> bool:
        fast_greedy = self.default_fast_temp is None or self.default_fast_temp <= 0
        return self.default_temp == 0.0 and fast_greedy


@dataclass
class NumericsMode:
    """Which half of the reference the arithmetic quirks follow (SURVEY.md §7 'quirks').

    ``embed_mask``: "torch" zeroes the codebook-embedding sum where code0 == 0
    (modeling/...:219); "mlx" zeroes it where the text-row token is not a semantic token
    (lm/rq_transformer.py:162-169).  ``rope_bf16``: cos/sin table rounded to bf16
    (modeling/...:624) or exact fp32 (MLX nn.RoPE)."""

    embed_mask: str = "torch"
    rope_bf16: bool = True

    @classmethod
    def torch_reference(cls) -> "NumericsMode":
        return cls("torch", True)

    @classmethod
    def mlx_reference(cls) -> "NumericsMode":
        return cls("mlx", False)
