"""The server's settings file, same schema and validation as the reference (server/settings.py:12-63): exactly one of
``model_id`` / ``checkpoint_dir``, a ``generation`` block (lm/generate.py:12-16) and a ``model_type`` block
(lm/config.py:5-12).  Extensions of this build: ``mimi_checkpoint`` (the reference downloads kyutai/mimi; there is no
network here), ``max_batch`` (slots per GPU), ``weight_format`` ("bf16" | "fp8").  ``model_id`` is accepted by the schema
but cannot be resolved without network access; ``get_checkpoint_dir`` says so."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Literal, Optional

from pydantic import BaseModel, Field, model_validator

from ..config import GenerationSettings as _GenerationSettings


class ModelType(BaseModel):
    family: Literal["fish", "dual_ar"] = "dual_ar"
    version: Optional[Literal["1.5", "1.4", "1.2"]] = None
    codec: Literal["mimi", "1.4", "1.2"] = "mimi"


class GenerationBlock(BaseModel):
    default_temp: float = 0.5
    default_fast_temp: Optional[float] = 0.0
    min_p: Optional[float] = 0.10
    max_new_tokens: int = Field(default=1024, ge=1)
    # extension: "reference" = what lm/utils/samplers.py:22-28 computes (its threshold never removes a token, so the draw is
    # categorical over logits / temp); "intended" = keep p >= min_p * p_max.  The drop-in default is the reference's behaviour.
    min_p_mode: Literal["reference", "intended"] = "reference"

    def to_settings(self) -> _GenerationSettings:
        return _GenerationSettings(default_temp=self.default_temp, default_fast_temp=self.default_fast_temp, min_p=self.min_p,
                                   max_new_tokens=self.max_new_tokens, min_p_mode=self.min_p_mode)


class ServerSettings(BaseModel):
    model_id: Optional[str] = None
    checkpoint_dir: Optional[str] = None
    generation: GenerationBlock = Field(default_factory=GenerationBlock)
    model_type: ModelType = Field(default_factory=ModelType)
    mimi_checkpoint: Optional[str] = None
    max_batch: int = Field(default=32, ge=1, le=256)
    weight_format: Literal["bf16", "fp8"] = "bf16"
    # (not in the reference's settings) bf16x3 products per operand pair in the codec's matrix-core kernels: 6 = fp32-grade (the
    # reference's codec runs fp32), 3 = the 2^-16-grade form: chunks 23 % faster, PCM RMS error 7e-7 (SMOLTTS_MIMI_OPT_PRODUCTS)
    codec_products: Literal[3, 6] = 6

    model_config = {"protected_namespaces": ()}

    @model_validator(mode="after")
    def validate_model_source(self):
        if self.model_id is not None and self.checkpoint_dir is not None:
            raise ValueError("Cannot specify both model_id and checkpoint_dir")
        if self.model_id is None and self.checkpoint_dir is None:
            raise ValueError("Must specify either model_id or checkpoint_dir")
        if self.model_type.family != "dual_ar" or self.model_type.codec != "mimi":
            raise ValueError("this build serves the dual_ar family with the mimi codec only")
        return self

    @classmethod
    def get_settings(cls, config_path: Optional[str]) -> "ServerSettings":
        if not config_path:
            raise ValueError("pass --config: the default settings name a Hugging Face model_id, which needs network access")
        return cls(**json.loads(Path(config_path).read_text()))

    def get_checkpoint_dir(self) -> Path:
        if self.checkpoint_dir is None:
            raise ValueError(f"model_id={self.model_id!r} would be downloaded from the Hugging Face hub; there is no network access: "
                             "set checkpoint_dir")
        return Path(self.checkpoint_dir)
