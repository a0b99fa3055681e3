"""Seeded synthetic checkpoints at real shapes (no trained weights are obtainable offline).

``synthetic_lm_state`` restates the reference's initialisation rule so that random checkpoints
have the reference's statistics: ``normal_(0, initializer_range)`` for every Linear / Embedding
(modeling/model/rq_transformer.py:262-271, applied by ``RQTransformer.__init__`` :399), ones for
RMSNorm weights (:605) and ``kaiming_uniform_(a=sqrt(5))`` for the depthwise head
(``DepthwiseLinear`` :589-590 => U(-1/sqrt(fan_in), 1/sqrt(fan_in)), fan_in = dim * codebook_size).
Keys and shapes are the reference torch ``state_dict`` ones, so the same dict can be written as
``model.pth`` and loaded by either side.

numpy's PCG64 stream is used (not torch's) because it is bit-reproducible across hosts, which
lets committed golden vectors be regenerated on the GPU box from the seed alone.
Matrices and embeddings are rounded to bf16 (the checkpoint dtype of the bf16 configs in
BASELINE.json) and returned as fp32 tensors holding bf16-representable values.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict

import numpy as np
import torch

from .config import RQTransformerModelArgs


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def synthetic_lm_state(cfg: RQTransformerModelArgs, seed: int = 0, bf16: bool = True) -> Dict[str, torch.Tensor]:
    rng = np.random.Generator(np.random.PCG64(seed))
    std = cfg.initializer_range

    def normal(*shape):
        t = torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * np.float32(std))
        return _bf16_round(t) if bf16 else t

    def block(prefix, dim, n_head, n_kv, hd, inter, out):
        out[prefix + "attention.wqkv.weight"] = normal((n_head + 2 * n_kv) * hd, dim)
        out[prefix + "attention.wo.weight"] = normal(dim, dim)
        out[prefix + "feed_forward.w1.weight"] = normal(inter, dim)
        out[prefix + "feed_forward.w3.weight"] = normal(inter, dim)
        out[prefix + "feed_forward.w2.weight"] = normal(dim, inter)
        out[prefix + "ffn_norm.weight"] = torch.ones(dim)
        out[prefix + "attention_norm.weight"] = torch.ones(dim)

    st: Dict[str, torch.Tensor] = {}
    st["embeddings.weight"] = normal(cfg.vocab_size, cfg.dim)
    st["codebook_embeddings.weight"] = normal(cfg.codebook_size * cfg.num_codebooks, cfg.dim)
    for i in range(cfg.n_layer):
        block(f"layers.{i}.", cfg.dim, cfg.n_head, cfg.n_local_heads, cfg.head_dim, cfg.intermediate_size, st)
    st["norm.weight"] = torch.ones(cfg.dim)
    if not cfg.tie_word_embeddings:
        st["output.weight"] = normal(cfg.vocab_size, cfg.dim)
    if cfg.fast_dim != cfg.dim:
        st["fast_project_in.weight"] = normal(cfg.fast_dim, cfg.dim)
        st["fast_project_in.bias"] = normal(cfg.fast_dim)  # the reference zero-inits it; non-zero exercises the bias path
    n_fast_emb = cfg.codebook_size * (cfg.num_codebooks - 1) if cfg.depthwise_wte else cfg.codebook_size
    st["fast_embeddings.weight"] = normal(n_fast_emb, cfg.fast_dim)
    for i in range(cfg.n_fast_layer):
        block(
            f"fast_layers.{i}.", cfg.fast_dim, cfg.fast_n_head, cfg.fast_n_local_heads,
            cfg.fast_head_dim, cfg.fast_intermediate_size, st,
        )
    st["fast_norm.weight"] = torch.ones(cfg.fast_dim)
    if cfg.depthwise_output:
        bound = 1.0 / math.sqrt(cfg.fast_dim * cfg.codebook_size)
        w = rng.uniform(-bound, bound, size=(cfg.max_fast_seqlen, cfg.fast_dim, cfg.codebook_size)).astype(np.float32)
        w = torch.from_numpy(w)
        st["fast_output.weight"] = _bf16_round(w) if bf16 else w
    else:
        st["fast_output.weight"] = normal(cfg.codebook_size, cfg.fast_dim)
    return st


def state_fingerprint(state: Dict[str, torch.Tensor]) -> float:
    """Cheap order-independent checksum used by golden fixtures to detect RNG-stream drift."""
    acc = 0.0
    for k in sorted(state):
        t = state[k].double()
        acc += float(t.sum()) + 1e-3 * float(t.abs().sum())
    return acc


def tiny_config() -> RQTransformerModelArgs:
    """2+2-layer config with every structural feature of the real ones (GQA 3:1, depthwise in/out)."""
    return RQTransformerModelArgs(
        vocab_size=2368, n_layer=2, n_head=6, n_local_heads=2, dim=384, intermediate_size=512,
        rope_base=100000, norm_eps=1e-5, max_seq_len=512, codebook_size=2048, num_codebooks=8,
        fast_dim=384, n_fast_layer=2, fast_n_head=6, fast_n_local_heads=2, fast_head_dim=64,
        fast_intermediate_size=512, depthwise_wte=True, depthwise_output=True,
        initializer_range=0.0416666, tie_word_embeddings=True,
    )


_SIZES = {
    "smoltts_byte_70m": dict(dim=576, n_head=9, n_local_heads=3, intermediate_size=1536),
    "smoltts_byte_150m": dict(dim=768, n_head=12, n_local_heads=4, intermediate_size=3072),
}


def named_config(name: str) -> RQTransformerModelArgs:
    """The two shapes of sample_model_sizes/smoltts_byte_{70m,150m}.json (values restated here so
    the GPU box, which has no reference tree, can build them)."""
    if name == "tiny":
        return tiny_config()
    if name == "tiny_nodup":  # code 0 carried by the slow token only: 7 depth steps, 8-row grid (modeling :344-360)
        return dataclasses.replace(tiny_config(), duplicate_code_0=False)
    if name == "tiny_proj":   # fast_dim != dim (fast_project_in Linear+bias :339-342), untied slow head, plain Linear depth head
        return dataclasses.replace(tiny_config(), fast_dim=128, fast_n_head=2, fast_n_local_heads=1, fast_intermediate_size=256,
                                   tie_word_embeddings=False, depthwise_output=False)
    s = _SIZES[name]
    return RQTransformerModelArgs(
        attention_qkv_bias=False, codebook_size=2048, dim=s["dim"], dropout=0.1,
        fast_attention_qkv_bias=False, fast_dim=s["dim"], fast_head_dim=64,
        fast_intermediate_size=s["intermediate_size"], fast_n_head=s["n_head"],
        fast_n_local_heads=s["n_local_heads"], head_dim=64, initializer_range=0.041666666666666664,
        intermediate_size=s["intermediate_size"], is_reward_model=False, max_seq_len=2048,
        model_type="dual_ar", n_fast_layer=4, n_head=s["n_head"], n_layer=10,
        n_local_heads=s["n_local_heads"], depthwise_wte=True, depthwise_output=True, norm_eps=1e-5,
        num_codebooks=8, rope_base=100000, scale_codebook_embeddings=False,
        share_codebook_embeddings=True, tie_word_embeddings=True, use_gradient_checkpointing=True,
        vocab_size=2368,
    )
