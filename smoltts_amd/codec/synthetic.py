"""Seeded synthetic Mimi *decoder-side* weights with the Hugging Face ``kyutai/mimi`` key names and
torch layouts (the contract ``load_mimi`` reads, mlx_inference/.../codec/mimi.py:107-156; shapes in
SURVEY.md §8c).  ``kyutai/mimi`` itself is not obtainable offline.

Conv / linear weights are U(-1/sqrt(fan_in), 1/sqrt(fan_in)); LayerNorm weights 1 + noise; layer
scales around the checkpoint's initial 0.01 scaled up so the transformer visibly contributes;
codebooks N(0, 1) with positive random ``cluster_usage`` so the ``embed_sum / usage`` division is
exercised.  numpy PCG64 => bit-reproducible on every host.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

RATIOS = (8, 6, 5, 4)


def synthetic_mimi_state(seed: int = 0, num_codebooks: int = 8, n_layers: int = 8) -> Dict[str, torch.Tensor]:
    rng = np.random.Generator(np.random.PCG64(seed))

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return torch.from_numpy(rng.uniform(-b, b, size=shape).astype(np.float32))

    def nrm(shape, std=1.0, mean=0.0):
        return torch.from_numpy((rng.standard_normal(shape, dtype=np.float32) * np.float32(std) + np.float32(mean)))

    st: Dict[str, torch.Tensor] = {}
    for grp, n in (("semantic", 1), ("acoustic", num_codebooks - 1)):
        p = f"quantizer.{grp}_residual_vector_quantizer."
        for i in range(n):
            usage = torch.from_numpy(rng.uniform(0.5, 4.0, size=(2048,)).astype(np.float32))
            st[p + f"layers.{i}.codebook.embed_sum"] = nrm((2048, 256)) * usage[:, None]
            st[p + f"layers.{i}.codebook.cluster_usage"] = usage
            st[p + f"layers.{i}.codebook.initialized"] = torch.ones(1)
        st[p + "output_proj.weight"] = uni((512, 256, 1), 256)
    st["upsample.conv.weight"] = uni((512, 1, 4), 2)
    for li in range(n_layers):
        p = f"decoder_transformer.layers.{li}."
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            st[p + f"self_attn.{nm}.weight"] = uni((512, 512), 512)
        st[p + "mlp.fc1.weight"] = uni((2048, 512), 512)
        st[p + "mlp.fc2.weight"] = uni((512, 2048), 2048)
        for nm in ("input_layernorm", "post_attention_layernorm"):
            st[p + nm + ".weight"] = nrm((512,), 0.05, 1.0)
            st[p + nm + ".bias"] = nrm((512,), 0.05)
        st[p + "self_attn_layer_scale.scale"] = nrm((512,), 0.02, 0.2)
        st[p + "mlp_layer_scale.scale"] = nrm((512,), 0.02, 0.2)

    def conv(key, cout, cin, k):
        st[f"decoder.layers.{key}.conv.weight"] = uni((cout, cin, k), cin * k)
        st[f"decoder.layers.{key}.conv.bias"] = uni((cout,), cin * k)

    def convtr(key, cin, cout, k):
        st[f"decoder.layers.{key}.conv.weight"] = uni((cin, cout, k), cin * k / 2)
        st[f"decoder.layers.{key}.conv.bias"] = uni((cout,), cin * k)

    conv("0", 1024, 512, 7)
    ch, li = 1024, 1
    for r in RATIOS:
        convtr(str(li + 1), ch, ch // 2, 2 * r)
        conv(f"{li + 2}.block.1", ch // 4, ch // 2, 3)
        conv(f"{li + 2}.block.3", ch // 2, ch // 4, 1)
        ch //= 2
        li += 3
    conv("14", 1, 64, 3)
    # lift the output level to speech-like amplitude so that absolute PCM tolerances are meaningful
    st["decoder.layers.14.conv.weight"] = st["decoder.layers.14.conv.weight"] * 6.0
    return st


def synthetic_mimi_encoder_state(seed: int = 0, n_layers: int = 8) -> Dict[str, torch.Tensor]:
    """Encoder-side keys (SEANet encoder, encoder transformer, downsample, the RVQ ``input_proj``s) for the
    voice-clone path; merge with ``synthetic_mimi_state`` (which holds the codebooks).  Own RNG stream, so
    the decoder-side fixtures do not move."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return torch.from_numpy(rng.uniform(-b, b, size=shape).astype(np.float32))

    def nrm(shape, std=1.0, mean=0.0):
        return torch.from_numpy((rng.standard_normal(shape, dtype=np.float32) * np.float32(std) + np.float32(mean)))

    st: Dict[str, torch.Tensor] = {}

    def conv(key, cout, cin, k, gain=1.0):
        st[f"encoder.layers.{key}.conv.weight"] = uni((cout, cin, k), cin * k) * gain
        st[f"encoder.layers.{key}.conv.bias"] = uni((cout,), cin * k)

    conv("0", 64, 1, 7)
    ch, li = 64, 1
    for r in reversed(RATIOS):
        conv(f"{li}.block.1", ch // 2, ch, 3)
        conv(f"{li}.block.3", ch, ch // 2, 1)
        conv(str(li + 2), 2 * ch, ch, 2 * r, gain=2.0)  # keep the signal from fading through the ELUs
        ch *= 2
        li += 3
    conv("14", 512, 1024, 3, gain=2.0)
    for l in range(n_layers):
        p = f"encoder_transformer.layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            st[p + f"self_attn.{nm}.weight"] = uni((512, 512), 512)
        st[p + "mlp.fc1.weight"] = uni((2048, 512), 512)
        st[p + "mlp.fc2.weight"] = uni((512, 2048), 2048)
        for nm in ("input_layernorm", "post_attention_layernorm"):
            st[p + nm + ".weight"] = nrm((512,), 0.05, 1.0)
            st[p + nm + ".bias"] = nrm((512,), 0.05)
        st[p + "self_attn_layer_scale.scale"] = nrm((512,), 0.02, 0.2)
        st[p + "mlp_layer_scale.scale"] = nrm((512,), 0.02, 0.2)
    st["downsample.conv.weight"] = uni((512, 512, 4), 512 * 4) * 2.0
    for grp in ("semantic", "acoustic"):
        # latents of O(1) against N(0,1) codebooks: scale the projection so residuals are codebook-sized
        st[f"quantizer.{grp}_residual_vector_quantizer.input_proj.weight"] = uni((256, 512, 1), 512) * 24.0
    return st


def synthetic_pcm(n_samples: int, seed: int = 0) -> np.ndarray:
    """Speech-like test signal in [-1, 1]: a few drifting harmonics under a slow envelope, plus noise."""
    rng = np.random.Generator(np.random.PCG64(seed + 104729))
    t = np.arange(n_samples, dtype=np.float64) / 24000.0
    x = np.zeros(n_samples)
    for _ in range(5):
        f0, a, ph = rng.uniform(90, 1800), rng.uniform(0.05, 0.25), rng.uniform(0, 2 * np.pi)
        x += a * np.sin(2 * np.pi * f0 * t * (1 + 0.05 * np.sin(2 * np.pi * rng.uniform(0.5, 3) * t)) + ph)
    x *= 0.6 + 0.4 * np.sin(2 * np.pi * 2.5 * t + rng.uniform(0, 6))
    x += 0.02 * rng.standard_normal(n_samples)
    return np.clip(x, -1, 1).astype(np.float32)
