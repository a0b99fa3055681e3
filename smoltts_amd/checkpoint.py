"""Checkpoint directory I/O with the reference's two on-disk surfaces.

* MLX side  (mlx_inference/src/smoltts_mlx/__init__.py:36-49): ``config.json`` + ``tokenizer.json``
  + ``model.safetensors`` (``fast_output.weight`` flattened to (n*2048, d) by
  train/convert_safetensors.py:10-15).
* torch side (modeling/model/rq_transformer.py:292-329): ``config.json`` + tokenizer files +
  ``model.pth`` (``torch.save(state_dict)``; keys may carry ``_orig_mod.`` prefixes).

``load_checkpoint`` accepts either; the packer (packing.pack_lm) accepts both weight layouts.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Tuple

import torch

from .config import RQTransformerModelArgs
from .tokenizer import ByteLevelTokenizer, load_tokenizer


def load_state(checkpoint_dir) -> Dict[str, torch.Tensor]:
    d = Path(checkpoint_dir)
    st = d / "model.safetensors"
    if st.exists():
        from safetensors.torch import load_file

        return load_file(str(st))
    pth = d / "model.pth"
    if pth.exists():
        return torch.load(pth, map_location="cpu", mmap=True, weights_only=True)
    raise FileNotFoundError(f"{d} holds neither model.safetensors nor model.pth")


def load_checkpoint(checkpoint_dir) -> Tuple[RQTransformerModelArgs, ByteLevelTokenizer, Dict[str, torch.Tensor]]:
    d = Path(checkpoint_dir)
    cfg = RQTransformerModelArgs.from_json_file(d / "config.json")
    tok = load_tokenizer(d, cfg.codebook_size)
    return cfg, tok, load_state(d)


def save_checkpoint(checkpoint_dir, cfg: RQTransformerModelArgs, state: Dict[str, torch.Tensor], fmt: str = "safetensors",
                    mlx_layout: bool = True) -> Path:
    """Write a reference-shaped checkpoint directory (tests, synthetic models)."""
    d = Path(checkpoint_dir)
    d.mkdir(parents=True, exist_ok=True)
    cfg.save(d / "config.json")
    ByteLevelTokenizer().save(d / "tokenizer.json")
    out = {k: v.contiguous() for k, v in state.items()}
    if fmt == "safetensors":
        if mlx_layout and cfg.depthwise_output and out["fast_output.weight"].dim() == 3:
            w = out["fast_output.weight"]  # (n, d, cs) -> (n*cs, d), train/convert_safetensors.py:10-15
            out["fast_output.weight"] = w.permute(1, 0, 2).reshape(w.shape[1], -1).T.contiguous()
        from safetensors.torch import save_file

        save_file({k: v.to(torch.bfloat16) if v.dim() >= 2 else v for k, v in out.items()}, str(d / "model.safetensors"))
    elif fmt == "pth":
        torch.save(out, d / "model.pth")
    else:
        raise ValueError(fmt)
    return d


def load_mimi_state(path) -> Dict[str, torch.Tensor]:
    """Hugging Face ``kyutai/mimi`` ``model.safetensors`` (decoder-side keys are all that is used)."""
    from safetensors.torch import load_file

    p = Path(path)
    if p.is_dir():
        p = p / "model.safetensors"
    return load_file(str(p))


def convert_training_checkpoint(pt_path, out_path="model.safetensors") -> Path:
    """Trainer ``.pt`` -> ``model.safetensors`` in the MLX layout (train/convert_safetensors.py:7-17).

    Same contract as the reference tool — reads ``["model_state_dict"]``, strips ``_orig_mod.``, flattens a
    3-D depthwise head ``(n, d, cs)`` to ``(n*cs, d)`` — except that ``d`` is read from the tensor instead
    of being fixed at 768, so 70m (d=576) and ``fast_dim != dim`` checkpoints convert too.  A bare state
    dict (``model.pth``) is accepted as well.  dtypes are kept."""
    from safetensors.torch import save_file

    data = torch.load(pt_path, map_location="cpu", weights_only=True)
    state = data["model_state_dict"] if isinstance(data, dict) and "model_state_dict" in data else data
    state = {k.replace("_orig_mod.", ""): v for k, v in state.items()}
    w = state.get("fast_output.weight")
    if w is not None and w.dim() == 3:
        state["fast_output.weight"] = w.permute(1, 0, 2).reshape(w.shape[1], -1).T
    out = Path(out_path)
    if out.is_dir():
        out = out / "model.safetensors"
    save_file({k: v.contiguous() for k, v in state.items()}, str(out))
    return out


if __name__ == "__main__":
    import sys

    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m smoltts_amd.checkpoint CHECKPOINT.pt [OUT.safetensors]")
    print(convert_training_checkpoint(*sys.argv[1:3]))
