"""The last inch of the drop-in surface: checkpoints as they exist on disk, not as this repo writes them.

* LM: every tensor stored in bf16 (what a bf16-trained ``model.safetensors`` holds, norms included), config.json with keys
  this package does not know, fused or split q/k/v, torch or MLX head layout -> ``load_checkpoint`` -> ``pack_lm`` must
  consume every key and produce the arena of the same values given in fp32.
* Mimi: the COMPLETE ``transformers.MimiModel`` key list (encoder, decoder, both transformers, both quantizers with their
  ``initialized`` / ``cluster_usage`` / ``embed_sum`` bookkeeping, down/upsample), written to ``model.safetensors`` ->
  ``load_mimi_state`` -> ``pack_mimi`` / ``pack_mimi_encoder`` must read every key or name the ones they ignore, with the reason
  (reference loader: mlx_inference/src/smoltts_mlx/codec/mimi.py:107-156, strict load of the same names).
* ``smoltts-server``: the console entry point of the reference (mlx_inference/pyproject.toml:21-22) exists under the same name."""
import json
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("name,split_qkv,mlx_head", [("tiny", False, True), ("tiny", True, False), ("tiny_proj", False, False), ("tiny_nodup", True, True)])
def test_bf16_stored_lm_checkpoint_is_consumed_whole(tmp_path, name, split_qkv, mlx_head):
    from safetensors.torch import save_file

    from smoltts_amd.checkpoint import load_checkpoint
    from smoltts_amd.config import NumericsMode
    from smoltts_amd.packing import pack_lm
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import ByteLevelTokenizer

    cfg = named_config(name)
    st = synthetic_lm_state(cfg, seed=2)
    disk = {}
    for k, v in st.items():
        if split_qkv and k.endswith("attention.wqkv.weight"):  # legacy checkpoints: wq / wk / wv (modeling/...:528-533)
            hq = (cfg.n_head if k.startswith("layers.") else cfg.fast_n_head) * 64
            hk = (cfg.n_local_heads if k.startswith("layers.") else cfg.fast_n_local_heads) * 64
            for n, part in zip(("wq", "wk", "wv"), torch.split(v, [hq, hk, hk])):
                disk[k.replace("wqkv", n)] = part
        else:
            disk[k] = v
    fo = disk["fast_output.weight"]
    if mlx_head and fo.dim() == 3:  # train/convert_safetensors.py:10-15
        disk["fast_output.weight"] = fo.permute(1, 0, 2).reshape(fo.shape[1], -1).T
    d = tmp_path / name
    d.mkdir()
    cfg.save(d / "config.json")
    cj = json.loads((d / "config.json").read_text())
    cj.update({"dropout": 0.0, "initializer_range": 0.02, "is_reward_model": False, "some_future_key": [1, 2]})
    (d / "config.json").write_text(json.dumps(cj))
    ByteLevelTokenizer().save(d / "tokenizer.json")
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in disk.items()}, str(d / "model.safetensors"))  # EVERY tensor bf16

    cfg2, _, st2 = load_checkpoint(d)
    assert all(v.dtype == torch.bfloat16 for v in st2.values()) and set(st2) == set(disk)
    rep = {}
    a2, o2 = pack_lm(cfg2, st2, NumericsMode.torch_reference(), report=rep)
    assert rep["unused"] == [], rep["unused"]
    # the same values handed over in fp32, in the original (fused, torch-layout) form
    a1, o1 = pack_lm(cfg, {k: v.to(torch.bfloat16).float() for k, v in st.items()}, NumericsMode.torch_reference())
    assert o1 == o2 and torch.equal(a1, a2)


def test_full_hf_mimi_key_list_is_consumed_or_named(tmp_path):
    transformers = pytest.importorskip("transformers")
    from safetensors.torch import save_file

    from smoltts_amd.checkpoint import load_mimi_state
    from smoltts_amd.packing import classify_mimi_keys, pack_mimi

    torch.manual_seed(0)
    model = transformers.MimiModel(transformers.MimiConfig())
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
    del model
    # the bookkeeping tensors a trained checkpoint carries
    assert any(k.endswith("codebook.initialized") for k in sd) and any(k.endswith("codebook.cluster_usage") for k in sd)
    assert any(k.startswith("encoder.layers.") for k in sd) and any(k.startswith("encoder_transformer.layers.") for k in sd)
    assert sum(1 for k in sd if re.fullmatch(r"quantizer\.acoustic_residual_vector_quantizer\.layers\.\d+\.codebook\.embed_sum", k)) == 31
    for k in sd:  # a non-trivial codebook state (fresh modules hold zeros / ones)
        if k.endswith("codebook.embed_sum"):
            sd[k] = torch.randn_like(sd[k])
        elif k.endswith("codebook.cluster_usage"):
            sd[k] = torch.rand_like(sd[k]) + 0.5
    save_file(sd, str(tmp_path / "model.safetensors"))
    st = load_mimi_state(tmp_path)  # a directory or the file itself
    assert set(st) == set(sd)

    c = classify_mimi_keys(st, num_codebooks=8)
    assert c["unknown"] == [], c["unknown"][:20]
    assert set(c["decoder"]) | set(c["encoder"]) | set(c["ignored"]) == set(sd)
    # what may be ignored: bookkeeping flags and the acoustic codebooks beyond the 7 the model emits -- nothing else
    for k in c["ignored"]:
        assert k.endswith("codebook.initialized") or int(k.split(".")[3]) >= 7, k
    assert "upsample.conv.weight" in c["decoder"] and "downsample.conv.weight" in c["encoder"]
    assert "quantizer.semantic_residual_vector_quantizer.layers.0.codebook.cluster_usage" in c["decoder"]
    assert "quantizer.acoustic_residual_vector_quantizer.layers.6.codebook.embed_sum" in c["decoder"]
    assert "quantizer.acoustic_residual_vector_quantizer.input_proj.weight" in c["encoder"]
    assert not any(k.startswith("encoder") for k in c["decoder"])

    # a bf16-stored copy of the decoder-side tensors packs to the arena of the same values in fp32
    dec16 = {k: st[k].to(torch.bfloat16) for k in c["decoder"]}
    a16, o16 = pack_mimi(dec16, 8, max_positions=64)
    a32, o32 = pack_mimi({k: v.float() for k, v in dec16.items()}, 8, max_positions=64)
    assert o16 == o32 and torch.equal(a16, a32)


def test_console_entry_point_of_the_reference():
    """`smoltts-server` (mlx_inference/pyproject.toml:21-22) -> smoltts_amd.server.app:main, same --config / --port flags."""
    try:
        import tomllib
    except ModuleNotFoundError:  # Python 3.10
        import tomli as tomllib
    meta = tomllib.loads((ROOT / "pyproject.toml").read_text())
    target = meta["project"]["scripts"]["smoltts-server"]
    assert target == "smoltts_amd.server.app:main"
    mod, fn = target.split(":")
    import importlib

    main = getattr(importlib.import_module(mod), fn)
    assert callable(main)
    src = (ROOT / "smoltts_amd" / "server" / "app.py").read_text()
    assert '"--config"' in src and '"--port"' in src and "default=8000" in src  # scripts/server.py:48-59
