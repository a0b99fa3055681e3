"""Checkpoint tooling: trainer .pt -> safetensors for any width, variant configs pack."""
import torch


def test_convert_training_checkpoint_any_width(tmp_path):
    from safetensors.torch import load_file

    from smoltts_amd.checkpoint import convert_training_checkpoint, load_checkpoint, save_checkpoint
    from smoltts_amd.config import NumericsMode
    from smoltts_amd.packing import pack_lm
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    for name in ("tiny", "tiny_nodup", "tiny_proj"):  # d = 384 / 384 / fast 128: none of them the 768 the reference tool assumes
        cfg = named_config(name)
        st = synthetic_lm_state(cfg, seed=1)
        d = tmp_path / name
        d.mkdir()
        torch.save({"model_state_dict": {"_orig_mod." + k: v for k, v in st.items()}, "step": 7}, d / "ckpt.pt")
        out = convert_training_checkpoint(d / "ckpt.pt", d / "model.safetensors")
        conv = load_file(str(out))
        assert set(conv) == set(st)
        if cfg.depthwise_output:
            w = st["fast_output.weight"]
            assert conv["fast_output.weight"].shape == (cfg.max_fast_seqlen * cfg.codebook_size, cfg.fast_dim)
            for i in (0, cfg.max_fast_seqlen - 1):  # rows [2048 i, 2048 (i+1)) == W[i]^T  (lm/rq_transformer.py:211-217)
                assert torch.equal(conv["fast_output.weight"][2048 * i: 2048 * (i + 1)], w[i].T)
        # the converted directory loads through the ordinary path and packs to the same arena as the source state
        save_checkpoint(d, cfg, st, fmt="pth")  # writes config.json + tokenizer.json (+ model.pth)
        (d / "model.pth").unlink()
        cfg2, _, st2 = load_checkpoint(d)
        a1, o1 = pack_lm(cfg, st, NumericsMode.torch_reference())
        a2, o2 = pack_lm(cfg2, st2, NumericsMode.torch_reference())
        assert torch.equal(a1, a2) and o1 == o2
