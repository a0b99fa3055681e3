"""CPU suite: host-side logic (config, tokenizer, prompt, packing), the C-ABI surface, and the
guarantee that the product never reaches into oracle/."""
import json
import re
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]


def test_config_mirrors_reference_surface(tmp_path):
    from smoltts_amd.config import RQTransformerModelArgs
    from smoltts_amd.synthetic import named_config

    cfg = named_config("smoltts_byte_150m")
    assert (cfg.dim, cfg.n_head, cfg.n_local_heads, cfg.head_dim, cfg.intermediate_size) == (768, 12, 4, 64, 3072)
    assert cfg.max_fast_seqlen == 8 and cfg.grid_height == 9 and cfg.duplicate_code_0 is True
    cfg.validate_for_engine()
    d = dict(cfg.__dict__, some_future_key=1, dropout=0.1)
    p = tmp_path / "config.json"
    p.write_text(json.dumps(d))
    again = RQTransformerModelArgs.from_json_file(tmp_path)  # directory or file, unknown keys ignored
    assert again.dim == 768 and again.fast_n_local_heads == 4
    minimal = RQTransformerModelArgs.from_dict(dict(vocab_size=2368, n_layer=2, n_head=6, n_local_heads=2, dim=384,
                                                    intermediate_size=512, codebook_size=2048, num_codebooks=8, fast_dim=None,
                                                    fast_n_head=None, duplicate_code_0=False))
    assert minimal.fast_dim == 384 and minimal.fast_n_head == 6 and minimal.max_fast_seqlen == 7 and minimal.grid_height == 8
    with pytest.raises(ValueError):
        RQTransformerModelArgs.from_dict(dict(dim=100, n_head=2, n_layer=1)).validate_for_engine()


def test_tokenizer_matches_reference_ids(golden_dir, tmp_path):
    from smoltts_amd.tokenizer import ByteLevelTokenizer, load_tokenizer

    g = json.loads((golden_dir / "tokenizer_golden.json").read_text())
    tok = load_tokenizer()
    assert tok.get_vocab_size() == g["vocab_size"] == 2368
    for s, ids in zip(g["samples"], g["ids"]):
        assert tok.encode(s).ids == ids
    for t, i in g["special"].items():
        assert tok.token_to_id(t) == i
    tok.save(tmp_path / "tokenizer.json")
    again = ByteLevelTokenizer.from_file(tmp_path / "tokenizer.json")
    assert again.encode("<|im_start|>user\nhé<|im_end|>").ids == tok.encode("<|im_start|>user\nhé<|im_end|>").ids
    tokenizers = pytest.importorskip("tokenizers")
    hf = tokenizers.Tokenizer.from_file(str(tmp_path / "tokenizer.json"))
    for s in g["samples"] + ["", "plain ascii", "你好 dropped", "<|semantic:7|><|speaker:48|>"]:
        assert hf.encode(s, add_special_tokens=True).ids == tok.encode(s).ids


def test_prompt_grid_layout():
    from smoltts_amd.prompt import PromptEncoder
    from smoltts_amd.tokenizer import load_tokenizer

    pe = PromptEncoder(load_tokenizer(), 320)
    g = pe.build_prompt("Hello world!", "heart")
    assert g.shape == (9, 24) and g.dtype == np.int32 and not g[1:].any()
    assert g[0, :5].tolist() == [269, 256, 10, 271, 270]  # <|im_start|>system\n<|speaker:0|><|im_end|>
    assert g[0, -3:].tolist() == [269, 258, 10]           # <|im_start|>assistant\n
    assert pe.build_prompt("x", "sky")[0, 3] == 271 + 3 and pe.build_prompt("x", "nobody")[0, 3] == 271
    codes = np.arange(16, dtype=np.int32).reshape(8, 2)
    vq = pe.encode_vq(codes)
    assert vq.shape == (9, 4) and vq[0, :2].tolist() == [320, 321] and vq[1:, :2].tolist() == codes.tolist()
    assert vq[0, 2:].tolist() == [270, 10]
    pe7 = PromptEncoder(load_tokenizer(), 320, duplicate_code_0=False)
    assert pe7.build_prompt("x").shape[0] == 8


def test_weight_tiles_roundtrip_and_layout():
    from smoltts_amd.packing import tile_t16x32, untile_t16x32

    w = torch.randn(40, 96)
    for dt in (torch.bfloat16, torch.float32):
        flat = tile_t16x32(w, dt)
        assert flat.numel() == 48 * 96
        assert torch.equal(untile_t16x32(flat, 40, 96), w.to(dt))
    # lane l = 16*q + r of tile (nt, kc) owns w[16*nt + r][32*kc + 8*q : +8]
    flat = tile_t16x32(w, torch.bfloat16).view(3, 3, 64, 8)
    assert torch.equal(flat[1, 2, 16 * 3 + 5], w[16 + 5, 64 + 24: 64 + 32].to(torch.bfloat16))
    with pytest.raises(ValueError):
        tile_t16x32(torch.zeros(16, 40), torch.bfloat16)


def test_conv_as_gemm_equals_torch_convs():
    from smoltts_amd.packing import conv_as_gemm

    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 5, generator=g)  # B, C, T
    w, b = torch.randn(6, 8, 3, generator=g), torch.randn(6, generator=g)
    gw, gb = conv_as_gemm(w, b, False, 1)
    xp = F.pad(x, (2, 0)).transpose(1, 2)  # B, T+2, C (2 halo rows)
    rows = torch.stack([xp[:, t: t + 3].reshape(2, -1) for t in range(5)], dim=1)  # window of k rows ending at t
    assert torch.allclose(rows @ gw.T + gb, F.conv1d(F.pad(x, (2, 0)), w, b).transpose(1, 2), atol=1e-5)
    wt, bt = torch.randn(8, 4, 6, generator=g), torch.randn(4, generator=g)  # cin, cout, k = 2*3
    gw, gb = conv_as_gemm(wt, bt, True, 3)
    xp = F.pad(x, (1, 0)).transpose(1, 2)
    rows = torch.stack([xp[:, t: t + 2].reshape(2, -1) for t in range(5)], dim=1)
    y = F.conv_transpose1d(x, wt, bt, stride=3)[..., :15].transpose(1, 2)
    assert torch.allclose((rows @ gw.T + gb).reshape(2, 15, 4), y, atol=1e-5)


def test_pack_lm_accepts_both_checkpoint_layouts():
    from smoltts_amd.config import NumericsMode
    from smoltts_amd.packing import pack_lm, rope_table
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from oracle.lm_oracle import rope_table as oracle_rope

    cfg = named_config("tiny")
    st = synthetic_lm_state(cfg, seed=0)
    arena, off = pack_lm(cfg, st, NumericsMode.torch_reference())
    assert off["fast_head_step_stride"] == 2048 and arena.numel() % 256 == 0
    # MLX layout: flattened fast_output, split wq/wk/wv, _orig_mod. prefixes
    st2 = {}
    for k, v in st.items():
        if k.endswith("attention.wqkv.weight"):
            q, kk, vv = v.split([cfg.n_head * 64, cfg.n_local_heads * 64, cfg.n_local_heads * 64])
            base = k[: -len("wqkv.weight")]
            st2["_orig_mod." + base + "wq.weight"], st2["_orig_mod." + base + "wk.weight"], st2["_orig_mod." + base + "wv.weight"] = q, kk, vv
        elif k == "fast_output.weight":
            st2[k] = v.permute(1, 0, 2).reshape(cfg.fast_dim, -1).T.contiguous()  # train/convert_safetensors.py:10-15
        else:
            st2["_orig_mod." + k] = v
    arena2, off2 = pack_lm(cfg, st2, NumericsMode.torch_reference())
    assert torch.equal(arena, arena2) and off == off2
    assert torch.equal(rope_table(32, 64, 1e5, True), oracle_rope(32, 64, 1e5, True))
    a3, _ = pack_lm(cfg, st, NumericsMode.mlx_reference())
    assert not torch.equal(arena, a3)  # exact RoPE table differs from the bf16-rounded one


def _declared_functions():
    text = (ROOT / "include" / "smoltts_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smoltts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    """No compute calls here: dlopen + symbol lookup only (works without a GPU)."""
    from smoltts_amd import engine

    if not engine.LIB_PATH.exists():
        from smoltts_amd.build import build_library

        build_library()
    lib = engine.load_library()
    declared = _declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/smoltts_hip.h but not exported"
    for name in engine.exported_symbols():
        assert name in declared, f"{name} is bound by engine.py but not declared in the header"
    assert lib.smoltts_abi_version() == 6
    assert isinstance(lib.smoltts_last_error(), bytes)


def test_product_library_has_no_debug_state_and_fixed_flags():
    """The event / cycle-stamp hooks exist only in -DSMOLTTS_DEBUG_HOOKS variants (separate directory); the product
    library's objects carry the flag set they were built with, and extra flags are refused for the product path."""
    import subprocess

    from smoltts_amd import build, engine

    build.build_library()
    syms = subprocess.run(["nm", "-D", str(engine.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    for banned in ("smoltts_debug_set_stamps", "smoltts_profile_begin", "smoltts_profile_end", "smoltts_debug_duplicate"):
        assert banned not in syms, f"{banned} is exported by the product library"
    stamp = (build.CSRC / "build" / "flags.txt").read_text()
    assert stamp.splitlines()[1] == " ".join(build.PRODUCT_FLAGS) and "-D" not in stamp
    with pytest.raises(ValueError):
        build.build_library(extra_flags=["-DSMOLTTS_DBG_PIECES=1"])      # variants only
    assert build.variant_path("hooks").parent.parent.name == "variants" and build.variant_path("hooks") != engine.LIB_PATH


def test_product_never_imports_the_oracle_and_fails_without_library(tmp_path):
    for p in (ROOT / "smoltts_amd").rglob("*.py"):
        src = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{p} imports the oracle"
        assert "oracle." not in src.replace("oracle/", ""), f"{p} references the oracle"
    from smoltts_amd import engine

    with pytest.raises(engine.SmolttsError):
        engine.load_library(tmp_path / "missing.so")
    if not torch.cuda.is_available():
        from smoltts_amd.config import TokenConfig
        from smoltts_amd.synthetic import named_config, synthetic_lm_state

        cfg = named_config("tiny")
        with pytest.raises(engine.SmolttsError):  # no GPU => loud failure, never a CPU fallback
            engine.LMEngine(cfg, synthetic_lm_state(cfg), TokenConfig(270, 266, 320, 2367))


def test_min_p_mode_defaults_to_the_references_behaviour():
    """lm/utils/samplers.py:22-28 never removes a token (its threshold is built from the token's own log-probability):
    the drop-in default must sample plain categorical; the intended rule is selectable."""
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.server.settings import GenerationBlock, ServerSettings

    assert GenerationSettings(min_p=0.1).effective_min_p == 0.0
    assert GenerationSettings(min_p=0.1, min_p_mode="intended").effective_min_p == pytest.approx(0.1)
    assert GenerationSettings(min_p=None, min_p_mode="intended").effective_min_p == 0.0
    with pytest.raises(ValueError):
        GenerationSettings(min_p_mode="sometimes")
    blk = GenerationBlock()
    assert blk.min_p == pytest.approx(0.1) and blk.min_p_mode == "reference" and blk.to_settings().effective_min_p == 0.0
    st = ServerSettings(checkpoint_dir="x", generation={"min_p": 0.2, "min_p_mode": "intended"})
    assert st.generation.to_settings().effective_min_p == pytest.approx(0.2)
