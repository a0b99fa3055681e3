"""The loader against a checkpoint directory the REFERENCE wrote: tests/golden/ref_ckpt_micro/ is the output of the unmodified
``RQTransformer.save_pretrained`` (modeling/model/rq_transformer.py:321-329: ``config.json`` + ``model.pth`` + the tokenizer files),
made by tests/golden/make_lm_goldens.py, which also read it back with the reference's own ``from_pretrained`` (:274-319) and pinned
the grids of ``lm_ref_ckpt_micro.npz`` on that model's forward.  Here: ``load_checkpoint`` -> ``pack_lm`` consumes every key, the
tokenizer file gives the ids of the built-in tokenizer, and the CPU oracle decodes the golden grids from the loaded tensors."""
import json
from pathlib import Path

import numpy as np
import torch

GOLD = Path(__file__).resolve().parent / "golden"
CKPT = GOLD / "ref_ckpt_micro"


def test_reference_written_directory_has_the_contracted_files():
    names = {p.name for p in CKPT.iterdir()}
    assert {"config.json", "model.pth", "tokenizer.json"} <= names
    cfg = json.loads((CKPT / "config.json").read_text())
    assert cfg["model_type"] == "dual_ar" and cfg["dim"] == 64 and cfg["codebook_size"] == 64


def test_loader_and_packer_consume_the_reference_checkpoint_whole():
    from smoltts_amd.checkpoint import load_checkpoint
    from smoltts_amd.config import NumericsMode
    from smoltts_amd.packing import pack_lm
    from smoltts_amd.synthetic import state_fingerprint

    cfg, tok, state = load_checkpoint(CKPT)
    assert (cfg.dim, cfg.n_layer, cfg.n_fast_layer, cfg.codebook_size, cfg.num_codebooks, cfg.vocab_size) == (64, 1, 1, 64, 8, 2368)
    assert all(v.dtype == torch.bfloat16 for v in state.values())  # stored as the released checkpoints are
    g = np.load(GOLD / "lm_ref_ckpt_micro.npz")
    assert state_fingerprint({k: v.float() for k, v in state.items()}) == float(g["fingerprint"])
    rep = {}
    arena, off = pack_lm(cfg, state, NumericsMode.torch_reference(), report=rep)
    assert rep["unused"] == [], rep["unused"]
    assert arena.numel() > 0
    # the tokenizer file the reference saved encodes like the built-in byte-level tokenizer
    gold = json.loads((GOLD / "tokenizer_golden.json").read_text())
    for s, ids in zip(gold["samples"], gold["ids"]):
        assert tok.encode(s).ids == ids


def test_oracle_decodes_the_golden_grids_from_the_reference_checkpoint():
    from oracle.lm_oracle import LMOracle, OracleLMConfig
    from smoltts_amd.checkpoint import load_checkpoint

    cfg, _, state = load_checkpoint(CKPT)
    g = np.load(GOLD / "lm_ref_ckpt_micro.npz")
    orc = LMOracle(OracleLMConfig.from_dict(cfg.__dict__), {k: v.float() for k, v in state.items()}, embed_mask="torch", rope_bf16=True)
    prompts = [torch.from_numpy(g[f"prompt_{b}"]).long() for b in range(2)]
    logs = orc.generate(prompts, max_frames=int(g["frames"]), stop_on_eos=False)
    for b in range(2):
        assert np.array_equal(logs[b].as_tensor().numpy(), g[f"grid_{b}"])
