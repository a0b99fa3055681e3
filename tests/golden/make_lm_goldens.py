"""Generate the committed LM golden vectors.  Runs ONLY in the build container.

It imports the *reference* torch model from /root/reference (read-only, never copied), loads a
seeded synthetic checkpoint into it and uses its unmodified teacher-forced ``RQTransformer.forward``
(modeling/model/rq_transformer.py:401-479) to pin our CPU oracle (oracle/lm_oracle.py):

1. oracle.teacher_forced(grid) logits == reference forward logits (fp32, tight tolerance);
2. the oracle's greedy, KV-cached generation is self-consistent with the reference forward:
   for every generated position s, argmax(reference token_logits[s]) == grid[0, s+1] and
   argmax(reference codebook_logits[s]) == grid[1:, s+1]  (SURVEY.md §8c);
3. the byte-level tokenizer ids == the ids of the tokenizer produced by the reference's own
   data_pipeline/scripts/create_bytelevel_init.py.

What is committed (tests/golden/lm_*.npz) is data only: seeds, prompt text/ids, generated id grids,
a few reference logits rows, top-2 margins and the weight fingerprint -- plus tests/golden/ref_ckpt_micro/, a checkpoint
directory written by the reference's own save_pretrained (make_reference_checkpoint below; argument `ref_ckpt` makes only that).

Usage:  PYTHONPATH=/root/reference:/root/repo TORCH_COMPILE_DISABLE=1 PYTHONDONTWRITEBYTECODE=1 \
        python tests/golden/make_lm_goldens.py
"""
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("TORCH_COMPILE_DISABLE", "1")

from oracle.lm_oracle import LMOracle, OracleLMConfig  # noqa: E402
from smoltts_amd.prompt import PromptEncoder  # noqa: E402
from smoltts_amd.synthetic import named_config, state_fingerprint, synthetic_lm_state  # noqa: E402
from smoltts_amd.tokenizer import load_tokenizer  # noqa: E402

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent

CASES = [
    # name, config, seed, frames, [(text, voice)]
    ("tiny", "tiny", 7, 12, [("Hello world!", "heart"), ("The quick brown fox.", "sky")]),
    ("70m", "smoltts_byte_70m", 0, 16, [("Hello world!", "heart")]),
    # checkpoint variants: duplicate_code_0=false; fast_dim != dim + untied head + Linear depth head
    ("tiny_nodup", "tiny_nodup", 11, 12, [("Hello world!", "heart"), ("No duplicated code zero.", "emma")]),
    ("tiny_proj", "tiny_proj", 12, 12, [("Hello world!", "heart"), ("A narrower depth transformer.", "liam")]),
    ("150m", "smoltts_byte_150m", 0, 16, [("Hello world!", "heart"), ("Streaming speech on MI355X, 12.5 frames per second.", "nova")]),
]


def build_reference(cfg, state, tokdir):
    sys.path.insert(0, str(REF))
    from modeling.model.rq_transformer import RQTransformer, RQTransformerModelArgs as RefArgs
    from transformers import AutoTokenizer

    ref_cfg = RefArgs(**{k: v for k, v in cfg.__dict__.items()})
    tok = AutoTokenizer.from_pretrained(tokdir)
    model = RQTransformer(ref_cfg, tokenizer=tok)
    missing = model.load_state_dict(state, strict=True)
    print("reference load_state_dict:", missing)
    model.eval()
    return model, tok


def main():
    tokdir = tempfile.mkdtemp(prefix="smoltts_tok_")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.run(
        [sys.executable, str(REF / "data_pipeline/scripts/create_bytelevel_init.py"), "--out-dir", tokdir],
        check=True, env=env, stdout=subprocess.DEVNULL,
    )
    ours = load_tokenizer()
    from tokenizers import Tokenizer

    ref_tok_json = Tokenizer.from_file(os.path.join(tokdir, "tokenizer.json"))
    tok_samples = ["Hello world!", "<|im_start|>system\n<|speaker:3|><|im_end|>", "a user asked the system assistant",
                   "café 你好 <|semantic:12|><|semantic:2047|>", "<|im_start|>assistant\n"]
    tok_ids = []
    for s in tok_samples:
        a = ref_tok_json.encode(s, add_special_tokens=True).ids
        assert a == ours.encode(s).ids, s
        tok_ids.append(a)
    special = {t: ref_tok_json.token_to_id(t) for t in
               ["system", "user", "assistant", "<|im_start|>", "<|im_end|>", "<|speaker:0|>", "<|speaker:48|>",
                "<|semantic:0|>", "<|semantic:2047|>", "<|pad|>", "<|semantic|>"]}
    (OUT / "tokenizer_golden.json").write_text(json.dumps(
        {"samples": tok_samples, "ids": tok_ids, "special": special, "vocab_size": ref_tok_json.get_vocab_size()},
        indent=1, ensure_ascii=True))
    print("tokenizer pinned:", special)

    only = set(sys.argv[1:])  # optional: names of the cases to (re)generate
    for name, cfgname, seed, frames, prompts in CASES:
        if only and name not in only:
            continue
        cfg = named_config(cfgname)
        state = synthetic_lm_state(cfg, seed=seed)
        ref, _ = build_reference(cfg, state, tokdir)
        save = {"seed": seed, "fingerprint": state_fingerprint(state), "config_name": cfgname}
        pin_case(name, cfg, state, ref, ours, prompts, frames, save)
        np.savez_compressed(OUT / f"lm_{name}.npz", **save)
        print("wrote", OUT / f"lm_{name}.npz")
    if not only or "ref_ckpt" in only:
        make_reference_checkpoint(tokdir, ours)


def pin_case(name, cfg, state, ref, ours, prompts, frames, save):
    """Oracle generation from `state`, checked against the reference model `ref` (teacher-forced logits + every generated id is the
    reference forward's argmax); prompts, grids and a few reference logits rows go into `save`."""
    ocfg = OracleLMConfig.from_dict(cfg.__dict__)
    oracle = LMOracle(ocfg, state, embed_mask="torch", rope_bf16=True)
    pe = PromptEncoder(ours, 320, cfg.num_codebooks, cfg.duplicate_code_0)
    grids = [torch.from_numpy(pe.build_prompt(t, v)).long() for t, v in prompts]
    logs = oracle.generate(grids, max_frames=frames, stop_on_eos=False)
    save.update({"frames": frames, "texts": np.array([t for t, _ in prompts]), "voices": np.array([v for _, v in prompts])})
    for b, (g, log) in enumerate(zip(grids, logs)):
        gen = log.as_tensor()  # 9,F
        full = torch.cat([g, gen], dim=1)  # 9, T+F
        with torch.no_grad():
            out = ref(full[None])
        tl, cl = out.token_logits[0], out.codebook_logits[0]  # (S,V), (S,n,2048)
        T = g.shape[1]
        # (1) oracle teacher-forced == reference forward
        otl, ocl = oracle.teacher_forced(full)
        e1 = float((otl - tl).abs().max()); e2 = float((ocl - cl).abs().max())
        print(f"[{name}/{b}] teacher-forced max|diff| token {e1:.3e} codebook {e2:.3e}")
        assert e1 < 2e-4 and e2 < 2e-5, (e1, e2)
        # (2) self-consistency of the generated ids under the reference forward
        for f in range(frames):
            s = T - 1 + f
            assert int(tl[s].argmax()) == int(gen[0, f]), (name, b, f, "slow")
            assert cl[s].argmax(-1).tolist() == gen[1:, f].tolist(), (name, b, f, "fast")
        print(f"[{name}/{b}] {frames} frames self-consistent; min top-2 margin {log.min_margin:.3e}")
        save[f"prompt_{b}"] = g.numpy().astype(np.int32)
        save[f"grid_{b}"] = gen.numpy().astype(np.int32)
        save[f"min_margin_{b}"] = log.min_margin
        rows = [T - 1, T, T + frames - 2]
        save[f"ref_rows_{b}"] = np.array(rows)
        save[f"ref_token_logits_{b}"] = tl[rows].numpy().astype(np.float32)
        save[f"ref_codebook_logits_{b}"] = cl[rows][:, :, :64].numpy().astype(np.float32)


def make_reference_checkpoint(tokdir, ours):
    """A checkpoint directory written by the REFERENCE's own writer (RQTransformer.save_pretrained, modeling/model/rq_transformer.py
    :321-329: config.json + model.pth + the tokenizer files), read back by its own reader (from_pretrained :274-319) and by ours
    (smoltts_amd.checkpoint.load_checkpoint): tests/golden/ref_ckpt_micro/ + lm_ref_ckpt_micro.npz (prompts, grids).  The config is
    as small as the real tokenizer allows (vocab 2368 x dim 64 is most of model.pth): 1 + 1 layers, one 64-wide head, 64-entry
    codebooks; weights = the reference's own initialisation (seeded), norm weights perturbed, stored in bf16 like the released ones."""
    import dataclasses
    import shutil

    sys.path.insert(0, str(REF))
    from modeling.model.rq_transformer import BaseTransformer, RQTransformer, RQTransformerModelArgs as RefArgs
    from transformers import AutoTokenizer

    from smoltts_amd.checkpoint import load_checkpoint
    from smoltts_amd.synthetic import tiny_config

    cfg = dataclasses.replace(tiny_config(), dim=64, n_head=1, n_local_heads=1, n_layer=1, intermediate_size=128, fast_dim=64,
                              fast_n_head=1, fast_n_local_heads=1, n_fast_layer=1, fast_intermediate_size=128, codebook_size=64,
                              max_seq_len=256)
    torch.manual_seed(4321)
    model = RQTransformer(RefArgs(**cfg.__dict__), tokenizer=AutoTokenizer.from_pretrained(tokdir))
    for k, p in model.named_parameters():
        if k.endswith("norm.weight"):
            p.data.normal_(1.0, 0.1)
    out = OUT / "ref_ckpt_micro"
    if out.exists():
        shutil.rmtree(out)
    model.to(torch.bfloat16).save_pretrained(str(out))
    print("reference save_pretrained wrote", sorted(p.name for p in out.iterdir()))
    # the reference reads its own directory back (bf16 parameters, widened here for the fp32 comparison) ...
    ref = BaseTransformer.from_pretrained(str(out), load_weights=True).float().eval()
    # ... and so does this repo's loader
    lcfg, ltok, lstate = load_checkpoint(out)
    assert lcfg == cfg, (lcfg, cfg)
    assert set(lstate) == set(model.state_dict()), set(lstate) ^ set(model.state_dict())
    state = {k: v.float() for k, v in lstate.items()}
    save = {"seed": 4321, "fingerprint": state_fingerprint(state), "config_name": "ref_ckpt_micro"}
    pin_case("ref_ckpt_micro", lcfg, state, ref, ours, [("Hello world!", "heart"), ("Written by save_pretrained.", "nova")], 12, save)
    np.savez_compressed(OUT / "lm_ref_ckpt_micro.npz", **save)
    print("wrote", OUT / "lm_ref_ckpt_micro.npz")

if __name__ == "__main__":
    main()
