"""Frame-graph time at a long context (smoltts_byte_150m, B=32): argv[1] = context to reach (default 1000)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from bench import make_prompts  # noqa: E402
from smoltts_amd.config import NumericsMode, TokenConfig  # noqa: E402
from smoltts_amd.engine import LMEngine, LMSession, load_library  # noqa: E402
from smoltts_amd.packing import pack_lm  # noqa: E402
from smoltts_amd.prompt import PromptEncoder  # noqa: E402
from smoltts_amd.synthetic import named_config, synthetic_lm_state  # noqa: E402
from smoltts_amd.tokenizer import load_tokenizer  # noqa: E402

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
load_library()
cfg = named_config("smoltts_byte_150m")
tok = load_tokenizer()
tc = TokenConfig.from_tokenizer(tok, cfg)
num = NumericsMode.torch_reference()
arena, off = pack_lm(cfg, synthetic_lm_state(cfg, seed=0), num)
eng = LMEngine(cfg, None, tc, num, arena=arena, offsets=off)
pe = PromptEncoder(tok, tc.semantic_start_id, cfg.num_codebooks, cfg.duplicate_code_0)
prompts = make_prompts(pe, 32)
sess = LMSession(eng, max_batch=32, max_seq=ctx + 64, max_rows=sum(p.shape[1] for p in prompts), max_frames=ctx + 32)
sess.prefill(prompts, stop_on_eos=False)
sess.decode(ctx - 165)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    sess.decode(8)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 8 * 1e6)
print(f"context ~{ctx}: {best:.1f} us per frame (32 slots) -> {32 / best * 1e6:.0f} frames/s LM-only")
