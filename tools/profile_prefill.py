"""Prefill-only loop for `rocprofv3 --kernel-trace --stats` (B=32, 150m, the bench's prompts)."""
import sys, time
import torch
sys.path.insert(0, ".")
from smoltts_amd.config import NumericsMode, TokenConfig
from smoltts_amd.engine import LMEngine, LMSession, load_library
from smoltts_amd.packing import pack_lm
from smoltts_amd.prompt import PromptEncoder
from smoltts_amd.synthetic import named_config, synthetic_lm_state
from smoltts_amd.tokenizer import load_tokenizer
from bench import make_prompts

load_library()
cfg = named_config(sys.argv[1] if len(sys.argv) > 1 else "smoltts_byte_150m")
tok = load_tokenizer()
pe = PromptEncoder(tok, cfg.num_codebooks)
prompts = make_prompts(pe, 32)
num = NumericsMode.torch_reference()
a, o = pack_lm(cfg, synthetic_lm_state(cfg, seed=0), num)
eng = LMEngine(cfg, None, TokenConfig.from_tokenizer(tok, cfg), num, arena=a, offsets=o)
ls = LMSession(eng, max_batch=32, max_seq=max(p.shape[1] for p in prompts) + 8, max_rows=sum(p.shape[1] for p in prompts), max_frames=4)
print("rows", sum(p.shape[1] for p in prompts))
for i in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    ls.prefill(prompts, stop_on_eos=False)
    torch.cuda.synchronize(); print("prefill ms", (time.perf_counter() - t) * 1e3)
