"""In-situ (marginal) cost of each kernel class of the frame graph: the graph is re-captured with every launch of one
class issued twice (they are idempotent) and timed with HIP events; (t_dup - t_base) / extra launches.
smoltts_byte_150m, B=32 (argv[1] overrides B)."""
import sys

import torch

sys.path.insert(0, ".")
from bench import make_prompts  # noqa: E402
from smoltts_amd.config import NumericsMode, TokenConfig  # noqa: E402
from smoltts_amd.engine import LMEngine, LMSession, check, load_library  # noqa: E402
from smoltts_amd.packing import pack_lm  # noqa: E402
from smoltts_amd.prompt import PromptEncoder  # noqa: E402
from smoltts_amd.synthetic import named_config, synthetic_lm_state  # noqa: E402
from smoltts_amd.tokenizer import load_tokenizer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = load_library()
cfg = named_config("smoltts_byte_150m")
tok = load_tokenizer()
tc = TokenConfig.from_tokenizer(tok, cfg)
num = NumericsMode.torch_reference()
arena, off = pack_lm(cfg, synthetic_lm_state(cfg, seed=0), num)
eng = LMEngine(cfg, None, tc, num, arena=arena, offsets=off)
pe = PromptEncoder(tok, tc.semantic_start_id, cfg.num_codebooks, cfg.duplicate_code_0)
prompts = make_prompts(pe, B)
sess = LMSession(eng, max_batch=B, max_seq=700, max_rows=sum(p.shape[1] for p in prompts), max_frames=600)
sess.prefill(prompts, stop_on_eos=False)
sess.decode(200)  # context ~ 300, as in the middle of the bench
torch.cuda.synchronize()


def timed(n=8):
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sess.decode(1)
        a.record()
        sess.decode(n)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / n)
    return best


base = timed()
print(f"frame graph: {base:.1f} us")
L, F, S = cfg.n_layer, cfg.n_fast_layer, cfg.max_fast_seqlen
wqkv_launches = L + F * S - ((S - 1) if eng.fast_qkv is not None else 0)  # the table replaces depth layer 0's wqkv in steps 1..S-1
cases = [("wqkv GEMM + RoPE + cache (N=1280)", 5, (cfg.n_head + 2 * cfg.n_local_heads) * 64, wqkv_launches),
         ("w1|w3 GEMM + SwiGLU", 2, 2 * cfg.intermediate_size, L + F * S),
         ("slow head GEMM (N=2368)", 0, cfg.vocab_size, 1),
         ("depth head GEMM (N=2048)", 0, cfg.codebook_size, S),
         ("depth attention (<= 8 keys)", 100, 0, F * (S - 1)),
         ("slow attention (context ~300)", 101, 0, L)]
for name, epi, n, count in cases:
    sess.measure_duplicate(epi, n)  # this session only; drops its captured graphs
    t = timed()
    print(f"{name:40s}: {count:3d} extra launches, frame {t:7.1f} us -> {(t - base) / count:5.2f} us per launch")
sess.measure_duplicate(-1)
