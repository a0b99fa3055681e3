"""Frame-graph time over the context length (smoltts_byte_150m, B=32) for the slow-attention variants:
fp32 / bf16 KV cache x key split on / off (SMOLTTS_OPT_SPLIT_ATTN).  argv: contexts to reach (default 300 600 1000 1800).
One session per variant, grown step by step, timed with HIP events over 8 frames at every stop."""
import sys

import torch

sys.path.insert(0, ".")
from bench import make_prompts  # noqa: E402
from smoltts_amd.config import NumericsMode, TokenConfig  # noqa: E402
from smoltts_amd.engine import LMEngine, LMSession, load_library  # noqa: E402
from smoltts_amd.packing import pack_lm  # noqa: E402
from smoltts_amd.prompt import PromptEncoder  # noqa: E402
from smoltts_amd.synthetic import named_config, synthetic_lm_state  # noqa: E402
from smoltts_amd.tokenizer import load_tokenizer  # noqa: E402

stops = [int(a) for a in sys.argv[1:]] or [300, 600, 1000, 1800]
load_library()
cfg = named_config("smoltts_byte_150m")
cfg.max_seq_len = max(cfg.max_seq_len, max(stops) + 128)  # (the RoPE table follows the config)
tok = load_tokenizer()
tc = TokenConfig.from_tokenizer(tok, cfg)
num = NumericsMode.torch_reference()
arena, off = pack_lm(cfg, synthetic_lm_state(cfg, seed=0), num)
eng = LMEngine(cfg, None, tc, num, arena=arena, offsets=off)
pe = PromptEncoder(tok, tc.semantic_start_id, cfg.num_codebooks, cfg.duplicate_code_0)
prompts = make_prompts(pe, 32)
T = max(p.shape[1] for p in prompts)
res = {}
for kv in ("fp32", "bf16"):
    for split in (True, False):
        sess = LMSession(eng, max_batch=32, max_seq=max(stops) + 96, max_rows=sum(p.shape[1] for p in prompts), max_frames=max(stops) + 64, kv_dtype=kv)
        sess.use_split_attention(split)
        sess.prefill(prompts, stop_on_eos=False)
        done = 0
        for ctx in stops:
            n = ctx - T - done
            if n > 0:
                sess.decode(n)
                done += n
            sess.decode(8)  # graphs warm
            done += 8
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                sess.decode(8)
                b.record()
                torch.cuda.synchronize()
                done += 8
                best = min(best, a.elapsed_time(b) * 1e3 / 8)
            res[kv, split, ctx] = best
        sess.close()
print(f"{'context (longest slot)':24s}" + "".join(f"{c:>10d}" for c in stops))
for kv in ("fp32", "bf16"):
    for split in (False, True):
        print(f"{kv + ' KV, ' + ('keys split over 2 WGs' if split else 'one WG per pair'):24s}" + "".join(f"{res[kv, split, c]:10.1f}" for c in stops) + "   us per frame")
for c in stops:
    print(f"context {c}: bf16+split vs fp32 unsplit (round 2's default) {100 * (1 - res['bf16', True, c] / res['fp32', False, c]):.1f} % faster; "
          f"fp32 split vs unsplit {100 * (1 - res['fp32', True, c] / res['fp32', False, c]):.1f} %; bf16 split vs fp32 split {100 * (1 - res['bf16', True, c] / res['fp32', True, c]):.1f} %")
