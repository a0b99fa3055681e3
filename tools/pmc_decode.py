"""Round 1's decode-only harness for the rocprofv3 --pmc passes: smoltts_byte_150m, B=32 slots, 64 frame-graph replays,
synchronised every 3 frames, prompts prefilled 8 slots at a time.  It exists because round 1's passes hung on bench.py; the cause
turned out to be the number of dispatches queued without a synchronisation, not the prefill kernels, and
since `smoltts_lm_decode` bounds the host's run-ahead (SMOLTTS_MAX_FRAMES_IN_FLIGHT) tools/collect_pmc.sh profiles bench.py itself.
Kept for comparison with the round-1 numbers."""
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

load_library()
cfg = named_config("smoltts_byte_150m")
tok = load_tokenizer()
tc = TokenConfig.from_tokenizer(tok, cfg)
num = NumericsMode.torch_reference()
arena, off = pack_lm(cfg, synthetic_lm_state(cfg, seed=0), num)
eng = LMEngine(cfg, None, tc, num, arena=arena, offsets=off)
pe = PromptEncoder(tok, tc.semantic_start_id, cfg.num_codebooks, cfg.duplicate_code_0)
prompts = [p[:, -28:] for p in make_prompts(pe, 32)]  # 28 columns each: 8 slots = 224 rows per prefill call
sess = LMSession(eng, max_batch=32, max_seq=128, max_rows=256, max_frames=80)
print("session ready", flush=True)
for g in range(4):
    sess.prefill(prompts[8 * g: 8 * g + 8], slots=list(range(8 * g, 8 * g + 8)), stop_on_eos=False)
    torch.cuda.synchronize()
    print("prefilled group", g, flush=True)
sess.decode(1)
torch.cuda.synchronize()
print("first frame graph done", flush=True)
for _ in range(21):  # rocprofv3's counter service deadlocks behind a long queue of graph launches: 3 frames per sync
    sess.decode(3)
    torch.cuda.synchronize()
print("frames", sess.fetch()[1].tolist()[:4])
