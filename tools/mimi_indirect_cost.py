"""Where does a step lose time around its Mimi chunk?  bench.py's workload (150m, B=32, chunk 32): HIP events around the 32
frames of a chunk (one graph launch per frame here, so that each frame can be timed) and around the Mimi chunk decode, over
several steps: per-position frame time after a codec pass, the codec pass's own time in situ, and the same frames with no codec
pass in between."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from bench import make_prompts  # noqa: E402
from smoltts_amd.codec.synthetic import synthetic_mimi_state  # noqa: E402
from smoltts_amd.config import NumericsMode, TokenConfig  # noqa: E402
from smoltts_amd.engine import LMEngine, LMSession, MimiEngine, MimiSession, load_library  # noqa: E402
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
meng = MimiEngine(synthetic_mimi_state(seed=0), 8, window=0, max_positions=2 * 700 + 16)
pe = PromptEncoder(tok, tc.semantic_start_id, cfg.num_codebooks, cfg.duplicate_code_0)
prompts = make_prompts(pe, 32)
CH, STEPS = 32, 12


def run(with_mimi):
    sess = LMSession(eng, max_batch=32, max_seq=900, max_rows=sum(p.shape[1] for p in prompts), max_frames=700)
    ms = MimiSession(meng, max_batch=32, max_chunk_frames=CH)
    pcm = torch.zeros(32, 700 * 1920, device="cuda")
    sess.prefill(prompts, stop_on_eos=False)
    sess.set_frames_per_graph(1)
    ms.reset()
    sess.decode(CH * 2)  # warm
    torch.cuda.synchronize()
    frame_us = np.zeros((STEPS, CH))
    mimi_us = np.zeros(STEPS)
    for st in range(STEPS):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(CH + 2)]
        evs[0].record()
        for f in range(CH):
            sess.decode(1)
            evs[f + 1].record()
        if with_mimi:
            ms.decode_chunk(sess.codes, (2 + st) * CH, CH, pcm, code_offset=1)
        evs[CH + 1].record()
        torch.cuda.synchronize()
        frame_us[st] = [evs[f].elapsed_time(evs[f + 1]) * 1e3 for f in range(CH)]
        mimi_us[st] = evs[CH].elapsed_time(evs[CH + 1]) * 1e3
    sess.close(); ms.close()
    return frame_us[2:], mimi_us[2:]


def codec_alone():
    """The same codec passes (same stream positions) back to back, no frame graphs in between."""
    ms = MimiSession(meng, max_batch=32, max_chunk_frames=CH)
    codes = torch.randint(0, 2048, (32, (STEPS + 2) * CH, 9), dtype=torch.int32, device="cuda")
    pcm = torch.zeros(32, 700 * 1920, device="cuda")
    ms.reset()
    out = []
    for st in range(STEPS + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ms.decode_chunk(codes, st * CH, CH, pcm, code_offset=1)
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    ms.close()
    return np.array(out[4:])  # the positions of run()'s measured steps


fa, ma = run(True)
fb, _ = run(False)
mc = codec_alone()
print("frame time by position inside a chunk (us, mean over steps), with a codec pass after every chunk / without any:")
for f in list(range(8)) + [15, 31]:
    print(f"  frame {f:2d}: {fa[:, f].mean():8.1f}   {fb[:, f].mean():8.1f}")
print(f"sum of the 32 frames: {fa.sum(1).mean():.0f} us with codec passes, {fb.sum(1).mean():.0f} us without -> {fa.sum(1).mean() - fb.sum(1).mean():.0f} us lost by the frames")
print(f"codec pass in situ: {ma.mean():.0f} us; the same passes (same stream positions) back to back without frame graphs: {mc.mean():.0f} us")
print("  by step (in situ / alone): " + "  ".join(f"{a:.0f}/{b:.0f}" for a, b in zip(ma, mc)))
